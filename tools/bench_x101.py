"""BASELINE configs[4] on one GPU: X-101-32x8d-FPN inference on native 2048x2048 synthetic micrographs (dense: DETECTIONS_PER_IMAGE 500)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ampis_amd import _lib, params as P, synth
from ampis_amd.model import MaskRCNN

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
S, K, D = 2048, 2, 500
ctx = _lib.Context(0)
t = time.time()
model = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, detections_per_image=D, pixel_std=(57.375, 57.120, 58.395), arch="X101")
p = P.init_params(K, seed=0, style="spread", arch="X101")
p["backbone.bottom_up.stem.conv1.weight"] = p["backbone.bottom_up.stem.conv1.weight"] * np.float32(57.0)
model.load_params(p)
print("workspace GiB", model.workspace_bytes / 2**30, "setup s", time.time() - t, flush=True)
imgs, _ = synth.batch(B, S, S)
d_imgs = ctx.malloc(imgs.nbytes); ctx.h2d(d_imgs, imgs)
for i in range(2):
    t = time.time(); d = model.infer_raw(None, device_ptr=d_imgs, shape=(B, S, S)); print("warm", time.time() - t, [d.n[b] for b in range(B)], flush=True)
n = 5
ctx.prof_begin(max_launches=n * 200)
t = time.time()
for i in range(n):
    model.infer_raw(None, device_ptr=d_imgs, shape=(B, S, S))
ctx.sync()
dt = (time.time() - t) / n
prof = ctx.prof_end()
print(json.dumps({"x101_2048_ms_per_batch": dt * 1e3, "images_per_s": B / dt, "B": B,
                  "conv_tflops_useful": (prof["flops"][0] + prof["flops"][1]) / ((prof["ms"][0] + prof["ms"][1]) * 1e-3) / 1e12}))
