"""Times one training step (forward + losses + backward + SGD) at BASELINE configs[2]: B=16 synthetic 1024x1024, K=2."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ampis_amd import _lib, params as P, synth
from ampis_amd.model import MaskRCNN
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
S, K = 1024, 2
ctx = _lib.Context(0)
t = time.time()
model = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=B * 700, max_poly_doubles=B * 700 * 64)
print("workspace GiB", model.workspace_bytes / 2**30, "create s", time.time() - t, flush=True)
model.load_params(P.init_params(K, seed=0, style="spread"))
imgs, gts = synth.batch(B, S, S)
print("gt per image", [len(g["boxes"]) for g in gts][:4], flush=True)
for i in range(2):
    t = time.time(); L = model.forward_losses(imgs, gts, seed=i, backward=True); model.sgd_step(0.001); ctx.sync(); print("warm", time.time() - t, L, flush=True)
n = int(os.environ.get("BENCH_TRAIN_STEPS", "10"))
dts = []
for rep in range(3):          # three timed blocks, the fastest one reported (A/B runs of a few tenths of a millisecond need it)
    t = time.time()
    for i in range(n):
        L = model.forward_losses(imgs, gts, seed=10 + i, backward=True)
        model.sgd_step(0.001)
    ctx.sync()
    dts.append((time.time() - t) / n)
dt = min(dts)
print(json.dumps({"train_step_ms": dt * 1e3, "images_per_s": B / dt, "B": B, "losses": L}))
