"""X-101-32x8d-FPN training step time (BASELINE configs[4]'s backbone; correctness-first grouped gradients, grouped_bwd.hip)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ampis_amd import _lib, params as P, synth
from ampis_amd.model import MaskRCNN, PackedGt
B, S, K = int(os.environ.get("B", "4")), int(os.environ.get("S", "1024")), 2
ctx = _lib.Context(0)
imgs, gts = synth.batch(B, S, S, first_index=50)
m = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, arch="X101", train=True, max_gt=B * 800, max_poly_doubles=B * 800 * 64)
m.load_params(P.init_params(K, seed=0, style="spread", arch="X101"))
print("workspace GiB", round(m.workspace_bytes / 2**30, 1), flush=True)
d = ctx.malloc(imgs.nbytes); ctx.h2d(d, imgs)
g = PackedGt(gts)
for i in range(3):
    L = m.forward_losses(None, g, seed=i, backward=True, device_ptr=d, shape=(B, S, S)); m.sgd_step(1e-3)
ctx.sync(); t0 = time.perf_counter()
n = 8
for i in range(n):
    L = m.forward_losses(None, g, seed=10 + i, backward=True, device_ptr=d, shape=(B, S, S)); m.sgd_step(1e-3)
ctx.sync(); el = time.perf_counter() - t0
print(f"X-101-32x8d-FPN training, B={B} at {S}x{S}: {el / n * 1e3:.1f} ms per step = {B * n / el:.1f} images/s; losses {L}")
