"""Box sizes / run counts of the bench workload's detections (what the paste kernel works on)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ampis_amd import _lib, params as P, synth
from ampis_amd.model import MaskRCNN
ctx = _lib.Context(0)
m = MaskRCNN(ctx, 2, max_batch=8, max_h=1024, max_w=1024, max_out_hw=1024, detections_per_image=200)
m.load_params(P.init_params(2, seed=0, style="spread"))
imgs, _ = synth.batch(8, 1024, 1024, first_index=0)
res = m.infer(imgs, rle="counts")
w = np.concatenate([r["boxes"][:, 2] - r["boxes"][:, 0] for r in res]); h = np.concatenate([r["boxes"][:, 3] - r["boxes"][:, 1] for r in res])
print("dets", len(w), "w pct", np.percentile(w, [5, 25, 50, 75, 95, 100]).round(1), "h pct", np.percentile(h, [5, 25, 50, 75, 95, 100]).round(1))
print("mean area", float((w * h).mean()), "sum area Mpx", float((w * h).sum() / 1e6))
runs = np.array([len(mk) for r in res for mk in r["masks"]]) if isinstance(res[0]["masks"][0], np.ndarray) else None
if runs is not None: print("runs pct", np.percentile(runs, [5, 50, 95, 100]))
m.tap_enable(True) if hasattr(m, "tap_enable") else None
