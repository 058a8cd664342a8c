"""How far is the fp32 oracle from an exact-convolution evaluation of the same network?  (oracle/exact.py)
python tools/oracle_noise_floor.py [image indices ...]   -- CPU only, ~25 s per 1024x1024 image."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ampis_amd import params as P, synth   # noqa: E402
from oracle import exact, maskrcnn as O    # noqa: E402

S, K, D = 1024, 2, 200
idx = [int(a) for a in sys.argv[1:]] or [7, 12]
p = O.to_torch_params(P.init_params(K, seed=0, style="spread"))
cfg = O.Cfg(num_classes=K, detections_per_image=D)
for i in idx:
    img, _ = synth.batch(1, S, S, first_index=i)
    r32 = O.infer(img, p, cfg)[0]
    with exact.exact_convs():
        rex = O.infer(img, p, cfg)[0]
    b32, bex = r32["boxes"].numpy().astype(np.float64), rex["boxes"].numpy().astype(np.float64)
    rows = []
    for k in range(len(bex)):
        d = np.abs(b32 - bex[k]).max(axis=1)
        rows.append((float(d.min()), float(bex[k][2] - bex[k][0]), float(bex[k][3] - bex[k][1])))
    rows.sort(reverse=True)
    print(f"image {i}: {len(b32)} / {len(bex)} detections; worst |box_fp32 - box_exact| = {rows[0][0]:.2e} px on a {rows[0][1]:.0f} x {rows[0][2]:.0f} px box "
          f"({rows[0][0] / max(rows[0][1], rows[0][2]) * 1e6:.2f} ppm of its side); {sum(r[0] > 1e-3 for r in rows)} boxes > 1e-3 px, "
          f"{sum(r[0] > 5e-4 for r in rows)} > 5e-4 px")
