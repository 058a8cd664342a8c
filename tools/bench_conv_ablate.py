"""Ablation of conv_mfma_kernel<128,128> (guide §5.4 rule 17): where does a K-step spend its time?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops
from ampis_amd._lib import lib
SHAPES = [("fpn.out.p2 3x3", 8, 256, 256, 256, 256, 3, 1, 1), ("res4.3x3", 8, 64, 64, 256, 256, 3, 1, 1), ("fc1", 1, 1, 8000, 12544, 1024, 1, 1, 0),
          ("res3.1x1c", 8, 128, 128, 128, 512, 1, 1, 0)]
ctx = ops.torch_context(0); d = "cuda:0"
for name, B, H, W, Cin, Cout, k, s, p in SHAPES:
    x = torch.randn(B, H, W, Cin, device=d); w = torch.randn(Cout, k, k, Cin, device=d) * 0.05
    Ho = (H + 2 * p - k) // s + 1; Wo = (W + 2 * p - k) // s + 1
    fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    res = []
    for rnd in range(3):
        for mode in (0, 1, 2, 3):
            lib().amp_debug_set_conv_ablate(mode)
            ops.conv2d_nhwc(ctx, x, w, stride=s, pad=p, relu=True); torch.cuda.synchronize()
            ctx.timer_start()
            for _ in range(5): ops.conv2d_nhwc(ctx, x, w, stride=s, pad=p, relu=True)
            ms = ctx.timer_stop() / 5
            res.append((rnd, mode, ms))
    lib().amp_debug_set_conv_ablate(0)
    best = {m: min(r[2] for r in res if r[1] == m) for m in (0, 1, 2, 3)}
    print(f"{name:16s} " + "  ".join(f"ABL{m}: {best[m]:.3f} ms ({fl / best[m] / 1e9:6.1f} TF)" for m in (0, 1, 2, 3)), flush=True)
