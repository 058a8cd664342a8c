"""RoIAlign backward at the training bench's shape (B=16, 1024^2 -> p2..p5, 512 RoIs / image, P=7; 128 fg RoIs / image, P=14):
the owner-computes kernel (no atomics, bitwise reproducible) against the float-atomic kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops, _lib

ctx = ops.torch_context(0)
d = "cuda:0"
B, S = int(os.environ.get("B", 16)), 1024
g = torch.Generator().manual_seed(0)
strides = (4, 8, 16, 32)
for P, per in ((7, 512), (14, 128)):
    R = B * per
    ctr = torch.rand(R, 2, generator=g) * S
    size = torch.exp(torch.randn(R, 2, generator=g) * 0.6 + 4.3).clamp(8, 800)
    rois = torch.cat([ctr - size / 2, ctr + size / 2], 1).clamp(0, S).float().to(d)
    bidx = (torch.arange(R) // per).int().to(d)
    dout = torch.randn(R, P, P, 256, device=d)
    for tag, atomics in (("owner-computes", 0), ("float atomics", 1)):
        _lib.lib().amp_debug_set_roi_bwd_atomics(atomics)
        feats = [torch.zeros(B, S // s, S // s, 256, device=d) for s in strides]
        for _ in range(2):
            ops.roi_align_bwd(ctx, feats, strides, rois, bidx, P, dout, B=B)
        torch.cuda.synchronize()
        ctx.timer_start()
        for _ in range(3):
            ops.roi_align_bwd(ctx, feats, strides, rois, bidx, P, dout, B=B)
        ms = ctx.timer_stop() / 3
        print(f"P={P:2d} R={R:5d} {tag:16s} {ms:8.3f} ms", flush=True)
        del feats
_lib.lib().amp_debug_set_roi_bwd_atomics(0)
