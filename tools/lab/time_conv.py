"""Lab: time one 3x3 layer under (AMP_PATCH256, AMP_KORDER) settings in one process (A/B on one box).
usage: python3 tools/lab/time_conv.py B H W Cin Cout p256:korder [p256:korder ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ampis_amd import ops, _lib
B, H, W, Cin, Cout = map(int, sys.argv[1:6])
cfgs = [tuple(map(int, c.split(":"))) for c in sys.argv[6:]]
ctx = ops.torch_context(0)
L = _lib.lib()
zero = os.environ.get("AMP_LAB_ZERO") == "1"        # all-zero operands: the same instruction stream without the data-dependent switching energy
x = ops.split_rows(ctx, torch.zeros(B, H, W, Cin, device="cuda:0") if zero else torch.randn(B, H, W, Cin, device="cuda:0"))
w = torch.zeros(Cout, 3, 3, Cin, device="cuda:0") if zero else torch.randn(Cout, 3, 3, Cin, device="cuda:0") * 0.05
sc = torch.ones(Cout, device="cuda:0"); sh = torch.zeros(Cout, device="cuda:0")
def run(n):
    for _ in range(n):
        ops.conv2d_nhwc(ctx, x, w, sc, sh, stride=1, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
for rep in range(3):
    for p, k in cfgs:
        L.amp_debug_set_patch256(p); L.amp_debug_set_korder(k)
        run(20); torch.cuda.synchronize()
        ctx.timer_start(); run(50); ms = ctx.timer_stop() / 50
        print(f"{'zero ' if zero else 'randn'} patch256={p} korder={k}: {ms * 1e3:8.1f} us  {2.0 * B * H * W * Cout * 9 * Cin / ms / 1e9:6.1f} TFLOP/s")
L.amp_debug_set_patch256(0); L.amp_debug_set_korder(0)
