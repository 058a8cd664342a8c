"""Lab: one 3x3 layer, N launches (for rocprofv3 --pmc runs). usage: python3 tools/lab/run_conv.py B H W Cin Cout [n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ampis_amd import ops
B, H, W, Cin, Cout = map(int, sys.argv[1:6])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 10
ctx = ops.torch_context(0)
x = ops.split_rows(ctx, torch.randn(B, H, W, Cin, device="cuda:0"))
w = torch.randn(Cout, 3, 3, Cin, device="cuda:0") * 0.05
sc = torch.ones(Cout, device="cuda:0"); sh = torch.zeros(Cout, device="cuda:0")
for _ in range(n):
    ops.conv2d_nhwc(ctx, x, w, sc, sh, stride=1, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
torch.cuda.synchronize()
