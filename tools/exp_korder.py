"""Channel-major K order of conv_split_kernel (ConvArgs::korder, AMP_KORDER=1) against the tap-major one: same convolution within fp32
re-association (checked here against an fp64 torch reference on three 3x3 shapes), timed per layer by tools/layer_roofline.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops, _lib

ctx = ops.torch_context(0)
L = _lib.lib()
torch.manual_seed(0)
for (B, H, W, Cin, Cout, s) in [(2, 64, 64, 256, 256, 1), (1, 33, 47, 128, 256, 1), (2, 32, 32, 512, 512, 2), (4, 14, 14, 256, 256, 1)]:
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.05
    sc = torch.rand(Cout, device="cuda") + 0.5
    sh = torch.randn(Cout, device="cuda")
    xs = ops.split_rows(ctx, x)
    ref = torch.nn.functional.conv2d(ops.unsplit_rows(ctx, xs).double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=s, padding=1)
    ref = torch.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).permute(0, 2, 3, 1)
    outs = []
    for ko in (0, 1):
        L.amp_debug_set_korder(ko)
        y = ops.conv2d_nhwc(ctx, xs, w, sc, sh, stride=s, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
        y = ops.unsplit_rows(ctx, y)
        torch.cuda.synchronize()
        outs.append(y)
        err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
        print(f"B{B} {H}x{W} {Cin}->{Cout} s{s} korder {ko}: max err / max|ref| = {err:.2e}", flush=True)
        assert err < 2e-6, err
    print("   korder 0 vs 1: max |d| =", (outs[0] - outs[1]).abs().max().item(), "bit-identical" if torch.equal(outs[0], outs[1]) else "")
L.amp_debug_set_korder(0)
print("exp_korder OK")
