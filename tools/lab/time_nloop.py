"""Lab: conv1x1_nloop_kernel against conv_split_kernel<128,256> on the short-K 1x1 layers (A/B in one process)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ampis_amd import ops, _lib
ctx = ops.torch_context(0)
L = _lib.lib()
SH = [("train deconv 2048 RoIs", 2048, 14, 14, 256, 1024, 1, True), ("res3.0 shortcut B=16", 16, 256, 256, 256, 512, 2, False),
      ("res4.0 shortcut B=16", 16, 128, 128, 512, 1024, 2, False), ("res3.0 shortcut B=8", 8, 256, 256, 256, 512, 2, False)]
for name, B, H, W, Cin, Cout, stride, dc in SH:
    x = ops.split_rows(ctx, torch.randn(B, H, W, Cin, device="cuda:0"))
    w = torch.randn(Cout, 1, 1, Cin, device="cuda:0") * 0.05
    sh = torch.zeros(Cout, device="cuda:0")
    kw = dict(stride=stride, pad=0, relu=dc, deconv2x2=dc, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
    for rep in range(2):
        for on in (0, 1):
            L.amp_debug_set_nloop(on)
            for _ in range(5):
                ops.conv2d_nhwc(ctx, x, w, None, sh, **kw)
            torch.cuda.synchronize()
            ctx.timer_start()
            for _ in range(20):
                ops.conv2d_nhwc(ctx, x, w, None, sh, **kw)
            ms = ctx.timer_stop() / 20
            print(f"{name:28s} nloop={on}: {ms * 1e3:8.1f} us", flush=True)
L.amp_debug_set_nloop(0)
