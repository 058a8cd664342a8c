"""Micro-benchmark of amp_conv2d_wgrad on the R50-FPN training layer shapes (B=16, 1024x1024). Prints TFLOP/s per shape."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops

SHAPES = [
    # name, B, H, W, Cin, Cout, k, stride, pad
    ("res3.3x3", 16, 128, 128, 128, 128, 3, 1, 1),
    ("res3.1x1c", 16, 128, 128, 128, 512, 1, 1, 0),
    ("res3.1x1a", 16, 128, 128, 512, 128, 1, 1, 0),
    ("res4.3x3", 16, 64, 64, 256, 256, 3, 1, 1),
    ("res4.1x1c", 16, 64, 64, 256, 1024, 1, 1, 0),
    ("res5.3x3", 16, 32, 32, 512, 512, 3, 1, 1),
    ("fpn.out.p2", 16, 256, 256, 256, 256, 3, 1, 1),
    ("fpn.lat.p2", 16, 256, 256, 256, 256, 1, 1, 0),
    ("fpn.out.p3", 16, 128, 128, 256, 256, 3, 1, 1),
    ("mask.3x3(2048)", 2048, 14, 14, 256, 256, 3, 1, 1),
    ("fc1(8192)", 1, 1, 8192, 12544, 1024, 1, 1, 0),
]

def main():
    ctx = ops.torch_context(0)
    if os.environ.get("AMP_CONV_MODE"):
        ctx.conv_mode = os.environ["AMP_CONV_MODE"]
    print("conv mode:", ctx.conv_mode, flush=True)
    d = "cuda:0"
    only = os.environ.get("AMP_ONLY")
    tot = 0.0
    for name, B, H, W, Cin, Cout, k, s, p in SHAPES:
        if only and not any(name.startswith(o) for o in only.split(',')):
            continue
        x = torch.randn(B, H, W, Cin, device=d)
        Ho = (H + 2 * p - k) // s + 1; Wo = (W + 2 * p - k) // s + 1
        dy = torch.randn(B, Ho, Wo, Cout, device=d)
        flops = 2.0 * B * Ho * Wo * Cout * k * k * Cin
        xs = int(os.environ.get("AMP_SPLIT_IN", "0")) if Cin % 32 == 0 and Cout % 32 == 0 else 0   # 1: x split; 3: dy too (scaled by 2^16)
        if xs & 1:
            x = ops.split_rows(ctx, x)
            name += " [x split]"
        if xs & 2:
            dy = ops.split_rows(ctx, dy * 65536.0)
            name += " [dy split]"
        g = None
        for _ in range(2):
            g = ops.conv2d_wgrad(ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, grad=g, dy_shift=16, x_split=xs)
        torch.cuda.synchronize()
        n = 5
        ctx.timer_start()
        for _ in range(n):
            ops.conv2d_wgrad(ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, grad=g, dy_shift=16, x_split=xs)
        ms = ctx.timer_stop() / n
        tot += ms
        print(f"{name:18s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s  (M={B*Ho*Wo}, N={Cout}, K'={k*k*Cin})", flush=True)
        del x, dy, g
    print(json.dumps({"total_ms": tot}))

if __name__ == "__main__":
    main()
