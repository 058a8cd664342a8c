"""Micro-benchmark of amp_conv2d_nhwc on the R50-FPN layer shapes (B=8, 1024x1024). Prints TFLOP/s per shape."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops

SHAPES = [
    # name, B, H, W, Cin, Cout, k, stride, pad
    ("res2.1x1c+res", 8, 256, 256, 64, 256, 1, 1, 0, "res"),
    ("res3.1x1c+res", 8, 128, 128, 128, 512, 1, 1, 0, "res"),
    ("res4.1x1c+res", 8, 64, 64, 256, 1024, 1, 1, 0, "res"),
    ("fpn.lat.p2+up", 8, 256, 256, 256, 256, 1, 1, 0, "up"),
    ("mask.deconv", 1600, 14, 14, 256, 1024, 1, 1, 0, "deconv"),
    ("mask.pred(K=2)", 1600, 28, 28, 256, 2, 1, 1, 0),
    ("stem7x7(kw8,c4)", 8, 1024, 1024, 4, 64, (7, 8), 2, 3),
    ("res2.1x1a", 8, 256, 256, 64, 64, 1, 1, 0),
    ("res2.3x3", 8, 256, 256, 64, 64, 3, 1, 1),
    ("res2.1x1c", 8, 256, 256, 64, 256, 1, 1, 0),
    ("res2.1x1a'", 8, 256, 256, 256, 64, 1, 1, 0),
    ("res3.3x3", 8, 128, 128, 128, 128, 3, 1, 1),
    ("res3.1x1c", 8, 128, 128, 128, 512, 1, 1, 0),
    ("res4.3x3", 8, 64, 64, 256, 256, 3, 1, 1),
    ("res4.1x1c", 8, 64, 64, 256, 1024, 1, 1, 0),
    ("res5.3x3", 8, 32, 32, 512, 512, 3, 1, 1),
    ("res5.1x1c", 8, 32, 32, 512, 2048, 1, 1, 0),
    ("fpn.out.p2", 8, 256, 256, 256, 256, 3, 1, 1),
    ("fpn.lat.p2", 8, 256, 256, 256, 256, 1, 1, 0),
    ("rpn.pred.p2", 8, 256, 256, 256, 15, 1, 1, 0),
    ("mask.3x3(1600)", 1600, 14, 14, 256, 256, 3, 1, 1),
    ("fc1(8000)", 1, 1, 8000, 12544, 1024, 1, 1, 0),
]

def main():
    ctx = ops.torch_context(0)
    only = os.environ.get("AMP_ONLY")
    if os.environ.get("AMP_BN256"):
        from ampis_amd import _lib
        _lib.lib().amp_debug_set_f16x3_bn256(int(os.environ["AMP_BN256"]))
    d = "cuda:0"
    out = []
    for shp in SHAPES:
        name, B, H, W, Cin, Cout, k, s, p = shp[:9]
        if only and not any(name.startswith(o) for o in only.split(',')):
            continue
        mode = shp[9] if len(shp) > 9 else ""
        kh, kw = (k if isinstance(k, tuple) else (k, k))
        x = torch.randn(B, H, W, Cin, device=d)
        w = torch.randn(Cout, kh, kw, Cin, device=d) * 0.05
        sc = torch.ones(Cout, device=d); sh = torch.zeros(Cout, device=d)
        Ho = (H + 2 * p - kh) // s + 1; Wo = (W + 2 * p - kw) // s + 1
        flops = 2.0 * B * Ho * Wo * Cout * kh * kw * Cin
        kw_args = dict(stride=s, pad=p, relu=True)
        if mode == "res":
            kw_args.update(res=torch.randn(B, Ho, Wo, Cout, device=d), res_mode=1)
        elif mode == "up":
            kw_args.update(res=torch.randn(B, Ho // 2, Wo // 2, Cout, device=d), res_mode=2)
        elif mode == "deconv":
            kw_args.update(deconv2x2=True)
        if os.environ.get("AMP_SPLIT_IN") and Cin % 32 == 0 and mode != "deconv" and Cout % 32 == 0:
            # the trunk-native data path: input (and residual) already in the split row format, output written in it
            x = ops.split_rows(ctx, x)
            fmt = ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT
            if "res" in kw_args:
                kw_args["res"] = ops.split_rows(ctx, kw_args["res"])
                fmt |= ops.FMT_RES_SPLIT
            kw_args["fmt"] = fmt
            name = name + " [split]"
        if os.environ.get("AMP_SPLIT_RING") is not None:
            from ampis_amd import _lib
            _lib.lib().amp_debug_set_split_ring(int(os.environ["AMP_SPLIT_RING"]))
        for _ in range(2):
            ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw_args)
        torch.cuda.synchronize()
        n = 5
        ctx.timer_start()
        for _ in range(n):
            ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw_args)
        ms = ctx.timer_stop() / n
        tf = flops / ms / 1e9
        out.append((name, ms, tf))
        print(f"{name:18s} {ms:8.3f} ms  {tf:7.1f} TFLOP/s  (M={B*Ho*Wo}, N={Cout}, K={kh*kw*Cin})", flush=True)
        del x, w
    tot_ms = sum(o[1] for o in out)
    print(json.dumps({"total_ms": tot_ms}))

if __name__ == "__main__":
    main()
