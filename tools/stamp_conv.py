"""Lab: in-kernel phase timing of conv_split_kernel (library built with `make EXTRA=-DAMP_STAMP`, see conv.hip g_stamp).
usage (GPU box): AMP_STAGGER=0|1 python tools/stamp_conv.py [layer ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops, _lib

LAYERS = {"res4.3x3": (8, 64, 64, 256, 256, 3, 1), "fpn.out.p2": (8, 256, 256, 256, 256, 3, 1), "fc1": (1, 1, 8000, 12544, 1024, 1, 0),
          "fpn.lat.p2": (8, 256, 256, 256, 256, 1, 0), "res4.1x1c": (8, 64, 64, 256, 1024, 1, 0),
          "res2.1x1c": (8, 256, 256, 64, 256, 1, 0), "deconv.gemm": (1, 560, 560, 256, 1024, 1, 0), "res3.1x1c": (8, 128, 128, 128, 512, 1, 0), "res4.3x3": (8, 64, 64, 256, 256, 3, 1)}

def main():
    ctx = ops.torch_context(0)
    L = _lib.lib()
    buf = (ctypes.c_ulonglong * 64)()
    for name in (sys.argv[1:] or list(LAYERS)):
        B, H, W, Cin, Cout, k, p = LAYERS[name]
        x = ops.split_rows(ctx, torch.randn(B, H, W, Cin, device="cuda:0"))
        w = torch.randn(Cout, k, k, Cin, device="cuda:0") * 0.05
        sc = torch.ones(Cout, device="cuda:0"); sh = torch.zeros(Cout, device="cuda:0")
        kw = dict(stride=1, pad=p, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
        if os.environ.get("AMP_STAMP_RES"):      # with a split residual (the trunk's conv3)
            kw.update(res=ops.split_rows(ctx, torch.randn(B, H, W, Cout, device="cuda:0")), res_mode=1, fmt=kw["fmt"] | ops.FMT_RES_SPLIT)
            name = name + "+res"
        for _ in range(2):
            ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw)
        torch.cuda.synchronize()
        assert L.amp_debug_read_stamps(buf) == 0
        if os.environ.get("AMP_STAMP_CLOCK"):      # the clock the chip holds in the K loop: >= 2 s of back-to-back launches on random data first
            import time
            clk = (ctypes.c_ulonglong * 2)()
            t0 = time.time()
            while time.time() - t0 < 2.5:
                for _ in range(50):
                    ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw)
                torch.cuda.synchronize()
            assert L.amp_debug_read_stamp_clock(clk) == 0
            for _ in range(200):
                ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw)
            torch.cuda.synchronize()
            assert L.amp_debug_read_stamp_clock(clk) == 0
            print(f"{name}: in-loop clock {clk[0] / max(clk[1], 1) * 0.1:.3f} GHz (s_memtime / s_memrealtime over the K loop, 200 launches after 2.5 s of warm-up)")
            assert L.amp_debug_read_stamps(buf) == 0
        n = 4
        ctx.timer_start()
        for _ in range(n):
            ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw)
        ms = ctx.timer_stop() / n
        torch.cuda.synchronize()
        assert L.amp_debug_read_stamps(buf) == 0
        M = B * H * W
        nsteps = k * k * Cin // 32
        nwg = ((M + 127) // 128) * (Cout // 256)
        if os.environ.get("AMP_STAMP_TILES") == "patch":      # conv3x3_patch_kernel: 8 x 16 pixel tiles
            nwg = B * ((H + 7) // 8) * ((W + 15) // 16) * (Cout // 256)
        per = nwg * n * nsteps               # (wave slot, step) samples behind every counter
        print(f"{name}: {ms:.3f} ms, {2.0 * M * Cout * k * k * Cin / ms / 1e9:.0f} TFLOP/s, {nsteps} steps, {nwg} workgroups; cycles per step and wave:")
        print("   wave   wait+barrier   dma-issue   frag-reads   mfma-issue   | sum    loop/steps")
        for wv in range(8):
            v = [buf[wv * 8 + q] / per for q in range(5)]
            pro, epi = buf[wv * 8 + 5] / (nwg * n), buf[wv * 8 + 6] / (nwg * n)
            if os.environ.get("AMP_STAMP_TILES") == "patch":      # conv3x3_patch_kernel re-uses slots 6 / 7: the wait of the steps at taps 3..5, the vmcnt part of the wait
                print(f"   {wv}      {v[0]:9.0f}   {v[1]:9.0f}   {v[2]:9.0f}   {v[3]:9.0f}      | {sum(v[:4]):6.0f}  {v[4]:6.0f}   per workgroup: prologue {pro:7.0f}  loop {v[4] * nsteps:8.0f}  "
                      f"wait per step at taps 3..5 {buf[wv * 8 + 6] / (nwg * n) / (nsteps / 3):6.0f}, vmcnt part of the wait per step {buf[wv * 8 + 7] / per:6.0f}")
                continue
            print(f"   {wv}      {v[0]:9.0f}   {v[1]:9.0f}   {v[2]:9.0f}   {v[3]:9.0f}      | {sum(v[:4]):6.0f}  {v[4]:6.0f}   per workgroup: prologue {pro:7.0f}  loop {v[4] * nsteps:8.0f}  epilogue {epi:7.0f} (of which final barrier {buf[wv * 8 + 7] / (nwg * n):6.0f})")

main()
