"""Race screen of the LDS-DMA ring / ping-pong kernel (conv_split_kernel): the arithmetic is deterministic, so every repetition of a
launch must reproduce the first one BIT FOR BIT; an LDS tile read before its DMA landed, or restaged before its last reader, shows up
as a rare mismatch that comes and goes with timing (cdna_hip_programming.md: "place reads by the vmcnt/barrier count, never by clean
runs" -- this is the clean-runs half of that check, the count half is in the kernel's comments).  Timing is perturbed on purpose: a
second context streams memory and runs other convs on its own HIP stream while the screened launches repeat.
usage (GPU box): python tools/race_screen.py [repetitions]"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ampis_amd import ops, _lib

SHAPES = [
    # name, B, H, W, Cin, Cout, k, pad, residual
    ("3x3 256->256 @64 (72 steps, one round)", 8, 64, 64, 256, 256, 3, 1, False),
    ("3x3 256->256 @128 (4 rounds)", 8, 128, 128, 256, 256, 3, 1, False),
    ("1x1 256->1024 @64 + residual (8 steps)", 8, 64, 64, 256, 1024, 1, 0, True),
    ("1x1 64->256 @128 + residual (2 steps)", 8, 128, 128, 64, 256, 1, 0, True),
    ("3x3 128->128 @96 (256x128 tiles, ragged M)", 3, 96, 97, 128, 128, 3, 1, False),
    ("fc 12544->1024, 999 rows (392 steps, ragged)", 1, 1, 999, 12544, 1024, 1, 0, False),
    ("1x1 32->256 @64 (one step)", 4, 64, 64, 32, 256, 1, 0, False),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ctx = ops.torch_context(0)
    stop = threading.Event()

    def noise():      # a second context on its own stream: memory streams + convs of other shapes, to move the timing around
        c2 = _lib.Context(0)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            a = torch.randn(64 << 20, device="cuda:0")
            while not stop.is_set():
                a.mul_(1.0001)
                torch.cuda.current_stream().synchronize()
        c2.close()

    th = threading.Thread(target=noise)
    th.start()
    bad = 0
    try:
        for name, B, H, W, Cin, Cout, k, p, res in SHAPES:
            g = torch.Generator().manual_seed(Cin + Cout + k)
            x = ops.split_rows(ctx, torch.randn(B, H, W, Cin, generator=g).cuda())
            w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).cuda()
            sc = torch.rand(Cout, generator=g).cuda() + 0.5
            sh = torch.randn(Cout, generator=g).cuda()
            kw = dict(stride=1, pad=p, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
            if res:
                kw.update(res=ops.split_rows(ctx, torch.randn(B, H, W, Cout, generator=g).cuda()), res_mode=1, fmt=kw["fmt"] | ops.FMT_RES_SPLIT)
            first = ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw).clone()
            torch.cuda.synchronize()
            mism = 0
            for _ in range(reps):
                y = ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw)
                torch.cuda.synchronize()
                if not torch.equal(y.view(torch.int32), first.view(torch.int32)):
                    mism += 1
            print(f"{name:52s} {reps} repetitions, {mism} mismatches", flush=True)
            bad += mism
        # the weight-gradient ring kernel (wgrad_split_kernel: LDS-DMA of both operands + transposing LDS reads), same rule
        for name, B, H, W, Cin, Cout, k, p in (("wgrad 3x3 256->256 @64", 8, 64, 64, 256, 256, 3, 1), ("wgrad 3x3 128->128 @48 (two taps per tile)", 4, 48, 49, 128, 128, 3, 1),
                                               ("wgrad 1x1 1024->256 @32", 8, 32, 32, 1024, 256, 1, 0)):
            g = torch.Generator().manual_seed(Cin + Cout + k + 1)
            xs = ops.split_rows(ctx, torch.randn(B, H, W, Cin, generator=g).clamp_(min=0).cuda())
            dys = ops.split_rows(ctx, (torch.randn(B, H, W, Cout, generator=g) * 1e-4 * 65536.0).cuda())
            first = ops.conv2d_wgrad(ctx, xs, dys, (Cout, k, k, Cin), stride=1, pad=p, dy_shift=16, x_split=3).clone()
            torch.cuda.synchronize()
            mism = 0
            for _ in range(reps):
                y = ops.conv2d_wgrad(ctx, xs, dys, (Cout, k, k, Cin), stride=1, pad=p, dy_shift=16, x_split=3)
                torch.cuda.synchronize()
                if not torch.equal(y.view(torch.int32), first.view(torch.int32)):
                    mism += 1
            print(f"{name:52s} {reps} repetitions, {mism} mismatches", flush=True)
            bad += mism
        # round 4: the fused res2 tail (conv3x3_c64_kernel<false, true>: patch + two weight buffers by LDS-DMA, conv3's weights staged into the dead patch)
        for name, B, H, W, C3 in (("res2 tail 64->64->256 @128x160", 4, 128, 160, 256), ("res2 tail, ragged 61x75, 128 outputs", 3, 61, 75, 128)):
            g = torch.Generator().manual_seed(B + H + C3)
            xs = ops.split_rows(ctx, torch.randn(B, H, W, 64, generator=g).cuda())
            rs = ops.split_rows(ctx, torch.randn(B, H, W, C3, generator=g).cuda())
            w2, w3 = (torch.randn(64, 3, 3, 64, generator=g) * 0.05).cuda(), (torch.randn(C3, 1, 1, 64, generator=g) * 0.1).cuda()
            sc2, sh2 = torch.rand(64, generator=g).cuda() + 0.5, torch.randn(64, generator=g).cuda() * 0.1
            sc3, sh3 = torch.rand(C3, generator=g).cuda() + 0.5, torch.randn(C3, generator=g).cuda() * 0.1
            _lib.lib().amp_debug_set_patch_conv(2)
            first = ops.bottleneck64_tail(ctx, xs, w2, sc2, sh2, w3, sc3, sh3, rs).clone()
            torch.cuda.synchronize()
            mism = 0
            for _ in range(reps):
                y = ops.bottleneck64_tail(ctx, xs, w2, sc2, sh2, w3, sc3, sh3, rs)
                torch.cuda.synchronize()
                mism += int(not torch.equal(y.view(torch.int32), first.view(torch.int32)))
            _lib.lib().amp_debug_set_patch_conv(1)
            print(f"{name:52s} {reps} repetitions, {mism} mismatches", flush=True)
            bad += mism
        # ... the patch-staged wide 3x3 (conv3x3_patch_kernel, AMP_PATCH256: two patch buffers + ring of three weight tiles, counted vmcnt with patch pieces in the queue)
        for name, B, H, W, Cin, Cout in (("patch256 3x3 256->256 @64", 8, 64, 64, 256, 256), ("patch256 3x3 128->256, ragged 37x51", 3, 37, 51, 128, 256), ("patch256 3x3 512->512 @32", 4, 32, 32, 512, 512)):
            g = torch.Generator().manual_seed(Cin + Cout + H)
            x = ops.split_rows(ctx, torch.randn(B, H, W, Cin, generator=g).cuda())
            w = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.03).cuda()
            sc, sh = torch.rand(Cout, generator=g).cuda() + 0.5, torch.randn(Cout, generator=g).cuda()
            kw = dict(stride=1, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
            _lib.lib().amp_debug_set_patch256(2)
            first = ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw).clone()
            torch.cuda.synchronize()
            mism = 0
            for _ in range(reps):
                y = ops.conv2d_nhwc(ctx, x, w, sc, sh, **kw)
                torch.cuda.synchronize()
                mism += int(not torch.equal(y.view(torch.int32), first.view(torch.int32)))
            _lib.lib().amp_debug_set_patch256(0)
            print(f"{name:52s} {reps} repetitions, {mism} mismatches", flush=True)
            bad += mism
        # ... and the mask head's tail through the model (mask_tail_kernel: ring carried across four taps, the halves' sums exchanged at the loop's barriers)
        import numpy as np
        from ampis_amd import params as P, synth
        from ampis_amd.model import MaskRCNN
        Bm, S = 4, 512
        imgs, _ = synth.batch(Bm, S, S, first_index=3)
        m = MaskRCNN(ctx, 2, max_batch=Bm, max_h=S, max_w=S, max_out_hw=S, detections_per_image=100)
        m.load_params(P.init_params(2, seed=0, style="spread"))
        m.infer(imgs, rle="counts")
        first_p, first_r = m.tap("mask_prob").copy(), m.tap("res2").copy()
        mism = 0
        for _ in range(max(20, reps // 4)):
            m.infer(imgs, rle="counts")
            mism += int(not (np.array_equal(m.tap("mask_prob").view(np.uint32), first_p.view(np.uint32)) and np.array_equal(m.tap("res2").view(np.uint32), first_r.view(np.uint32))))
        m.close()
        print(f"{'model: mask_prob + res2 taps, B=4 512x512':52s} {max(20, reps // 4)} repetitions, {mism} mismatches", flush=True)
        bad += mism
    finally:
        stop.set()
        th.join()
    print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} mismatching launches)")
    sys.exit(0 if bad == 0 else 1)


main()
