"""The two convolution arithmetics of the C ABI (amp_set_conv_mode): AMP_CONV_F32 (fp32 MFMA) and AMP_CONV_F16X3 (operands split
into two f16 halves, three exact-product f16 MFMAs, fp32 accumulation).  The split mode has to be an fp32-equivalent: its error
against an fp64 reference must not exceed the fp32-MFMA kernel's, and operands outside its range must be detected, not silently
mis-computed."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 16, 64, 64, 1, 1, 0),
    (2, 33, 29, 256, 256, 3, 1, 1),
    (1, 14, 14, 256, 256, 3, 1, 1),
    (2, 16, 16, 256, 128, 1, 2, 0),
    (1, 1, 200, 12544, 1024, 1, 1, 0),
    (3, 9, 11, 32, 16, 1, 1, 0),
]


def _run(ctx, mode, x, w, sc, sh, s, p, res=None):
    from ampis_amd import ops
    ctx.conv_mode = mode
    try:
        y = ops.conv2d_nhwc(ctx, x, w, sc, sh, res, stride=s, pad=p, relu=True)
        torch.cuda.synchronize()
    finally:
        ctx.conv_mode = "f16x3"
    return y.cpu().double()


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("xscale", [1.0, 300.0, 1e-3])
def test_f16x3_is_at_least_as_accurate_as_fp32_mfma(gpu_ctx, case, xscale):
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(Cin + k)
    x = torch.randn(B, H, W, Cin, generator=g) * xscale
    w = torch.randn(Cout, k, k, Cin, generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * xscale
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), stride=s, padding=p)
    ref = F.relu(ref * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    d = "cuda:0"
    args = (x.to(d), w.to(d), sc.to(d), sh.to(d), s, p)
    y32, y16 = _run(gpu_ctx, "f32", *args), _run(gpu_ctx, "f16x3", *args)
    assert not gpu_ctx.conv_range_flag()
    mag = ref.abs().max().item()
    e32, e16 = (y32 - ref).abs().max().item() / mag, (y16 - ref).abs().max().item() / mag
    r32, r16 = (y32 - ref).pow(2).mean().sqrt().item() / mag, (y16 - ref).pow(2).mean().sqrt().item() / mag
    assert e16 <= max(1.5 * e32, 3e-7), (e16, e32)
    assert r16 <= max(1.25 * r32, 3e-8), (r16, r32)


# ---- the proved bound, on adversarial operands ------------------------------------------------------------------------------
# Worst-case error of one output element, S = sum_i |x_i||w_i| over its receptive field, K products, u = 2^-24:
#   fp32 MFMA (v_mfma_f32_32x32x2_f32: exact products, K/2 dependent accumulations):      |err| <= (K/2 + 4) u S
#   f16x3 (x = hi + lo'/2^11, |x - x~| <= max(2^-23 |x|, 2^-36): hi and lo' are f16, a subnormal lo' keeps an ABSOLUTE step of
#          2^-25 / 2^11; three exact-product MFMAs, the dropped lo*lo term < 2^-22 |x w|; K/16 dependent accumulations of
#          16-product sums):   |err| <= (2^-21 + (K/8 + 6) u) S + 2^-36 (sum_i |w_i| + sum_i |x_i|)
# For every layer of the network (K >= 64) the f16x3 bound is the smaller one: 2^-21 + K/8 u <= K/2 u  <=>  K >= 22.
# The absolute term is the UNDERFLOW statement: values below 2^-13 are represented with an absolute error of 2^-36 instead of a
# relative one, which is harmless next to operands of ordinary size (the term is 2^-36 sum|w|) and costs relative accuracy only
# when a WHOLE receptive field is that small; forward activations of a FrozenBN network are O(1) and the loss gradients, which
# are not, go through the 2^16 shift (amp_conv2d_wgrad_scaled, conv_run in_shift) -- tests below and in test_wgrad_*.
U = 2.0 ** -24


def _gamma32(K):
    return (K / 2 + 4) * U


def _gamma16(K):
    return 2.0 ** -21 + (K / 8 + 6) * U


def _adversarial(kind, shape_x, shape_w, g):
    B, H, W, Cin = shape_x
    if kind == "cancel":          # neighbouring channels carry +v and -(1 + 2^-10) v against equal weights: sums cancel to 1e-3 of S
        x = torch.randn(B, H, W, Cin // 2, generator=g).abs() + 0.5
        x = torch.stack([x, -x * (1 + 2.0 ** -10)], dim=-1).reshape(B, H, W, Cin)
        w = torch.randn(shape_w[0], shape_w[1], shape_w[2], Cin // 2, generator=g).abs() * 0.05 + 0.01
        w = torch.stack([w, w], dim=-1).reshape(shape_w)
    elif kind == "loguniform":    # magnitudes spread over 12 decades, random signs
        x = 10.0 ** (torch.rand(shape_x, generator=g) * 12 - 8) * torch.sign(torch.randn(shape_x, generator=g))
        w = 10.0 ** (torch.rand(shape_w, generator=g) * 4 - 4) * torch.sign(torch.randn(shape_w, generator=g))
    elif kind == "tiny":          # every activation far below the f16 normal range
        x = torch.randn(shape_x, generator=g) * 1e-6
        w = torch.randn(shape_w, generator=g) * 0.05
    elif kind == "heavy_tail":    # Cauchy-like tails, clipped inside the f16 range
        x = (torch.randn(shape_x, generator=g) / torch.randn(shape_x, generator=g).abs().clamp_min(1e-3)).clamp(-6e4, 6e4)
        w = torch.randn(shape_w, generator=g) * 0.05
    else:
        raise ValueError(kind)
    return x.float().contiguous(), w.float().contiguous()


ADV_CASES = [(2, 16, 16, 64, 64, 1, 1, 0), (2, 19, 23, 256, 256, 3, 1, 1), (1, 1, 96, 12544, 1024, 1, 1, 0), (2, 16, 16, 256, 128, 1, 2, 0)]


@pytest.mark.parametrize("case", ADV_CASES)
@pytest.mark.parametrize("kind", ["cancel", "loguniform", "tiny", "heavy_tail"])
def test_both_arithmetics_hold_their_proved_bound_on_adversarial_operands(gpu_ctx, case, kind):
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(1000 + Cin + k)
    x, w = _adversarial(kind, (B, H, W, Cin), (Cout, k, k, Cin), g)
    K = k * k * Cin
    assert _gamma16(K) <= _gamma32(K)
    nchw = lambda t: t.double().permute(0, 3, 1, 2)
    conv = lambda a, b: F.conv2d(nchw(a), nchw(b), stride=s, padding=p).permute(0, 2, 3, 1)
    ref = conv(x, w)
    S = conv(x.abs(), w.abs())
    sum_w = conv(torch.ones_like(x), w.abs())          # sum of |w| over the taps that fall inside the image
    sum_x = conv(x.abs(), torch.ones_like(w))
    d = "cuda:0"
    gpu_ctx.conv_range_flag()
    y32 = _run_linear(gpu_ctx, "f32", x.to(d), w.to(d), s, p)
    y16 = _run_linear(gpu_ctx, "f16x3", x.to(d), w.to(d), s, p)
    assert not gpu_ctx.conv_range_flag()
    e32, e16 = (y32 - ref).abs(), (y16 - ref).abs()
    b32 = _gamma32(K) * S + 1e-45
    b16 = _gamma16(K) * S + 2.0 ** -36 * (sum_w + sum_x) + 1e-45
    assert bool((e32 <= b32).all()), (kind, float((e32 / b32).max()))
    assert bool((e16 <= b16).all()), (kind, float((e16 / b16).max()))
    # and in practice both sit far inside it; f16x3 is not the looser of the two by more than the operand-rounding term
    r32, r16 = float((e32 / b32).max()), float((e16 / b16).max())
    print(f"{kind} K={K}: worst err / bound  fp32 {r32:.3f}  f16x3 {r16:.3f}")


def _run_linear(ctx, mode, x, w, s, p):
    from ampis_amd import ops
    ctx.conv_mode = mode
    try:
        y = ops.conv2d_nhwc(ctx, x, w, None, None, None, stride=s, pad=p, relu=False)
        torch.cuda.synchronize()
    finally:
        ctx.conv_mode = "f16x3"
    return y.cpu().double()


def test_out_of_range_operand_raises_the_flag(gpu_ctx):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 8, 8, 64, generator=g)
    x[0, 3, 3, 5] = 1.0e5                       # beyond fp16: the split halves cannot hold it
    w = torch.randn(64, 1, 1, 64, generator=g) * 0.1
    d = "cuda:0"
    gpu_ctx.conv_range_flag()
    _run(gpu_ctx, "f16x3", x.to(d), w.to(d), None, None, 1, 0)
    assert gpu_ctx.conv_range_flag(), "an activation above 65504 must be reported"
    assert not gpu_ctx.conv_range_flag(), "reading with clear=True resets the flag"
    y = _run(gpu_ctx, "f32", x.to(d), w.to(d), None, None, 1, 0)
    assert torch.isfinite(y).all() and not gpu_ctx.conv_range_flag()


WGRAD_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, dy magnitude, dy_shift
    (2, 20, 24, 128, 128, 3, 1, 1, 1.0, 0),
    (2, 16, 16, 256, 128, 1, 2, 0, 1.0, 0),
    (3, 14, 14, 256, 256, 3, 1, 1, 1e-6, 16),     # loss-gradient magnitudes: need the shift
    (1, 1, 300, 1024, 12, 1, 1, 0, 1e-4, 16),
    (2, 37, 41, 256, 64, 3, 1, 1, 1e-8, 16),      # ragged pixel count, N tile half empty
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_reads_split_activations_bit_identically(gpu_ctx, case):
    """amp_conv2d_wgrad_fmt(x_split = 1): the activation stored in the trunk's split row format gives the SAME weight gradient, bit for
    bit, as the fp32 activation split inside the kernel (the halves are the same numbers)."""
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, s, p, mag, shift = case
    if Cin % 32:
        pytest.skip("split rows need Cin % 32 == 0")
    g = torch.Generator().manual_seed(3 * Cin + k + Cout)
    x = (torch.randn(B, H, W, Cin, generator=g) * 3).clamp_(min=0).cuda()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = (torch.randn(B, Ho, Wo, Cout, generator=g) * mag).cuda()
    want = ops.conv2d_wgrad(gpu_ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift)
    xs = ops.split_rows(gpu_ctx, x)
    got = ops.conv2d_wgrad(gpu_ctx, xs, dy, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift, x_split=True)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert not gpu_ctx.conv_range_flag()


WGRAD_SPLIT_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, dy magnitude
    (2, 20, 24, 256, 128, 3, 1, 1, 1e-5),        # 3x3: nine taps through the offset table, image borders
    (3, 17, 19, 512, 256, 1, 1, 0, 1e-6),        # ragged pixel count (969 rows), two K' tiles
    (2, 32, 32, 128, 128, 3, 1, 1, 1e-4),        # Cin = 128: a K' tile straddles two taps, nine taps = 4.5 tiles
    (2, 32, 32, 256, 512, 1, 2, 0, 1e-5),        # stride 2 (the first conv of a stage)
    (1, 1, 300, 1024, 128, 1, 1, 0, 1e-3),       # fc-like
    (1, 1, 32, 256, 128, 1, 1, 0, 1e-3),         # one K-step: the table / staging prologue without a loop body
    (1, 1, 64, 256, 128, 1, 1, 0, 1e-3),         # two K-steps
    (1, 1, 96, 256, 128, 1, 1, 0, 1e-3),         # three: the first step that loads a table row set inside the loop's guard
    (1, 1, 130, 256, 256, 1, 1, 0, 1e-3),        # five, the last one ragged
    (1, 6, 7, 256, 128, 3, 1, 1, 1e-3),          # 42 pixels, nine taps: two K-steps per tap tile with out-of-image rows
]


@pytest.mark.parametrize("case", WGRAD_SPLIT_CASES)
def test_wgrad_ring_kernel_on_split_operands(gpu_ctx, case):
    """amp_conv2d_wgrad_fmt(x_split = 3): activations AND scaled loss gradients in the split row format go through wgrad_split_kernel
    (LDS-DMA ring + transposing LDS reads).  Same operands as the in-kernel split, another summation order: equal to the fp32-storage
    result within accumulation noise, as accurate against fp64, bitwise reproducible."""
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, s, p, mag = case
    g = torch.Generator().manual_seed(5 * Cin + k + Cout)
    x = (torch.randn(B, H, W, Cin, generator=g) * 2).clamp_(min=0).cuda()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = (torch.randn(B, Ho, Wo, Cout, generator=g) * mag).cuda()
    ref = torch.nn.grad.conv2d_weight(x.cpu().double().permute(0, 3, 1, 2), (Cout, Cin, k, k), dy.cpu().double().permute(0, 3, 1, 2),
                                      stride=s, padding=p).permute(0, 2, 3, 1)
    plain = ops.conv2d_wgrad(gpu_ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=16)
    xs = ops.split_rows(gpu_ctx, x)
    dys = ops.split_rows(gpu_ctx, dy * 65536.0)
    got = ops.conv2d_wgrad(gpu_ctx, xs, dys, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=16, x_split=3)
    again = ops.conv2d_wgrad(gpu_ctx, xs, dys, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=16, x_split=3)
    torch.cuda.synchronize()
    assert not gpu_ctx.conv_range_flag()
    assert torch.equal(got, again)
    m = ref.abs().max().item()
    e_plain = (plain.cpu().double() - ref).abs().max().item() / m
    e_got = (got.cpu().double() - ref).abs().max().item() / m
    assert e_got <= max(1.5 * e_plain, 3e-7), (e_got, e_plain)
    assert (got - plain).abs().max().item() / m < 2e-6


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_fused_bias_gradient(gpu_ctx, case):
    """amp_conv2d_wgrad_fmt(bias_grad): the column sums of dy the MFMA kernel adds up on the side equal an fp64 sum to fp32 accuracy,
    accumulate when asked, repeat bit for bit, and leave the weight gradient untouched."""
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, s, p, mag, shift = case
    g = torch.Generator().manual_seed(11 * Cin + k + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).clamp_(min=0).cuda()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = (torch.randn(B, Ho, Wo, Cout, generator=g) * mag + 0.1 * mag).cuda()
    want_w = ops.conv2d_wgrad(gpu_ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift)
    b = torch.full((Cout,), 7.0, device="cuda:0")
    got_w = ops.conv2d_wgrad(gpu_ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift, bias_grad=b)
    torch.cuda.synchronize()
    assert torch.equal(got_w, want_w)
    ref = dy.double().sum((0, 1, 2))
    err = (b.double() - ref).abs().max().item()
    assert err <= 2e-6 * dy.double().abs().sum((0, 1, 2)).max().item() + 1e-30, err
    b2 = b.clone()
    ops.conv2d_wgrad(gpu_ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift, bias_grad=b2, bias_accumulate=True)
    b3 = torch.zeros_like(b)
    ops.conv2d_wgrad(gpu_ctx, x, dy, (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift, bias_grad=b3)
    torch.cuda.synchronize()
    assert torch.equal(b3, b), "bitwise reproducible"
    assert torch.equal(b2, b + b)


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_f16x3_is_at_least_as_accurate_as_fp32_mfma(gpu_ctx, case):
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, s, p, mag, shift = case
    g = torch.Generator().manual_seed(Cin + 7 * k + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).clamp_(min=0)          # post-ReLU activations
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(B, Ho, Wo, Cout, generator=g) * mag
    xt = x.double().permute(0, 3, 1, 2)
    ref = torch.nn.grad.conv2d_weight(xt, (Cout, Cin, k, k), dy.double().permute(0, 3, 1, 2), stride=s, padding=p).permute(0, 2, 3, 1)
    d = "cuda:0"
    out = {}
    for mode in ("f32", "f16x3"):
        gpu_ctx.conv_mode = mode
        try:
            out[mode] = ops.conv2d_wgrad(gpu_ctx, x.to(d), dy.to(d), (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift)
            torch.cuda.synchronize()
            again = ops.conv2d_wgrad(gpu_ctx, x.to(d), dy.to(d), (Cout, k, k, Cin), stride=s, pad=p, dy_shift=shift)
            torch.cuda.synchronize()
            assert torch.equal(again, out[mode]), "weight gradients are bitwise reproducible in both modes"
        finally:
            gpu_ctx.conv_mode = "f16x3"
        out[mode] = out[mode].cpu().double()
    assert not gpu_ctx.conv_range_flag()
    m = ref.abs().max().item()
    e32, e16 = (out["f32"] - ref).abs().max().item() / m, (out["f16x3"] - ref).abs().max().item() / m
    r32, r16 = (out["f32"] - ref).pow(2).mean().sqrt().item() / m, (out["f16x3"] - ref).pow(2).mean().sqrt().item() / m
    assert e16 <= max(1.5 * e32, 3e-7), (e16, e32)
    assert r16 <= max(1.25 * r32, 3e-8), (r16, r32)


def test_wgrad_x_shift_and_range_flag(gpu_ctx):
    """The deconv weight gradient has the loss gradient as its 'x' operand (x_shift); an operand that leaves fp16 after the
    shift must raise the flag."""
    from ampis_amd import ops
    g = torch.Generator().manual_seed(11)
    B, H, W, Cin, Cout = 2, 28, 28, 256, 256
    x = torch.randn(B, H, W, Cin, generator=g) * 1e-6
    dy = torch.randn(B, 14, 14, Cout, generator=g).clamp_(min=0)
    ref = torch.nn.grad.conv2d_weight(x.double().permute(0, 3, 1, 2), (Cout, Cin, 2, 2), dy.double().permute(0, 3, 1, 2), stride=2).permute(0, 2, 3, 1)
    d = "cuda:0"
    gpu_ctx.conv_range_flag()
    got = ops.conv2d_wgrad(gpu_ctx, x.to(d), dy.to(d), (Cout, 2, 2, Cin), stride=2, pad=0, x_shift=16).cpu().double()
    assert not gpu_ctx.conv_range_flag()
    assert (got - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    dy_big = torch.randn(B, 14, 14, Cout, generator=g) * 4.0
    ops.conv2d_wgrad(gpu_ctx, x.to(d), dy_big.to(d), (Cout, 2, 2, Cin), stride=2, pad=0, dy_shift=16)
    torch.cuda.synchronize()
    assert gpu_ctx.conv_range_flag(), "dy * 2^16 beyond 65504 must be reported"


def test_mode_switch_and_model_fallback(gpu_ctx):
    """A model whose activations leave the fp16 range must return exactly what the fp32-MFMA mode returns (the batch is re-run)."""
    from ampis_amd import params as P
    from ampis_amd.model import MaskRCNN
    from test_e2e_gpu import synth_image
    assert gpu_ctx.conv_mode == gpu_ctx.CONV_F16X3
    K, H, W = 2, 128, 160
    imgs = np.stack([synth_image(np.random.default_rng(2), H, W)])
    p = P.init_params(K, seed=3, style="spread")
    p["backbone.bottom_up.stem.conv1.norm.weight"] = p["backbone.bottom_up.stem.conv1.norm.weight"] * np.float32(4000.0)   # res2 inputs ~1e5
    m = MaskRCNN(gpu_ctx, K, max_batch=1, max_h=H, max_w=W, max_out_hw=W, detections_per_image=20)
    m.load_params(p)
    a = m.infer(imgs, rle="counts")
    assert m.tap("res2").max() > 65504
    gpu_ctx.conv_mode = "f32"
    try:
        b = m.infer(imgs, rle="counts")
    finally:
        gpu_ctx.conv_mode = "f16x3"
    for x, y in zip(a, b):
        assert np.array_equal(x["boxes"], y["boxes"]) and np.array_equal(x["scores"], y["scores"])
        assert all(np.array_equal(q["counts"], r["counts"]) for q, r in zip(x["masks"], y["masks"]))
    m.close()


def test_training_trajectories_agree_between_modes(gpu_ctx):
    """Eight SGD steps from the same weights, batches and sampling seeds: (a) everything on the fp32 MFMA, (b) forward + data-gradient
    convolutions in AMP_CONV_F16X3, (c) fp32 MFMA again with the stem weights perturbed by 1e-7 relative.  Detection training
    amplifies any rounding difference (NMS / sampling decisions flip), so (b) cannot track (a) forever -- but it must not drift
    faster than the fp32 run does under a one-ulp-sized perturbation: the split arithmetic is an fp32 equivalent in training too."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    K, B, S, steps = 2, 2, 256, 8
    imgs, gts = synth.batch(B, S, S, first_index=300)
    p0 = P.init_params(K, seed=2, style="spread")
    traj = {}
    for tag, mode, eps in (("f32", "f32", 0.0), ("f16x3", "f16x3", 0.0), ("f32_perturbed", "f32", 1e-7)):
        gpu_ctx.conv_mode = mode
        try:
            p = dict(p0)
            if eps:
                p["backbone.bottom_up.stem.conv1.weight"] = p0["backbone.bottom_up.stem.conv1.weight"] * np.float32(1 + eps)
            m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=B * 700, max_poly_doubles=B * 700 * 64)
            m.load_params(p)
            losses = []
            for i in range(steps):
                L = m.forward_losses(imgs, gts, seed=100 + i, backward=True)
                m.sgd_step(0.002)
                losses.append([L[k] for k in sorted(L)])
            traj[tag] = np.array(losses)
            m.close()
        finally:
            gpu_ctx.conv_mode = "f16x3"
    a = traj["f32"]
    assert all(np.isfinite(t).all() for t in traj.values())
    drift = lambda t: (np.abs(a - t) / np.maximum(np.abs(a), 1e-3)).max(axis=1)
    d16, dper = drift(traj["f16x3"]), drift(traj["f32_perturbed"])
    # measured drift per step: f16x3 2e-7 1e-5 4e-5 2e-4 9e-4 3e-3 | perturbed fp32 3e-7 4e-6 7e-6 4e-5 1e-4 3e-3 (the fp32 run
    # itself is not bitwise repeatable: RoIAlign backward uses float atomics), so the envelopes are loose by ~10x
    assert d16[0] < 1e-5, d16[0]                                   # same forward to fp32 rounding
    assert d16[1] < 2e-4 and d16[2] < 5e-4 and d16[3] < 3e-3, d16
    assert d16[:5].max() <= 100 * max(dper[:5].max(), 1e-5), (d16, dper)
    assert abs(traj["f16x3"][-1].sum() - a[-1].sum()) < 0.05 * a[-1].sum()


def test_native_split_trunk_against_fp32_activations(gpu_ctx):
    """AMP_CONV_F16X3 inference keeps every trunk and head activation in the split row format (model.hip run_trunk); with the
    switch off the same kernels exchange fp32 tensors and split them in the consumer.  Conv inputs are identical either way (the
    stage test below proves the convolutions bit-identical); what differs is that residual / FPN top-down adds and RoIAlign read the
    stored value hi + lo' * 2^-11 instead of the fp32 one: a 2^-23 relative rounding of the stored activation.  So: taps agree to a
    few 2^-23 per residual block, final detections agree within the end-to-end gate's box / score tolerances."""
    from ampis_amd import params as P
    from ampis_amd._lib import lib
    from ampis_amd.model import MaskRCNN
    from test_e2e_gpu import synth_image
    K, B, H, W = 2, 2, 224, 288
    rng = np.random.default_rng(9)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=W, detections_per_image=80)
    m.load_params(P.init_params(K, seed=3, style="spread"))
    outs, taps = [], []
    try:
        for on in (1, 0):
            lib().amp_debug_set_split_chain(on)
            outs.append(m.infer(imgs, rle="counts"))
            taps.append({k: m.tap(k) for k in ("stem_pool", "res2", "res3", "res4", "res5", "p2", "p5", "box_pooled")})
    finally:
        lib().amp_debug_set_split_chain(-1)
    m.close()
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    assert rel(taps[0]["stem_pool"], taps[1]["stem_pool"]) <= 2.0 ** -22          # the same values, stored with 22-23 bits
    for i, k in enumerate(("res2", "res3", "res4", "res5")):
        assert rel(taps[0][k], taps[1][k]) <= (i + 2) * 4 * 2.0 ** -22, (k, rel(taps[0][k], taps[1][k]))
    for k in ("p2", "p5", "box_pooled"):
        assert rel(taps[0][k], taps[1][k]) <= 2e-5, (k, rel(taps[0][k], taps[1][k]))
    for x, y in zip(*outs):
        assert len(x["boxes"]) > 10 and len(x["boxes"]) == len(y["boxes"])
        assert np.abs(x["boxes"] - y["boxes"]).max() < 1e-3 and np.abs(x["scores"] - y["scores"]).max() < 1e-5
        assert np.array_equal(x["classes"], y["classes"])


SPLIT_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, res
    (2, 40, 48, 256, 256, 3, 1, 1, 0),      # 128x256 ring kernel, 72 K-steps
    (2, 40, 48, 256, 256, 1, 1, 0, 1),      # 8 K-steps, split residual
    (3, 33, 29, 128, 128, 3, 1, 1, 0),      # 256x128 ring kernel, ragged M
    (2, 32, 32, 256, 512, 1, 2, 0, 0),      # strided 1x1 (res3.0 shortcut shape)
    (1, 1, 700, 1024, 1024, 1, 1, 0, 0),    # fc-shaped
    (2, 16, 16, 64, 64, 3, 1, 1, 0),        # narrow: 2-buffer kernel
    (2, 24, 24, 32, 256, 1, 1, 0, 2),       # ONE K-step, upsampled split residual (FPN top-down)
    (2, 24, 24, 64, 256, 1, 1, 0, 1),       # two K-steps
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_split_format_is_a_data_path_not_an_arithmetic(gpu_ctx, case):
    """The trunk-native activation format: input, residual and output in the split hi|lo' row format (amp_conv2d_nhwc_fmt).  The
    convolution on split operands must equal, bit for bit, the convolution that receives the fp32 tensor and splits it in the
    kernel (and adds the decoded residual), with its output then split: same halves, same products, same accumulation order --
    only who splits, and when, differs."""
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, s, p, res_mode = case
    g = torch.Generator().manual_seed(31 * Cin + Cout + k)
    d = "cuda:0"
    x = (torch.randn(B, H, W, Cin, generator=g) * 3).to(d)
    w = (torch.randn(Cout, k, k, Cin, generator=g) * (2.0 / (k * k * Cin)) ** 0.5).to(d)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(d), torch.randn(Cout, generator=g).to(d)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = None
    if res_mode == 1:
        res = torch.randn(B, Ho, Wo, Cout, generator=g).to(d)
    elif res_mode == 2:
        res = torch.randn(B, Ho // 2, Wo // 2, Cout, generator=g).to(d)
    xs = ops.split_rows(gpu_ctx, x)                     # what a producer's epilogue writes for the fp32 value x
    x_dec = ops.unsplit_rows(gpu_ctx, xs)
    assert (x_dec - x).abs().max().item() <= 2.0 ** -22 * x.abs().max().item()
    assert torch.equal(ops.unsplit_rows(gpu_ctx, ops.split_rows(gpu_ctx, x_dec)), x_dec), "a decoded value survives the format unchanged"
    rs = r_dec = None
    if res is not None:
        rs = ops.split_rows(gpu_ctx, res)
        r_dec = ops.unsplit_rows(gpu_ctx, rs)           # the residual the split path adds (exactly: hi + lo' * 2^-11)
    kw = dict(stride=s, pad=p, relu=True, res_mode=res_mode if res_mode else None)
    legacy = ops.conv2d_nhwc(gpu_ctx, x, w, sc, sh, r_dec, **kw)     # splits x in the kernel: the same (hi, lo') the producer would have stored
    fmt = ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT | (ops.FMT_RES_SPLIT if res is not None else 0)
    for ring in (1, 0):
        _lib_set_ring(ring)
        try:
            y_split = ops.conv2d_nhwc(gpu_ctx, xs, w, sc, sh, rs, fmt=fmt, **kw)
            y_plain = ops.conv2d_nhwc(gpu_ctx, xs, w, sc, sh, rs, fmt=fmt & ~ops.FMT_Y_SPLIT, **kw)
        finally:
            _lib_set_ring(1)
        torch.cuda.synchronize()
        assert torch.equal(y_plain, legacy), f"ring={ring}: fp32 output differs from the in-kernel-split convolution"
        assert torch.equal(y_split, ops.split_rows(gpu_ctx, legacy)), f"ring={ring}: split output differs"
    assert not gpu_ctx.conv_range_flag()


def _lib_set_ring(v):
    from ampis_amd import _lib
    _lib.lib().amp_debug_set_split_ring(int(v))


@pytest.mark.parametrize("case", [(2, 20, 24, 256, 32, 1), (1, 17, 19, 512, 32, 2), (1, 12, 12, 1024, 32, 2), (2, 9, 9, 128, 2, 1)])
def test_grouped_conv_reads_and_writes_the_split_format(gpu_ctx, case):
    """ResNeXt conv2 on a split-format input (amp_conv2d_grouped_nhwc_fmt): the 64-channel window of an N tile is the same 256 B of a
    split row; the result equals the grouped convolution that splits the fp32 tensor in the kernel, bit for bit, as fp32 and as split output."""
    from ampis_amd import ops
    B, H, W, Cw, groups, stride = case
    g = torch.Generator().manual_seed(7 * Cw + stride)
    cpg = Cw // groups
    d = "cuda:0"
    x = (torch.randn(B, H, W, Cw, generator=g) * 2).to(d)
    w = (torch.randn(Cw, 3, 3, cpg, generator=g) * (2.0 / (9 * cpg)) ** 0.5).to(d)
    sc, sh = (torch.rand(Cw, generator=g) + 0.5).to(d), torch.randn(Cw, generator=g).to(d)
    xs = ops.split_rows(gpu_ctx, x)                     # what the producer (conv1's epilogue) writes for the fp32 value x
    legacy = ops.conv2d_grouped_nhwc(gpu_ctx, x, w, groups, sc, sh, stride=stride, pad=1, relu=True)      # splits x in the kernel: the same (hi, lo')
    y_plain = ops.conv2d_grouped_nhwc(gpu_ctx, xs, w, groups, sc, sh, stride=stride, pad=1, relu=True, fmt=ops.FMT_X_SPLIT)
    y_split = ops.conv2d_grouped_nhwc(gpu_ctx, xs, w, groups, sc, sh, stride=stride, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
    torch.cuda.synchronize()
    assert legacy.abs().max() > 0
    assert torch.equal(y_plain, legacy)
    assert torch.equal(y_split, ops.split_rows(gpu_ctx, legacy))
    assert not gpu_ctx.conv_range_flag()


@pytest.mark.parametrize("C,P", [(256, 7), (256, 14), (64, 7)])
def test_roi_align_reads_and_writes_the_split_format(gpu_ctx, C, P):
    """RoIAlign on split-row feature maps (the trunk's native format) == RoIAlign on the decoded fp32 maps, bit for bit, with fp32 and
    with split output; C = 256 takes the two-bins-per-wave kernel, other widths the generic one."""
    from ampis_amd import ops
    g = torch.Generator().manual_seed(C + P)
    d = "cuda:0"
    B, H, W = 2, 96, 128
    feats = [(torch.randn(B, H // s, W // s, C, generator=g) * 2).to(d) for s in (1, 2, 4, 8)]     # strides 4, 8, 16, 32
    R = 301
    ctr = torch.rand(R, 2, generator=g) * torch.tensor([4.0 * W, 4.0 * H])
    size = torch.exp(torch.rand(R, 2, generator=g) * 6.0 + 1.0)           # 3 .. 1100 px: every level, bins inside and outside the image
    rois = torch.cat([ctr - size / 2, ctr + size / 2], 1).float().to(d)
    bidx = (torch.arange(R) % B).int().to(d)
    fs = [ops.split_rows(gpu_ctx, f) for f in feats]
    fd = [ops.unsplit_rows(gpu_ctx, f) for f in fs]
    from ampis_amd import _lib
    _lib.lib().amp_debug_set_roi_lanes(0)              # the reference kernel on the decoded maps
    try:
        ref, lv0 = ops.roi_align(gpu_ctx, fd, rois, bidx, P)
    finally:
        _lib.lib().amp_debug_set_roi_lanes(1)
    got, lv1 = ops.roi_align(gpu_ctx, fs, rois, bidx, P, fmt=ops.FMT_X_SPLIT)
    torch.cuda.synchronize()
    assert torch.equal(lv0, lv1) and len(set(lv0.tolist())) == 4
    assert torch.equal(got, ref)
    if C % 32 == 0:
        got_s, _ = ops.roi_align(gpu_ctx, fs, rois, bidx, P, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
        assert torch.equal(got_s, ops.split_rows(gpu_ctx, ref))


@pytest.mark.parametrize("shape", [(8, 64, 64, 256, 256, 3, 1, False), (8, 64, 64, 256, 1024, 1, 0, True), (3, 96, 97, 128, 128, 3, 1, False),
                                   (8, 128, 128, 64, 256, 1, 0, True)])
def test_ring_kernel_repeats_bit_for_bit(gpu_ctx, shape):
    """Race screen of the LDS-DMA ring / ping-pong kernel (tools/race_screen.py is the long version with a noisy neighbour): the
    arithmetic is deterministic, so a tile read before its DMA landed or restaged before its last reader shows as a launch that does
    not reproduce the first one."""
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, p, res = shape
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = ops.split_rows(gpu_ctx, torch.randn(B, H, W, Cin, generator=g).cuda())
    w = (torch.randn(Cout, k, k, Cin, generator=g) * 0.05).cuda()
    sc, sh = torch.rand(Cout, generator=g).cuda() + 0.5, torch.randn(Cout, generator=g).cuda()
    kw = dict(stride=1, pad=p, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
    if res:
        kw.update(res=ops.split_rows(gpu_ctx, torch.randn(B, H, W, Cout, generator=g).cuda()), res_mode=1, fmt=kw["fmt"] | ops.FMT_RES_SPLIT)
    first = ops.conv2d_nhwc(gpu_ctx, x, w, sc, sh, **kw).clone()
    for _ in range(40):
        y = ops.conv2d_nhwc(gpu_ctx, x, w, sc, sh, **kw)
        torch.cuda.synchronize()
        assert torch.equal(y.view(torch.int32), first.view(torch.int32))


@pytest.mark.parametrize("B,H,W", [(2, 224, 288), (1, 96, 352), (3, 160, 128), (1, 1024, 1024)])
def test_fused_stem_pool_is_bit_identical(gpu_ctx, B, H, W):
    """stem_pool_u8_kernel / stem_pool_f16x3_kernel (7x7 stem + ReLU + 3x3 max-pool in one kernel, the stem's output never written; the first
    also normalises and splits the uint8 pixels itself) against the kernels they replace: the pooled tensor bit for bit -- pooled sizes that are not multiples of the 8 x 7 tile, images of a batch, and the bench's
    size -- and therefore identical detections."""
    from ampis_amd import params as P
    from ampis_amd._lib import lib
    from ampis_amd.model import MaskRCNN
    from test_e2e_gpu import synth_image
    K = 2
    rng = np.random.default_rng(B * 1000 + H + W)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=40)
    m.load_params(P.init_params(K, seed=3, style="spread"))
    outs, taps = [], []
    try:
        # (fused from the uint8 image: stem_pool_u8_kernel) | (preprocess<split> + stem_pool_f16x3_kernel) | (preprocess, stem, max-pool)
        for pool_on, u8_on in ((1, 1), (1, 0), (0, 0)):
            lib().amp_debug_set_stem_pool(pool_on)
            lib().amp_debug_set_stem_u8(u8_on)
            outs.append(m.infer(imgs, rle="counts"))
            taps.append(m.tap("stem_pool"))
    finally:
        lib().amp_debug_set_stem_pool(1)
        lib().amp_debug_set_stem_u8(1)
    m.close()
    assert taps[0].shape == (B, (H // 2 + 1) // 2, (W // 2 + 1) // 2, 64) and float(np.abs(taps[0]).max()) > 0
    for k in (1, 2):
        assert np.array_equal(taps[0].view(np.uint32), taps[k].view(np.uint32)), k
        for x, y in zip(outs[0], outs[k]):
            assert np.array_equal(x["boxes"], y["boxes"]) and np.array_equal(x["scores"], y["scores"])


@pytest.mark.parametrize("shape", [(2, 256, 256, 64, 256, True, True), (3, 250, 237, 64, 256, True, False),
                                   (8, 128, 128, 32, 128, True, True), (2, 256, 256, 64, 128, True, True), (2, 128, 128, 128, 512, True, True), (4, 64, 64, 256, 1024, True, True), (4, 130, 127, 512, 256, True, False)])
def test_short_k_two_workgroup_tiles_are_bit_identical(gpu_ctx, shape):
    """K <= 512 layers with a residual (the trunk.s conv3: byte-bound) run on 128 x 128 tiles on two buffers, two workgroups per CU
    (`conv_split_kernel<128, 128, ., 2>`), instead of one 128 x 256 ring tile per CU: same products in the same order, same epilogue -- bit
    for bit with and without residual / scale / ReLU, on pixel counts that are not multiples of the tile; launches reproduce."""
    from ampis_amd import ops
    from ampis_amd._lib import lib
    B, H, W, Cin, Cout, res, relu = shape
    g = torch.Generator().manual_seed(B * H + W + Cin)
    x = ops.split_rows(gpu_ctx, torch.randn(B, H, W, Cin, generator=g).cuda())
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) * 0.1).cuda()
    sc = (torch.rand(Cout, generator=g) + 0.5).cuda() if res else None
    sh = torch.randn(Cout, generator=g).cuda()
    kw = dict(stride=1, pad=0, relu=relu, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
    if res:
        kw.update(res=ops.split_rows(gpu_ctx, torch.randn(B, H, W, Cout, generator=g).cuda()), res_mode=1, fmt=kw["fmt"] | ops.FMT_RES_SPLIT)
    try:
        lib().amp_debug_set_short_k(0)
        ref = ops.conv2d_nhwc(gpu_ctx, x, w, sc, sh, **kw).clone()
        lib().amp_debug_set_short_k(1)
        for _ in range(10):
            y = ops.conv2d_nhwc(gpu_ctx, x, w, sc, sh, **kw)
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.int32), ref.view(torch.int32))
    finally:
        lib().amp_debug_set_short_k(1)
    assert float(ops.unsplit_rows(gpu_ctx, ref).abs().max()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(5, 240, 251, 64, 3, True, False), (5, 240, 251, 256, 1, True, True), (6, 219, 230, 64, 1, False, False), (4, 256, 256, 64, 3, True, True)])
def test_tall_64_wide_tiles_are_bit_identical(gpu_ctx, shape):
    """Cout = 64 layers on pre-split input (res2's 3x3 and 1x1 convolutions) run on 256 x 64 tiles of `conv_split_kernel` (four waves of 64 x 64,
    two buffers, two workgroups per CU) instead of the 128 x 64 ring tiles of `conv_glds_kernel`: the same products in the same order through the
    same epilogue -- bit for bit, with split and fp32 output, on pixel counts that are not multiples of 256; launches reproduce."""
    from ampis_amd import ops
    from ampis_amd._lib import lib
    B, H, W, Cin, k, relu, y_split = shape
    g = torch.Generator().manual_seed(B * H + W + Cin + k)
    x = ops.split_rows(gpu_ctx, torch.randn(B, H, W, Cin, generator=g).cuda())
    w = (torch.randn(64, k, k, Cin, generator=g) * 0.1).cuda()
    sc = (torch.rand(64, generator=g) + 0.5).cuda()
    sh = torch.randn(64, generator=g).cuda()
    kw = dict(stride=1, pad=k // 2, relu=relu, fmt=ops.FMT_X_SPLIT | (ops.FMT_Y_SPLIT if y_split else 0))
    assert (B * H * W + 255) // 256 >= 1024          # the dispatch takes the tall tiles from 1024 of them
    try:
        lib().amp_debug_set_tall64(0)
        ref = ops.conv2d_nhwc(gpu_ctx, x, w, sc, sh, **kw).clone()
        lib().amp_debug_set_tall64(1)
        for _ in range(5):
            y = ops.conv2d_nhwc(gpu_ctx, x, w, sc, sh, **kw)
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.int32), ref.view(torch.int32))
    finally:
        lib().amp_debug_set_tall64(1)
    assert float((ops.unsplit_rows(gpu_ctx, ref) if y_split else ref).abs().max()) > 0


@pytest.mark.parametrize("case", ["dense64", "grouped32x8", "grouped_cpg32", "ragged"])
def test_patch_staged_3x3_over_a_64_channel_window_equals_the_implicit_gemm_kernels(gpu_ctx, case):
    """conv3x3_c64_kernel (round 4: res2's dense 64 -> 64 layers and the ResNeXt conv2: the pixel patch under an 8 x 16 output tile staged once,
    nine taps read out of it) against the implicit-GEMM kernels it replaces on the same split operands: the same exact products summed in the
    same K order (tap by tap, then the two 32-channel halves) -- bit for bit, with FrozenBN affine + ReLU, with a split ReLU mask (the data
    gradient of a training step), on image sizes that are not multiples of the 8 x 16 tile, and against an fp64 reference."""
    import torch
    from ampis_amd import _lib, ops
    torch.manual_seed(5)
    B, H, W = (2, 40, 64) if case != "ragged" else (3, 37, 51)
    C, groups = {"dense64": (64, 1), "grouped32x8": (256, 32), "grouped_cpg32": (128, 4), "ragged": (64, 1)}[case]
    cpg = C // groups
    x = torch.randn(B, H, W, C, device="cuda")
    wg = torch.randn(C, 3, 3, cpg, device="cuda") * 0.1                  # grouped OHWI
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
    xs = ops.split_rows(gpu_ctx, x)
    L = _lib.lib()
    outs = []
    for mode in (0, 2):
        L.amp_debug_set_patch_conv(mode)
        try:
            if groups == 1:
                y = ops.conv2d_nhwc(gpu_ctx, xs, wg, sc, sh, stride=1, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
            else:
                y = ops.conv2d_grouped_nhwc(gpu_ctx, xs, wg, groups, sc, sh, stride=1, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
            torch.cuda.synchronize()
        finally:
            L.amp_debug_set_patch_conv(1)
        outs.append(y)
    assert torch.equal(outs[0].view(torch.int32), outs[1].view(torch.int32)), f"{int((outs[0].view(torch.int32) != outs[1].view(torch.int32)).sum())} words differ"
    xd = ops.unsplit_rows(gpu_ctx, xs).double().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xd, wg.double().permute(0, 3, 1, 2), stride=1, padding=1, groups=groups)
    ref = torch.relu(ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).permute(0, 2, 3, 1)
    got = ops.unsplit_rows(gpu_ctx, outs[1]).double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6


@pytest.mark.parametrize("shape", [(2, 64, 128, 256), (3, 61, 75, 256), (2, 72, 96, 128), (1, 9, 17, 64), (1, 8, 16, 192)])
def test_fused_res2_block_tail_equals_the_two_launch_chain(gpu_ctx, shape):
    """amp_bottleneck64_tail (round 4: conv3x3_c64_kernel<false, true>): conv2 + FrozenBN + ReLU + conv3 + FrozenBN + shortcut + ReLU in one launch,
    conv2's output kept in registers as conv3's MFMA operand -- against the two amp_conv2d_nhwc_fmt launches on the same split operands: bit for
    bit (the intermediate is the same pair of f16 halves the chain stores and reloads), on ragged image sizes, repeatably; and against fp64."""
    import torch
    from ampis_amd import _lib, ops
    B, H, W, C3 = shape
    torch.manual_seed(B * H + W)
    x = torch.randn(B, H, W, 64, device="cuda")
    res = torch.randn(B, H, W, C3, device="cuda")
    w2 = torch.randn(64, 3, 3, 64, device="cuda") * 0.05
    w3 = torch.randn(C3, 1, 1, 64, device="cuda") * 0.1
    sc2, sh2 = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1
    sc3, sh3 = torch.rand(C3, device="cuda") + 0.5, torch.randn(C3, device="cuda") * 0.1
    xs, rs = ops.split_rows(gpu_ctx, x), ops.split_rows(gpu_ctx, res)
    S = ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT
    t2 = ops.conv2d_nhwc(gpu_ctx, xs, w2, sc2, sh2, stride=1, pad=1, relu=True, fmt=S)
    ref = ops.conv2d_nhwc(gpu_ctx, t2, w3, sc3, sh3, res=rs, stride=1, pad=0, relu=True, fmt=S | ops.FMT_RES_SPLIT)
    L = _lib.lib()
    L.amp_debug_set_patch_conv(2)           # the op-level shapes are below the 512-tile rule of the dispatch
    try:
        for _ in range(3):
            y = ops.bottleneck64_tail(gpu_ctx, xs, w2, sc2, sh2, w3, sc3, sh3, rs)
            torch.cuda.synchronize()
            ndiff = int((y.view(torch.int32) != ref.view(torch.int32)).sum())
            assert ndiff == 0, f"{ndiff} words differ"
    finally:
        L.amp_debug_set_patch_conv(1)
    xd, rd = ops.unsplit_rows(gpu_ctx, xs).double().permute(0, 3, 1, 2), ops.unsplit_rows(gpu_ctx, rs).double().permute(0, 3, 1, 2)
    a = torch.relu(torch.nn.functional.conv2d(xd, w2.double().permute(0, 3, 1, 2), padding=1) * sc2.double()[None, :, None, None] + sh2.double()[None, :, None, None])
    b = torch.nn.functional.conv2d(a, w3.double().permute(0, 3, 1, 2)) * sc3.double()[None, :, None, None] + sh3.double()[None, :, None, None]
    want = torch.relu(b + rd).permute(0, 2, 3, 1)
    got = ops.unsplit_rows(gpu_ctx, y).double()
    assert float((got - want).abs().max() / want.abs().max()) < 2e-6
    assert float(got.abs().max()) > 0


@pytest.mark.parametrize("B,H,W", [(2, 512, 640), (1, 1024, 1024)])
def test_fused_res2_blocks_leave_the_trunk_bit_identical(gpu_ctx, B, H, W):
    """The three res2 blocks with their conv2 + conv3 fused (amp_debug_set_fuse23(1), the default) against the two-launch chain (0): the res2 tap
    bit for bit and identical detections, through the model's own launch path."""
    from ampis_amd import params as P
    from ampis_amd._lib import lib
    from ampis_amd.model import MaskRCNN
    from test_e2e_gpu import synth_image
    K = 2
    rng = np.random.default_rng(B * 1000 + H + W)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=40)
    m.load_params(P.init_params(K, seed=3, style="spread"))
    outs, taps = [], []
    try:
        for on in (1, 0):
            lib().amp_debug_set_fuse23(on)
            outs.append(m.infer(imgs, rle="counts"))
            taps.append(m.tap("res2"))
    finally:
        lib().amp_debug_set_fuse23(1)
    m.close()
    assert float(np.abs(taps[0]).max()) > 0
    assert np.array_equal(taps[0].view(np.uint32), taps[1].view(np.uint32))
    for x, y in zip(outs[0], outs[1]):
        assert np.array_equal(x["boxes"], y["boxes"]) and np.array_equal(x["scores"], y["scores"])


@pytest.mark.parametrize("shape", [(8, 64, 64, 256, 256, False), (4, 127, 131, 128, 256, True), (4, 64, 64, 512, 512, False), (8, 96, 96, 64, 256, True), (1, 16, 16, 256, 256, False),
                                   (3, 37, 51, 128, 256, True)])
def test_patch_staged_wide_3x3_equals_the_channel_major_ring_kernel(gpu_ctx, shape):
    """conv3x3_patch_kernel (round 4, AMP_PATCH256=1, not the default: the FPN / RPN / res4 / res5 3x3 layers on 8 x 16 pixel tiles, the patch of a
    32-channel chunk staged once for its nine taps) against conv_split_kernel<128, 256, ., 3, CHAN = true> (amp_debug_set_korder(1)): the same exact products in the same
    channel-major order -- bit for bit, with FrozenBN affine + ReLU, with a split residual, on sizes that are not multiples of the tile, with
    two N tiles; against fp64; and close to the tap-major default (another summation order of the same products)."""
    import torch
    from ampis_amd import _lib, ops
    B, H, W, Cin, Cout, with_res = shape
    torch.manual_seed(B * H + W + Cin)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, 3, 3, Cin, device="cuda") * 0.03
    sc, sh = torch.rand(Cout, device="cuda") + 0.5, torch.randn(Cout, device="cuda") * 0.1
    xs = ops.split_rows(gpu_ctx, x)
    kw = dict(stride=1, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
    res = None
    if with_res:
        res = torch.randn(B, H, W, Cout, device="cuda")
        kw.update(res=ops.split_rows(gpu_ctx, res), res_mode=1, fmt=kw["fmt"] | ops.FMT_RES_SPLIT)
    L = _lib.lib()
    outs = {}
    try:
        for name, p256, korder in (("ring_chan", 0, 1), ("patch", 2, 0), ("ring_tap", 0, 0)):
            L.amp_debug_set_patch256(p256)
            L.amp_debug_set_korder(korder)
            outs[name] = ops.conv2d_nhwc(gpu_ctx, xs, w, sc, sh, **kw).clone()
            torch.cuda.synchronize()
        L.amp_debug_set_patch256(2)
        for _ in range(3):
            y = ops.conv2d_nhwc(gpu_ctx, xs, w, sc, sh, **kw)
            torch.cuda.synchronize()
            assert torch.equal(y.view(torch.int32), outs["patch"].view(torch.int32))
    finally:
        L.amp_debug_set_patch256(0)
        L.amp_debug_set_korder(0)
    ntiles, nsteps = (B * H * W + 127) // 128 * (Cout // 256), 9 * Cin // 32
    if ntiles >= 512 or (ntiles >= 192 and nsteps >= 64):          # the dispatch rule of the ring kernel (conv.hip wide256); smaller shapes: fp64 only
        nd = int((outs["patch"].view(torch.int32) != outs["ring_chan"].view(torch.int32)).sum())
        assert nd == 0, f"{nd} words differ from the channel-major ring kernel"
        assert not torch.equal(outs["ring_chan"].view(torch.int32), outs["ring_tap"].view(torch.int32))       # the switch did switch the order
    xd = ops.unsplit_rows(gpu_ctx, xs).double().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xd, w.double().permute(0, 3, 1, 2), padding=1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    if with_res:
        ref = ref + ops.unsplit_rows(gpu_ctx, kw["res"]).double().permute(0, 3, 1, 2)
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    got = ops.unsplit_rows(gpu_ctx, outs["patch"]).double()
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    tap = ops.unsplit_rows(gpu_ctx, outs["ring_tap"]).double()
    assert float((got - tap).abs().max() / ref.abs().max()) < 2e-6


def test_patch_staged_wide_3x3_through_the_model(gpu_ctx):
    """AMP_PATCH256=1 through the model's own launch path (FPN output convs, the RPN conv with its fused predictor tail -- conv_epilogue_rpn_rows on
    8 x 16 pixel tiles --, res4 / res5): another summation order of the same products, so the same detections with boxes within 1e-3 px and
    scores within 1e-5 of the default kernels', the RPN logits of level p2 within 2e-6 of their range."""
    from ampis_amd import params as P
    from ampis_amd._lib import lib
    from ampis_amd.model import MaskRCNN
    from test_e2e_gpu import synth_image
    B, H, W, K = 2, 512, 640, 2
    rng = np.random.default_rng(77)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=40)
    m.load_params(P.init_params(K, seed=3, style="spread"))
    outs, taps = [], []
    try:
        for on in (0, 2):
            lib().amp_debug_set_patch256(on)
            outs.append(m.infer(imgs, rle="counts"))
            taps.append({k: m.tap(k) for k in ("p2", "p5", "rpn_pred2", "rpn_pred6")})
    finally:
        lib().amp_debug_set_patch256(0)
    m.close()
    for k in ("p2", "p5", "rpn_pred2", "rpn_pred6"):
        a, b = taps[0][k], taps[1][k]
        assert a.shape == b.shape and float(np.abs(a).max()) > 0
        assert not np.array_equal(a, b)                                  # the other kernel did run
        assert float(np.abs(a - b).max()) <= 5e-6 * float(np.abs(a).max()), k
    for x, y in zip(outs[0], outs[1]):
        assert len(x["boxes"]) == len(y["boxes"]) > 0
        assert float(np.abs(x["boxes"] - y["boxes"]).max()) < 1e-3 and float(np.abs(x["scores"] - y["scores"]).max()) < 1e-5
        assert np.array_equal(x["classes"], y["classes"])


@pytest.mark.parametrize("B,H,W,dets,K", [(2, 512, 640, 40, 3), (1, 384, 384, 7, 3), (4, 512, 512, 100, 3), (1, 256, 320, 1, 80), (2, 320, 256, 33, 80)])
def test_mask_tail_with_four_taps_per_workgroup_is_bit_identical(gpu_ctx, B, H, W, dets, K):
    """mask_tail_kernel (round 4: the fused deconv + ReLU + predictor + sigmoid with the four taps of a 128-pixel block in one workgroup, the ring
    carried across the taps, the tap epilogue out of the role-swapped accumulators without LDS staging) against conv_split_kernel<128, 256, 3> + conv_epilogue_predict
    (amp_debug_set_mask_tail_loop(0)): the same sums in the same order -- every mask probability bit for bit, on RoI counts whose pixel count is not a
    multiple of the tile, repeatably; identical RLE masks."""
    from ampis_amd import params as P
    from ampis_amd._lib import lib
    from ampis_amd.model import MaskRCNN
    from test_e2e_gpu import synth_image
    rng = np.random.default_rng(B * 100 + dets)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=dets)
    m.load_params(P.init_params(K, seed=5, style="spread"))
    outs, taps = [], []
    try:
        for on in (1, 0, 1):
            lib().amp_debug_set_mask_tail_loop(on)
            outs.append(m.infer(imgs, rle="counts"))
            taps.append(m.tap("mask_prob"))
    finally:
        lib().amp_debug_set_mask_tail_loop(1)
    m.close()
    n = sum(len(o["boxes"]) for o in outs[0])
    assert n > 0 and taps[0].shape[1:] == (28, 28)
    p = taps[0][:n]
    assert float(p.min()) >= 0.0 and float(p.max()) <= 1.0 and float(p.max() - p.min()) > 0.1
    for k in (1, 2):
        assert np.array_equal(taps[0][:n].view(np.uint32), taps[k][:n].view(np.uint32)), f"{int((taps[0][:n].view(np.uint32) != taps[k][:n].view(np.uint32)).sum())} probabilities differ"
        for x, y in zip(outs[0], outs[k]):
            assert np.array_equal(x["boxes"], y["boxes"]) and len(x["masks"]) == len(y["masks"])
            assert all(np.array_equal(r["counts"], t["counts"]) for r, t in zip(x["masks"], y["masks"]))


@pytest.mark.parametrize("shape", [(8, 128, 128, 256, 512, 2, False), (8, 128, 128, 512, 1024, 2, False), (5, 120, 121, 256, 512, 1, False), (8, 64, 64, 64, 2048, 1, False),
                                   (400, 14, 14, 256, 1024, 1, True), (37, 14, 14, 256, 1024, 1, True)])
def test_n_tile_loop_for_short_k_layers_is_bit_identical(gpu_ctx, shape):
    """conv1x1_nloop_kernel (round 4, AMP_NLOOP=1, not the default: several N tiles of a 128-pixel block in one workgroup, the ring carried across them, the tile epilogue between two
    K-steps with its stores counted into the ring's vmcnt waits) against conv_split_kernel<128, 256> (amp_debug_set_nloop(0)) on the short-K 1x1 layers it
    takes -- stride-2 shortcuts, a ragged last pixel block, K = 64, the mask head's deconv with its 2x2 scatter: bit for bit, repeatably; and against fp64."""
    import torch
    from ampis_amd import _lib, ops
    B, H, W, Cin, Cout, stride, deconv = shape
    torch.manual_seed(B + H + Cin + Cout)
    x = torch.randn(B, H, W, Cin, device="cuda")
    w = torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.05
    sc = None if deconv else torch.rand(Cout, device="cuda") + 0.5
    sh = torch.randn(Cout // (4 if deconv else 1), device="cuda").repeat(4 if deconv else 1) * 0.1
    xs = ops.split_rows(gpu_ctx, x)
    kw = dict(stride=stride, pad=0, relu=deconv, deconv2x2=deconv, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)
    L = _lib.lib()
    try:
        L.amp_debug_set_nloop(0)
        ref = ops.conv2d_nhwc(gpu_ctx, xs, w, sc, sh, **kw).clone()
        L.amp_debug_set_nloop(1)
        for _ in range(4):
            y = ops.conv2d_nhwc(gpu_ctx, xs, w, sc, sh, **kw)
            torch.cuda.synchronize()
            nd = int((y.view(torch.int32) != ref.view(torch.int32)).sum())
            assert nd == 0, f"{nd} words differ"
    finally:
        L.amp_debug_set_nloop(0)      # (not the default: AMP_NLOOP=1)
    xd = ops.unsplit_rows(gpu_ctx, xs).double()[:, ::stride, ::stride]
    want = torch.einsum("bhwc,nc->bhwn", xd, w.double().view(Cout, Cin))
    if sc is not None:
        want = want * sc.double()
    want = want + sh.double()
    got = ops.unsplit_rows(gpu_ctx, y).double()
    if deconv:      # [B, H, W, (ky, kx, co)] -> [B, 2H, 2W, co]
        want = torch.relu(want).view(B, H, W, 2, 2, Cout // 4).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, Cout // 4)
    assert got.shape == want.shape
    assert float((got - want).abs().max() / want.abs().max()) < 2e-6
