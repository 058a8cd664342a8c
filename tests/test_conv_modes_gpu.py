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


def test_split_format_chains_change_no_bit(gpu_ctx):
    """In AMP_CONV_F16X3 inference the box / mask head tensors travel in the split operand format and their convs stage both
    operands by LDS-DMA.  That is a change of data path, not of arithmetic: results must be bit-identical to splitting in the kernel."""
    from ampis_amd import params as P
    from ampis_amd._lib import lib
    from ampis_amd.model import MaskRCNN
    from test_e2e_gpu import synth_image
    K, B, H, W = 2, 2, 224, 288
    rng = np.random.default_rng(9)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=W, detections_per_image=80)
    m.load_params(P.init_params(K, seed=3, style="spread"))
    outs, pooled = [], []
    try:
        for on in (1, 0):
            lib().amp_debug_set_split_chain(on)
            outs.append(m.infer(imgs, rle="counts"))
            pooled.append(m.tap("box_pooled"))
    finally:
        lib().amp_debug_set_split_chain(-1)
    m.close()
    # the pooled features read back from the split format (hi + lo' / 2^11) equal the fp32 ones to 2^-22
    assert np.abs(pooled[0] - pooled[1]).max() <= 2.0 ** -21 * np.abs(pooled[1]).max()
    for x, y in zip(*outs):
        assert len(x["boxes"]) > 10
        assert np.array_equal(x["boxes"], y["boxes"]) and np.array_equal(x["scores"], y["scores"]) and np.array_equal(x["classes"], y["classes"])
        assert all(np.array_equal(p["counts"], q["counts"]) for p, q in zip(x["masks"], y["masks"]))
