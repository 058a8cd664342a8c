"""Parity of the fp32-MFMA implicit-GEMM convolution against torch-CPU conv2d (the op the oracle is built on)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# fp32 MFMA is an exact-f32 fma chain; torch CPU sums in another order -> tolerance scales with sum |a*b|.
RTOL = 2e-5


def _ref_conv(x, w, scale, shift, res, stride, pad, relu, res_mode):
    y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), stride=stride, padding=pad)
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.view(1, -1, 1, 1)
    if res is not None:
        r = res.permute(0, 3, 1, 2)
        if res_mode == 2:
            r = F.interpolate(r, scale_factor=2, mode="nearest")
        y = y + r
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 1).contiguous()


CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, relu, res_mode
    (2, 16, 16, 64, 64, 1, 1, 0, True, 0),
    (2, 16, 16, 64, 64, 3, 1, 1, True, 0),
    (1, 24, 20, 64, 256, 1, 1, 0, True, 1),      # bottleneck conv3 + shortcut + relu
    (2, 16, 16, 256, 128, 1, 2, 0, False, 0),    # strided 1x1 (stride in 1x1)
    (1, 14, 14, 256, 256, 3, 1, 1, True, 0),     # mask head conv, M tail (196 rows)
    (1, 16, 16, 512, 256, 1, 1, 0, False, 2),    # FPN lateral + nearest-up2 top-down
    (3, 9, 11, 32, 15, 1, 1, 0, False, 0),       # tiny Cout (RPN predictors), ragged M
    (1, 32, 32, 4, 64, 7, 2, 3, True, 0),        # stem: Cin padded 3->4
    (1, 1, 200, 12544, 1024, 1, 1, 0, True, 0),  # fc1 as a 1x1 conv over RoIs
]


@pytest.mark.parametrize("case", CASES)
def test_conv_matches_torch_cpu(gpu_ctx, case):
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, stride, pad, relu, res_mode = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, k, k, Cin, generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g)
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (W + 2 * pad - k) // stride + 1
    res = None
    if res_mode == 1:
        res = torch.randn(B, Ho, Wo, Cout, generator=g)
    elif res_mode == 2:
        res = torch.randn(B, Ho // 2, Wo // 2, Cout, generator=g)
    ref = _ref_conv(x, w, scale, shift, res, stride, pad, relu, res_mode)
    dev = "cuda:0"
    y = ops.conv2d_nhwc(gpu_ctx, x.to(dev), w.to(dev), scale.to(dev), shift.to(dev),
                        None if res is None else res.to(dev), stride=stride, pad=pad, relu=relu,
                        res_mode=res_mode)
    torch.cuda.synchronize()
    y = y.cpu()
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    mag = ref.abs().max().item()
    assert err <= RTOL * max(mag, 1.0) * max(1.0, (k * k * Cin / 256) ** 0.5), f"max err {err} (mag {mag})"


def test_stem_kw8_zero_tap(gpu_ctx):
    """The stem runs as KH=7, KW=8 with a zero 8th tap and Cin 3->4; must equal the true 7x7x3 conv."""
    from ampis_amd import ops
    g = torch.Generator().manual_seed(7)
    x3 = torch.randn(1, 64, 64, 3, generator=g)
    w3 = torch.randn(64, 7, 7, 3, generator=g) * 0.1
    ref = F.conv2d(x3.permute(0, 3, 1, 2), w3.permute(0, 3, 1, 2), stride=2, padding=3).permute(0, 2, 3, 1)
    x4 = torch.zeros(1, 64, 64, 4); x4[..., :3] = x3
    w4 = torch.zeros(64, 7, 8, 4); w4[:, :, :7, :3] = w3
    d = "cuda:0"
    # KW=8 with pad 3 would give Wo = (64+6-8)//2+1 = 32 = the 7-wide conv's Wo: same output grid.
    from ampis_amd._lib import ConvDesc, check, lib, ptr
    import ctypes as C
    xd, wd = x4.to(d), w4.to(d)
    y = torch.empty(1, 32, 32, 64, device=d)
    # call through ops with a non-square window: emulate via direct desc
    desc = ConvDesc(1, 64, 64, 4, 64, 7, 8, 2, 3, 0, 0, 0)
    # Ho computed by the library from KH, Wo from KW: (64+6-7)//2+1 = 32, (64+6-8)//2+1 = 32
    check(lib().amp_conv2d_nhwc(gpu_ctx.handle, C.byref(desc), ptr(xd), ptr(wd), None, None, None, ptr(y)))
    torch.cuda.synchronize()
    assert (y.cpu() - ref).abs().max().item() < 1e-4


def test_deconv2x2_scatter(gpu_ctx):
    from ampis_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 14, 14, 256, generator=g)
    wt = torch.randn(256, 64, 2, 2, generator=g) * 0.05      # torch ConvTranspose2d layout [Cin, Cout, kH, kW]
    bias = torch.randn(64, generator=g)
    ref = F.relu(F.conv_transpose2d(x.permute(0, 3, 1, 2), wt, bias, stride=2)).permute(0, 2, 3, 1)
    # our layout: [Cout_total = (ky,kx,co)][1][1][Cin]
    w = wt.permute(2, 3, 1, 0).reshape(4 * 64, 1, 1, 256).contiguous()
    shift = bias.repeat(4)
    d = "cuda:0"
    y = ops.conv2d_nhwc(gpu_ctx, x.to(d), w.to(d), None, shift.to(d), relu=True, deconv2x2=True)
    torch.cuda.synchronize()
    assert y.shape == (3, 28, 28, 64)
    assert (y.cpu() - ref).abs().max().item() < 1e-4


@pytest.mark.parametrize("mode", ["plain", "res", "up", "deconv", "s2scatter_mask", "ragged"])
def test_fast_epilogue_equals_generic_bitwise(gpu_ctx, mode):
    """The straight-line epilogue (Cout % 4 == 0) and the generic one do the same fp32 operations: outputs must be identical."""
    import ctypes as C
    from ampis_amd._lib import ConvDesc, check, lib, ptr
    g = torch.Generator().manual_seed(3)
    d = "cuda:0"
    B, H, W, Cin, Cout = (2, 18, 22, 64, 192) if mode != "ragged" else (1, 13, 7, 32, 72)
    x = torch.randn(B, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, 1, 1, Cin, generator=g) * 0.1).to(d)
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).to(d), torch.randn(Cout, generator=g).to(d)
    res = mask = None
    res_mode = out_mode = 0
    yshape = (B, H, W, Cout)
    if mode in ("res", "ragged"):
        res, res_mode = torch.randn(B, H, W, Cout, generator=g).to(d), 1
    elif mode == "up":
        res, res_mode = torch.randn(B, H // 2, W // 2, Cout, generator=g).to(d), 2
    elif mode == "deconv":
        out_mode, yshape = 1, (B, 2 * H, 2 * W, Cout // 4)
    elif mode == "s2scatter_mask":
        out_mode, yshape = 2, (B, 2 * H, 2 * W, Cout)
        mask = torch.randn(*yshape, generator=g).to(d)
    desc = ConvDesc(B, H, W, Cin, Cout, 1, 1, 1, 0, 1, res_mode, out_mode)
    outs = []
    for generic in (0, 1):
        lib().amp_debug_set_conv_generic_epilogue(generic)
        y = torch.full(yshape, 7.0, device=d)
        try:
            check(lib().amp_conv2d_nhwc_ex(gpu_ctx.handle, C.byref(desc), ptr(x), ptr(w), ptr(sc), ptr(sh),
                                           None if res is None else ptr(res), None if mask is None else ptr(mask), ptr(y)))
            torch.cuda.synchronize()
        finally:
            lib().amp_debug_set_conv_generic_epilogue(0)
        outs.append(y.cpu())
    assert torch.equal(outs[0], outs[1])
    assert outs[0].abs().max() > 0


@pytest.mark.parametrize("case", [(2, 20, 24, 256, 32, 1), (1, 17, 19, 512, 32, 2), (1, 12, 12, 1024, 32, 2), (1, 8, 8, 2048, 32, 1), (2, 9, 9, 128, 2, 1)])
def test_grouped_conv3x3_matches_torch(gpu_ctx, case):
    """ResNeXt conv2 (groups=32, stride 1 or 2 in the 3x3) against torch conv2d(groups=...)."""
    from ampis_amd import ops
    B, H, W, Cw, groups, stride = case
    g = torch.Generator().manual_seed(Cw + stride)
    cpg = Cw // groups
    x = torch.randn(B, H, W, Cw, generator=g)
    w = torch.randn(Cw, 3, 3, cpg, generator=g) * (2.0 / (9 * cpg)) ** 0.5
    scale, shift = torch.rand(Cw, generator=g) + 0.5, torch.randn(Cw, generator=g)
    ref = F.conv2d(x.permute(0, 3, 1, 2), w.permute(0, 3, 1, 2), stride=stride, padding=1, groups=groups)
    ref = F.relu(ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    d = "cuda:0"
    y = ops.conv2d_grouped_nhwc(gpu_ctx, x.to(d), w.to(d), groups, scale.to(d), shift.to(d), stride=stride, pad=1, relu=True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert (y.cpu() - ref).abs().max().item() <= RTOL * max(ref.abs().max().item(), 1.0)
