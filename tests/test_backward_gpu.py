"""Backward building blocks against torch-CPU autograd of the same convolution: weight gradient
(split-K MFMA kernel + fixed-order reduce), data gradient (= the forward conv kernel on flipped / transposed / scaled weights,
with stride-2 scatter and ReLU-mask epilogues), bias gradient (column sums)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 20, 24, 128, 128, 3, 1, 1),
    (2, 16, 16, 256, 512, 1, 1, 0),
    (2, 16, 16, 256, 128, 1, 2, 0),       # stride in the 1x1 (first block of a stage)
    (3, 14, 14, 256, 256, 3, 1, 1),       # mask head: M tail, Wo < 32
    (1, 1, 300, 1024, 12, 1, 1, 0),       # fused box predictor (padded to 12 outputs)
    (2, 28, 28, 256, 256, 2, 2, 0),       # ConvTranspose 2x2 s2 seen as a conv (deconv backward)
    (1, 64, 64, 256, 256, 3, 1, 1),
]


def _ref(x, w, dy, stride, pad):
    xt = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    wt = w.permute(0, 3, 1, 2).clone().requires_grad_(True)
    y = F.conv2d(xt, wt, stride=stride, padding=pad)
    y.backward(dy.permute(0, 3, 1, 2))
    return xt.grad.permute(0, 2, 3, 1).contiguous(), wt.grad.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("case", CASES)
def test_wgrad_and_dgrad(gpu_ctx, case):
    from ampis_amd import ops
    B, H, W, Cin, Cout, k, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(B, H, W, Cin, generator=g)
    w = torch.randn(Cout, k, k, Cin, generator=g) * 0.1
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = torch.randn(B, Ho, Wo, Cout, generator=g)
    scale = torch.rand(Cout, generator=g) + 0.5
    dx_ref, dw_ref = _ref(x, w, dy * scale, stride, pad)      # forward was y = conv * scale -> gradients see dy * scale
    xd, wd, dyd, sd = x.to(DEV), w.to(DEV), dy.to(DEV), scale.to(DEV)
    # ---- weight gradient ----
    dw = ops.conv2d_wgrad(gpu_ctx, xd, dyd, w.shape, stride=stride, pad=pad, scale=sd)
    torch.cuda.synchronize()
    err = (dw.cpu() - dw_ref).abs().max().item()
    assert err <= 3e-5 * max(1.0, dw_ref.abs().max().item()) * max(1.0, (B * Ho * Wo / 256) ** 0.5), err
    dw2 = ops.conv2d_wgrad(gpu_ctx, xd, dyd, w.shape, stride=stride, pad=pad, scale=sd, grad=dw.clone())
    torch.cuda.synchronize()
    assert torch.allclose(dw2.cpu(), 2 * dw.cpu(), rtol=1e-6, atol=1e-6)       # accumulate mode
    # bitwise reproducible
    dw3 = ops.conv2d_wgrad(gpu_ctx, xd, dyd, w.shape, stride=stride, pad=pad, scale=sd)
    torch.cuda.synchronize()
    assert torch.equal(dw3, dw)
    # ---- data gradient ----
    if Cout % 4 == 0 and stride in (1, 2) and (stride == 1 or k in (1, 2)):
        wt = ops.dgrad_weights(gpu_ctx, wd, sd)
        if stride == 1:
            dx = ops.conv2d_nhwc(gpu_ctx, dyd, wt, stride=1, pad=k - 1 - pad)
        elif k == 1:
            dx = ops.conv2d_nhwc(gpu_ctx, dyd, wt, stride=1, pad=0, scatter2=True)
        else:
            return   # 2x2 s2: its data gradient is the deconv forward (tested in test_conv_gpu)
        torch.cuda.synchronize()
        assert dx.shape == dx_ref.shape
        err = (dx.cpu() - dx_ref).abs().max().item()
        assert err <= 3e-5 * max(1.0, dx_ref.abs().max().item()) * max(1.0, (k * k * Cout / 256) ** 0.5), err


def test_dgrad_mask_and_residual(gpu_ctx):
    """d(x) = (dgrad(conv1 path) + shortcut gradient) * (x > 0): the bottleneck's input gradient in one launch."""
    from ampis_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W, C, Cm = 2, 12, 12, 256, 128
    dy = torch.randn(B, H, W, Cm, generator=g)
    w = torch.randn(Cm, 1, 1, C, generator=g) * 0.1
    res = torch.randn(B, H, W, C, generator=g)
    xfwd = torch.randn(B, H, W, C, generator=g)
    ref = (torch.einsum("bhwn,nc->bhwc", dy, w[:, 0, 0, :]) + res) * (xfwd > 0)
    wt = ops.dgrad_weights(gpu_ctx, w.to(DEV))
    dx = ops.conv2d_nhwc(gpu_ctx, dy.to(DEV), wt, res=res.to(DEV), mask=xfwd.to(DEV))
    torch.cuda.synchronize()
    assert (dx.cpu() - ref).abs().max().item() < 1e-4


def test_colsum(gpu_ctx):
    from ampis_amd import ops
    g = torch.Generator().manual_seed(6)
    dy = torch.randn(10000, 16, generator=g)
    out = ops.colsum(gpu_ctx, dy.to(DEV))
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), dy.double().sum(0).float(), rtol=1e-5, atol=1e-3)
