"""amp_sigmoid_focal_loss (north_star: "focal/smooth-L1/BCE losses are hand-written HIP kernels") against the published formula in
torch fp32 on the CPU (fvcore sigmoid_focal_loss restated below) and torch autograd for the gradient.  The reference's Mask R-CNN path
does not use it: there is no reference behaviour beyond the formula.  Tolerance: fp32, 2e-6 relative on the sum,
1e-6 absolute + 1e-5 relative on gradients (expf / log1pf / powf of the device against torch's)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def focal_reference(x, t, alpha, gamma):
    p = torch.sigmoid(x)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(x, t, reduction="none")
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return loss.sum()


@pytest.mark.parametrize("alpha,gamma,K", [(0.25, 2.0, 1), (0.25, 2.0, 7), (-1.0, 0.0, 3), (0.5, 1.5, 80), (0.25, 1.0, 2)])
def test_focal_loss_and_gradient(gpu_ctx, alpha, gamma, K):
    from ampis_amd._lib import check, lib, ptr
    g = torch.Generator().manual_seed(int(K * 10 + gamma))
    N = 30011
    x = torch.randn(N, K, generator=g) * 3.0
    x[:50] = torch.tensor([-40.0, 40.0, 0.0, 1e-4, -1e-4]).repeat(10)[:, None]            # saturated and near-zero logits
    lab = torch.randint(0, K + 1, (N,), generator=g, dtype=torch.int32)                     # K = background
    lab[::13] = -1                                                                          # ignored rows
    scale = 1.0 / 517.0
    d = "cuda:0"
    xd, ld = x.to(d), lab.to(d)
    dx = torch.full((N, K), 7.0, device=d)
    part = torch.zeros(2048, device=d)
    loss = torch.zeros(1, device=d)
    for rep in range(2):
        check(lib().amp_sigmoid_focal_loss(gpu_ctx.handle, N, K, ptr(xd), ptr(ld), alpha, gamma, scale, ptr(dx), ptr(part), 2048, ptr(loss)), "amp_sigmoid_focal_loss")
        torch.cuda.synchronize()
        if rep == 0:
            first = (loss.clone(), dx.clone())
    assert torch.equal(first[0], loss) and torch.equal(first[1], dx)                        # fixed summation order
    valid = lab >= 0
    xr = x[valid].clone().requires_grad_(True)
    t = torch.nn.functional.one_hot(lab[valid].long(), K + 1)[:, :K].to(torch.float32)
    ref = focal_reference(xr, t, alpha, gamma) * scale
    ref.backward()
    assert float(loss.item()) == pytest.approx(float(ref), rel=2e-6, abs=1e-9)
    got = dx.cpu()
    assert torch.all(got[~valid] == 0)
    assert torch.allclose(got[valid], xr.grad, rtol=1e-5, atol=1e-6 * scale * 10)
    # loss only (no gradient buffer)
    check(lib().amp_sigmoid_focal_loss(gpu_ctx.handle, N, K, ptr(xd), ptr(ld), alpha, gamma, scale, None, ptr(part), 2048, ptr(loss)), "amp_sigmoid_focal_loss")
    torch.cuda.synchronize()
    assert torch.equal(first[0], loss)
