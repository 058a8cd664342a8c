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


def _bwd_setup(seed, sizes):
    from oracle import maskrcnn as M
    g = torch.Generator().manual_seed(seed)
    B = 2
    feats = [torch.zeros(B, h, w, 256) for h, w in sizes]
    R = 90
    img_h, img_w = sizes[0][0] * 4, sizes[0][1] * 4
    ctr = torch.rand(R, 2, generator=g) * torch.tensor([float(img_w), float(img_h)])
    size = torch.exp(torch.rand(R, 2, generator=g) * 5.5 + 1.0)
    rois = torch.cat([ctr - size / 2, ctr + size / 2], 1).float()
    rois[0] = torch.tensor([-50.0, -40.0, 30.0, 20.0])                 # sticks out top-left
    rois[1] = torch.tensor([img_w - 20.0, img_h - 30.0, img_w + 60.0, img_h + 40.0])
    rois[2] = torch.tensor([5.0, 5.0, 5.0, 25.0])                      # zero width: no gradient
    bidx = (torch.arange(R) % B).int()
    lv = M.assign_levels(rois)                                          # 0 .. 3 = p2 .. p5
    return feats, rois, bidx, lv, B


@pytest.mark.parametrize("P,sizes", [(7, [(32, 40), (16, 20), (8, 10), (4, 5)]), (14, [(30, 37), (15, 19), (8, 10), (4, 5)])])
def test_roi_align_backward_owner_computes(gpu_ctx, P, sizes):
    """amp_roi_align_bwd(_batched) -- every 4x4 tile of a gradient map summed by one wave over the RoIs in index order -- against
    torch autograd of the oracle's differentiable RoIAlign, against the float-atomic kernel it replaces, and against itself: two runs
    are bitwise identical (map sizes that are not multiples of the tile, boxes outside the map, an empty box)."""
    from ampis_amd import ops, _lib
    from oracle import train as T
    feats, rois, bidx, lv, B = _bwd_setup(3 + P, sizes)
    g = torch.Generator().manual_seed(99)
    dout = torch.randn(rois.shape[0], P, P, 256, generator=g)
    strides = (4, 8, 16, 32)
    # reference: autograd through roi_align_torch, RoI by RoI (level and image as the forward assigns them)
    ref = [torch.zeros(B, 256, h, w, requires_grad=True) for h, w in sizes]
    loss = 0
    for r in range(rois.shape[0]):
        l, b = int(lv[r]), int(bidx[r])
        out = T.roi_align_torch(ref[l][b], rois[r:r + 1], P, 1.0 / strides[l])          # [1, C, P, P]
        loss = loss + (out[0] * dout[r].permute(2, 0, 1)).sum()
    loss.backward()
    runs = {}
    for tag, atomics in (("tile", 0), ("tile2", 0), ("atomic", 1)):
        _lib.lib().amp_debug_set_roi_bwd_atomics(atomics)
        try:
            d = [torch.zeros_like(f).to(DEV) for f in feats]
            ops.roi_align_bwd(gpu_ctx, d, strides, rois.to(DEV), bidx.to(DEV), P, dout.to(DEV), B=B if tag != "tile2" else 0)
            torch.cuda.synchronize()
            runs[tag] = [x.cpu() for x in d]
        finally:
            _lib.lib().amp_debug_set_roi_bwd_atomics(0)
    for l in range(4):
        want = ref[l].grad.permute(0, 2, 3, 1)
        scale = max(float(want.abs().max()), 1e-6)
        assert float((runs["tile"][l] - want).abs().max()) <= 2e-5 * scale, l
        assert float((runs["atomic"][l] - want).abs().max()) <= 2e-5 * scale, l
        assert torch.equal(runs["tile"][l], runs["tile2"][l]), f"level {l}: not bitwise reproducible"
    assert any(float(r.abs().max()) > 0 for r in runs["tile"])


def test_training_step_is_bitwise_reproducible(gpu_ctx):
    """With the RoIAlign backward free of atomics every kernel of a training step sums in a fixed order: the same batch, weights and
    seed give the same losses and bit-identical gradients for EVERY trainable tensor, run after run."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    K, B, H, W = 2, 2, 192, 256
    imgs, gts = synth.batch(B, H, W, seed=9)
    gts = [dict(boxes=g["boxes"][:40], classes=g["classes"][:40], polygons=g["polygons"][:40]) for g in gts]
    npp = P.init_params(K, seed=2, style="spread")
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=2048, max_poly_doubles=2048 * 64)
    m.load_params(npp)
    names = [k for k in npp if ".norm." not in k and not k.startswith(("backbone.bottom_up.stem", "backbone.bottom_up.res2"))]
    grads = []
    for rep in range(3):
        L = m.forward_losses(imgs, gts, seed=3, backward=True)
        ptr, n = m.grad_arena()
        arena = np.empty(n, dtype=np.float32)
        gpu_ctx.sync()
        gpu_ctx.d2h(arena, ptr)
        grads.append((L, arena))
    m.close()
    for L, a in grads[1:]:
        assert L == grads[0][0]
        assert np.array_equal(a, grads[0][1]), f"{int((a != grads[0][1]).sum())} of {a.size} gradient values differ between runs"
    assert np.abs(grads[0][1]).max() > 0 and len(names) > 50


def test_fullsize_step_f16x3_storage_paths_agree_with_fp32_mode():
    """A training step at a size where every split-storage path is live (B = 4 at 1024^2: p2 has 262 144 rows -- the bias-sum pass converts dY,
    FPN / RPN / mask-head weight gradients run on wgrad_split_kernel, their data gradients stage the same copy, the backbone chain runs on
    scaled split gradients) against the same step with every convolution on the fp32 MFMA and fp32 storage throughout: same losses, same
    gradients within the tolerance the small-size test holds against autograd."""
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    ctx = _lib.Context(0)
    K, B, S = 2, 4, 1024
    imgs, gts = synth.batch(B, S, S, first_index=300)
    gts = [dict(boxes=g["boxes"][:150], classes=g["classes"][:150], polygons=g["polygons"][:150]) for g in gts]
    npp = P.init_params(K, seed=0, style="spread")
    m = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=B * 200, max_poly_doubles=B * 200 * 64)
    m.load_params(npp)
    names = [k for k in npp if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2")]
    out = {}
    for mode in ("f32", "f16x3"):
        ctx.conv_mode = mode
        try:
            losses = m.forward_losses(imgs, gts, seed=11, backward=True)
            out[mode] = (losses, {k: m.get_tensor(k, grad=True) for k in names})
        finally:
            ctx.conv_mode = "f16x3"
    assert not ctx.conv_range_flag()
    for k, v in out["f32"][0].items():
        assert out["f16x3"][0][k] == pytest.approx(v, rel=2e-4, abs=1e-6), k
    bad = []
    for k in names:
        r, g = out["f32"][1][k], out["f16x3"][1][k]
        err = float(np.abs(g - r).max()) / max(float(np.abs(r).max()), 1e-8)
        if err > 2e-3:
            bad.append((err, k))
    assert not bad, sorted(bad, reverse=True)[:8]
    again = m.forward_losses(imgs, gts, seed=11, backward=True)          # and the step repeats bit for bit with all of them live
    assert again == out["f16x3"][0]
    for k in names:
        assert np.array_equal(m.get_tensor(k, grad=True), out["f16x3"][1][k]), k
    m.close()
    ctx.close()


def test_configs2_real_size_step_b16_1024_all_ground_truth():
    """BASELINE configs[2] at its REAL size -- local batch 16, 1024 x 1024, every ground-truth instance of the synthetic micrographs
    (particles + satellites, several hundred per image) -- not only inside bench.py: the five losses against the training oracle
    (forward only: a minute of host time), the step bitwise repeatable (losses and every value of the gradient arena), and the fp32-MFMA
    mode agreeing on the losses."""
    import hashlib
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    ctx = _lib.Context(0)
    K, B, S = 2, 16, 1024
    imgs, gts = synth.batch(B, S, S, first_index=1000)
    ngt = [len(g["boxes"]) for g in gts]
    assert min(ngt) >= 150 and sum(ngt) / B >= 250, ngt
    npp = P.init_params(K, seed=0, style="spread")
    m = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=B * 800, max_poly_doubles=B * 800 * 64)
    m.load_params(npp)
    runs = []
    for rep in range(2):
        L = m.forward_losses(imgs, gts, seed=21, backward=True)
        ptr, n = m.grad_arena()
        arena = np.empty(n, dtype=np.float32)
        ctx.sync(); ctx.d2h(arena, ptr)
        runs.append((L, hashlib.sha256(arena.tobytes()).hexdigest(), float(np.abs(arena).max())))
    assert runs[0][0] == runs[1][0] and runs[0][1] == runs[1][1] and runs[0][2] > 0 and np.isfinite(runs[0][2])
    assert not ctx.conv_range_flag()
    ctx.conv_mode = "f32"
    try:
        L32 = m.forward_losses(imgs, gts, seed=21)
    finally:
        ctx.conv_mode = "f16x3"
    m.close(); ctx.close()
    for k, v in L32.items():
        assert runs[0][0][k] == pytest.approx(v, rel=3e-4, abs=1e-6), k
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        ref = T.forward_losses(imgs, gts, M.to_torch_params(npp), T.TrainCfg(num_classes=K, seed=21))
    print("configs[2] losses:", runs[0][0], "oracle:", {k: round(float(v), 6) for k, v in ref.items()}, "GT per image:", ngt)
    for k, v in ref.items():
        assert runs[0][0][k] == pytest.approx(float(v), rel=5e-4, abs=1e-6), (k, runs[0][0][k], float(v))


@pytest.mark.parametrize("shape,scaled", [((256, 3, 3, 128), True), ((1024, 1, 1, 256), False), ((96, 3, 3, 36), True), ((64, 1, 1, 12544), True)])
def test_dgrad_weights_split_is_transpose_then_split(gpu_ctx, shape, scaled):
    """One pass (LDS-tiled where both channel counts are multiples of 64, pairwise otherwise) against the two passes it replaces in a
    training step -- amp_dgrad_weights then amp_split_weights: the same bytes."""
    from ampis_amd import ops
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    w = (torch.randn(shape, generator=g) * 0.05).cuda()
    scale = (torch.rand(shape[0], generator=g) + 0.5).cuda() if scaled else None
    two = ops.split_rows(gpu_ctx, ops.dgrad_weights(gpu_ctx, w, scale))
    one = ops.dgrad_weights_split(gpu_ctx, w, scale)
    torch.cuda.synchronize()
    assert torch.equal(one.view(torch.int32), two.view(torch.int32))


def test_deconv_scatter_written_as_split_rows_is_the_fp32_scatter_split(gpu_ctx):
    """conv_epilogue_direct with out_mode 1 (round 4): the ConvTranspose 2x2 s2 of the mask head (a 1x1 conv to 4 x 256 channels whose rows are
    scattered to the doubled resolution) written as split rows straight from the ring kernel's accumulators == the same deconv's fp32 output,
    split afterwards, bit for bit (bias, ReLU; 1568 RoIs so that the ring kernel takes it, a row count that is not a multiple of the tile)."""
    import torch
    from ampis_amd import ops
    torch.manual_seed(3)
    N = 1568 + 3
    x = torch.randn(N, 14, 14, 256, device="cuda")
    w = torch.randn(1024, 1, 1, 256, device="cuda") * 0.05
    b = torch.randn(1024, device="cuda")
    xs = ops.split_rows(gpu_ctx, x)
    y32 = ops.conv2d_nhwc(gpu_ctx, xs, w, None, b, relu=True, deconv2x2=True, fmt=ops.FMT_X_SPLIT)                       # fp32 [N,28,28,256], staged epilogue
    ysp = ops.conv2d_nhwc(gpu_ctx, xs, w, None, b, relu=True, deconv2x2=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT)     # split rows, direct epilogue
    torch.cuda.synchronize()
    assert y32.shape == ysp.shape == (N, 28, 28, 256)
    want = ops.split_rows(gpu_ctx, y32)
    assert torch.equal(ysp.view(torch.int32), want.view(torch.int32))
    ref = torch.relu(torch.einsum("nhwc,oc->nhwo", ops.unsplit_rows(gpu_ctx, xs).double(), w.view(1024, 256).double()) + b.double())
    ref = ref.view(N, 14, 14, 2, 2, 256).permute(0, 1, 3, 2, 4, 5).reshape(N, 28, 28, 256)
    assert float((y32.double() - ref).abs().max() / ref.abs().max()) < 2e-6


def test_mask_tail_on_presplit_operands_agrees_with_the_fp32_storage_path():
    """The mask head's tail of a training step with fcn4's output and d(deconv out) kept as split rows (deconv forward on the ring kernel, the
    predictor's data gradient writing d * 2^16 as split rows with the deconv's bias sums on the side, the deconv's weight- and data-gradient
    launches staging both operands as they are) against the same step with fp32 storage there (amp_debug_set_mask_tail_split(0)): identical
    losses (the forward values do not depend on the storage), the same gradients up to summation order -- 512 foreground RoIs, so the path is live."""
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    ctx = _lib.Context(0)
    K, B, S = 2, 4, 1024
    imgs, gts = synth.batch(B, S, S, first_index=300)
    gts = [dict(boxes=g["boxes"][:150], classes=g["classes"][:150], polygons=g["polygons"][:150]) for g in gts]
    npp = P.init_params(K, seed=0, style="spread")
    m = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=B * 200, max_poly_doubles=B * 200 * 64)
    m.load_params(npp)
    names = [k for k in npp if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2")]
    out = {}
    try:
        for on in (0, 1):
            _lib.lib().amp_debug_set_mask_tail_split(on)
            losses = m.forward_losses(imgs, gts, seed=11, backward=True)
            out[on] = (losses, {k: m.get_tensor(k, grad=True) for k in names})
    finally:
        _lib.lib().amp_debug_set_mask_tail_split(-1)
    assert not ctx.conv_range_flag()
    # round 4: with the tail on, the deconv's OUTPUT is stored as split rows as well (22-23 significant bits instead of 24), so the mask
    # predictor -- and with it loss_mask -- sees values rounded once more: equal to a few 1e-7, no longer bit for bit; the other four losses
    # do not depend on the mask head's storage at all
    for k in out[0][0]:
        if k == "loss_mask":
            assert out[1][0][k] == pytest.approx(out[0][0][k], rel=2e-6), (out[0][0], out[1][0])
        else:
            assert out[1][0][k] == out[0][0][k], (k, out[0][0], out[1][0])
    changed = 0
    for k in names:
        r, g = out[0][1][k], out[1][1][k]
        err = float(np.abs(g - r).max()) / max(float(np.abs(r).max()), 1e-8)
        assert err < 2e-5, (k, err)
        changed += int(not np.array_equal(r, g))
    assert changed > 0, "the switch changed nothing: the split mask tail did not run"
    m.close()
    ctx.close()


@pytest.mark.parametrize("B,S,mode", [(4, 1024, "f16x3"), (2, 512, "f16x3"), (2, 512, "f32")])
def test_sparse_rpn_backward_equals_the_dense_one(B, S, mode):
    """The RPN head's backward pass over the sampled anchors' pixels only (round 4, rpn_sparse.hip: <= 256 rows per image instead of every pixel of
    five maps; every term it leaves out is an exact zero) against the dense pass (amp_debug_set_rpn_sparse(0)): the same losses bit for bit (the forward
    pass does not change), every trainable gradient -- the head's own four tensors and everything upstream of the FPN maps it scatters into -- equal up
    to the order of the fp32 sums, the step repeats bit for bit, and the switch did switch."""
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    ctx = _lib.Context(0)
    ctx.conv_mode = mode
    K = 2
    imgs, gts = synth.batch(B, S, S, first_index=410)
    npp = P.init_params(K, seed=0, style="spread")
    m = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=B * 800, max_poly_doubles=B * 800 * 64)
    m.load_params(npp)
    names = [k for k in npp if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2")]
    out, ran = {}, {}
    try:
        _lib.lib().amp_debug_set_rpn_train_fuse(0)      # the same forward pass in both (the dense backward needs the head's hidden tensor saved)
        for on in (0, 1, 1):
            _lib.lib().amp_debug_set_rpn_sparse(on)
            losses = m.forward_losses(imgs, gts, seed=7, backward=True)
            got = (losses, {k: m.get_tensor(k, grad=True) for k in names})
            if on in out:      # the sparse step again: bit for bit
                assert got[0] == out[on][0]
                for k in names:
                    assert np.array_equal(got[1][k], out[on][1][k]), k
            out[on] = got
            ran[on] = _lib.lib().amp_debug_last_rpn_sparse(m._h)
    finally:
        _lib.lib().amp_debug_set_rpn_sparse(-1)
        _lib.lib().amp_debug_set_rpn_train_fuse(-1)
    assert ran == {0: 0, 1: 1}
    assert not ctx.conv_range_flag()
    assert out[0][0] == out[1][0]
    worst, changed = {}, 0
    for k in names:
        r, g = out[0][1][k], out[1][1][k]
        assert float(np.abs(r).max()) > 0, k
        err = float(np.abs(g - r).max()) / float(np.abs(r).max())
        worst[k] = err
        assert err < (2e-5 if mode == "f16x3" else 1e-4), (k, err)
        changed += int(not np.array_equal(r, g))
    head = [k for k in names if k.startswith("proposal_generator.rpn_head")]
    assert len(head) == 6 and changed >= 6, (head, changed)
    print("worst relative differences:", {k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    m.close()
    ctx.close()


def test_training_forward_with_the_fused_rpn_head_and_recomputed_hidden_rows():
    """A training step whose forward pass runs the RPN head like inference does -- the predictors in the 3x3 conv's epilogue, the hidden tensor never
    written -- and whose sparse backward pass recomputes the <= 256 hidden rows per image it needs from the gathered patches (round 4), against the step
    that saves the hidden tensor (amp_debug_set_rpn_train_fuse(0)): the predictor maps agree to 2e-6 of their range (another order of the same sums), the
    losses to 1e-5, every gradient to 5e-3 of its tensor's largest entry (proposals are a discrete function of the logits), the fused step repeats bit for
    bit, and after an SGD step both forward passes see the new predictor weights."""
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    ctx = _lib.Context(0)
    K, B, S = 2, 4, 1024
    imgs, gts = synth.batch(B, S, S, first_index=520)
    npp = P.init_params(K, seed=0, style="spread")
    m = MaskRCNN(ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=B * 800, max_poly_doubles=B * 800 * 64)
    m.load_params(npp)
    names = [k for k in npp if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2")]
    out = {}
    try:
        for fuse in (0, 1, 1):
            _lib.lib().amp_debug_set_rpn_train_fuse(fuse)
            losses = m.forward_losses(imgs, gts, seed=9, backward=True)
            got = (losses, {k: m.get_tensor(k, grad=True) for k in names}, m.tap("rpn_pred2"), m.tap("rpn_pred4"))
            if fuse in out:
                assert got[0] == out[fuse][0] and all(np.array_equal(got[1][k], out[fuse][1][k]) for k in names)
            out[fuse] = got
    finally:
        _lib.lib().amp_debug_set_rpn_train_fuse(-1)
    assert not ctx.conv_range_flag()
    for t in (2, 3):
        a, b = out[0][t], out[1][t]
        assert float(np.abs(a - b).max()) <= 2e-6 * float(np.abs(a).max())
    assert not np.array_equal(out[0][2], out[1][2])          # p2 did take the fused head (p4 at this size is below its threshold: the same launches either way)
    for k in out[0][0]:
        assert out[1][0][k] == pytest.approx(out[0][0][k], rel=1e-5), (k, out[0][0], out[1][0])
    worst = {}
    for k in names:
        r, g = out[0][1][k], out[1][1][k]
        worst[k] = float(np.abs(g - r).max()) / float(np.abs(r).max())
        # (1e-3 on the box head here: logits that moved by 1e-7 let near-duplicate proposals trade places in the top-k / NMS -- the RoI sets are discrete
        # functions of the logits; the backward pass itself is held to 6e-7 by test_sparse_rpn_backward_equals_the_dense_one on ONE forward pass)
        assert worst[k] < 5e-3, (k, worst[k])
    print("worst relative differences:", {k: f"{v:.1e}" for k, v in sorted(worst.items(), key=lambda kv: -kv[1])[:6]})
    # ... and after an SGD step both forward passes see the NEW predictor weights (the fused head reads their split copy, refreshed at the start of a step)
    m.sgd_step(0.05, 0.9, 1e-4)
    after = {}
    try:
        for fuse in (1, 0):
            _lib.lib().amp_debug_set_rpn_train_fuse(fuse)
            after[fuse] = m.forward_losses(imgs, gts, seed=9, backward=True)
    finally:
        _lib.lib().amp_debug_set_rpn_train_fuse(-1)
    assert abs(after[1]["loss_rpn_cls"] - out[1][0]["loss_rpn_cls"]) > 1e-4          # the step did move the head
    for k in after[0]:
        assert after[1][k] == pytest.approx(after[0][k], rel=2e-5), (k, after)
    m.close()
    ctx.close()


@pytest.mark.parametrize("case", ["one_small_image", "no_ground_truth_in_one_image", "ragged_maps"])
def test_sparse_rpn_backward_edge_cases(case):
    """The sparse RPN backward on inputs away from the bench's: one 256 x 320 image (maps down to 4 x 5 pixels, most sampled anchors on the coarse levels),
    a batch in which one image has no ground truth at all (256 negatives, no box loss there), image sizes that are not multiples of 64 (ragged level
    sizes) -- against the dense pass on the same forward pass: every gradient to 2e-5 of its tensor's largest entry."""
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    ctx = _lib.Context(0)
    K = 2
    B, H, W = {"one_small_image": (1, 256, 320), "no_ground_truth_in_one_image": (2, 384, 384), "ragged_maps": (2, 328, 440)}[case]
    imgs, gts = synth.batch(B, H, W, first_index=610)
    if case == "no_ground_truth_in_one_image":
        gts[1] = dict(boxes=gts[1]["boxes"][:0], classes=gts[1]["classes"][:0], polygons=gts[1]["polygons"][:0])
    npp = P.init_params(K, seed=1, style="spread")
    m = MaskRCNN(ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=B * 800, max_poly_doubles=B * 800 * 64)
    m.load_params(npp)
    names = [k for k in npp if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2")]
    out = {}
    try:
        _lib.lib().amp_debug_set_rpn_train_fuse(0)
        for on in (0, 1):
            _lib.lib().amp_debug_set_rpn_sparse(on)
            losses = m.forward_losses(imgs, gts, seed=5, backward=True)
            out[on] = (losses, {k: m.get_tensor(k, grad=True) for k in names})
            assert _lib.lib().amp_debug_last_rpn_sparse(m._h) == on
    finally:
        _lib.lib().amp_debug_set_rpn_sparse(-1)
        _lib.lib().amp_debug_set_rpn_train_fuse(-1)
    assert out[0][0] == out[1][0] and all(np.isfinite(v) for v in out[1][0].values())
    for k in names:
        r, g = out[0][1][k], out[1][1][k]
        scale = float(np.abs(r).max())
        assert np.isfinite(g).all() and (scale > 0 or not g.any()), k
        if scale > 0:
            assert float(np.abs(g - r).max()) / scale < 2e-5, (k, float(np.abs(g - r).max()) / scale)
    m.close()
    ctx.close()
