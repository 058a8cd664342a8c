"""X-101-32x8d-FPN (BASELINE configs[4]: grouped 3x3 convs, stride in the 3x3, blocks 3-4-23-3) through the same C ABI, against
the CPU oracle on identical weights and inputs.  Same gates as tests/test_e2e_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(gpu_ctx):
    from ampis_amd import params as P
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as O
    from test_e2e_gpu import synth_image
    K, B, H, W, D = 2, 2, 192, 256, 60
    rng = np.random.default_rng(11)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    std = (57.375, 57.120, 58.395)          # the X101 model-zoo config's PIXEL_STD
    np_params = P.init_params(K, seed=4, style="spread", arch="X101")
    # the seeded stem weights assume unit PIXEL_STD; compensate so that activations (and scores) are as spread as in the R50 tests
    np_params["backbone.bottom_up.stem.conv1.weight"] = np_params["backbone.bottom_up.stem.conv1.weight"] * np.float32(57.0)
    cfg = O.Cfg(num_classes=K, detections_per_image=D, pixel_std=std, resnet_blocks=(3, 4, 23, 3), num_groups=32, stride_in_1x1=False)
    stages = {}
    ref = O.infer(imgs, O.to_torch_params(np_params), cfg, stages=stages)
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D, pixel_std=std, arch="X101")
    model.load_params(np_params)
    out = model.infer(imgs)
    yield dict(model=model, out=out, ref=ref, stages=stages, H=H, W=W, params=np_params, np_params=np_params, imgs=imgs, cfg=cfg)
    model.close()


def test_x101_backbone_taps(setup):
    from test_e2e_gpu import _nhwc, _relerr
    m, st = setup["model"], setup["stages"]
    for i, name in enumerate(["res2", "res3", "res4", "res5"]):
        got, ref = m.tap(name), _nhwc(st["res"][name])
        assert got.shape == ref.shape, name
        assert _relerr(got, ref) < 4e-5 * (i + 2), name


def test_x101_final_outputs_match_oracle(setup):
    from test_e2e_gpu import test_final_outputs_match_oracle as gate
    gate(setup)


def test_x101_grouped_weight_round_trip(setup):
    m, p = setup["model"], setup["params"]
    for name in ("backbone.bottom_up.res2.0.conv2.weight", "backbone.bottom_up.res4.22.conv2.weight", "backbone.bottom_up.res5.2.conv2.weight"):
        assert np.array_equal(m.get_tensor(name), p[name]), name


def _expand_windows(w, groups):
    """grouped OIHW [O, O/G, KH, KW] -> the window layout [O][KH][KW][64] of the forward kernel (slot = input channel - start of o's 64-tile)."""
    O, cpg, KH, KW = w.shape
    out = np.zeros((O, KH, KW, 64), np.float32)
    for o in range(O):
        j0 = (o // cpg) * cpg - (o & ~63)
        out[o, :, :, j0:j0 + cpg] = w[o].transpose(1, 2, 0)
    return out


@pytest.mark.parametrize("C,groups,stride,H,W", [(256, 32, 1, 37, 41), (256, 32, 2, 38, 45), (512, 32, 2, 24, 20), (1024, 32, 1, 9, 14), (128, 2, 1, 16, 16)])
def test_grouped_conv_gradients_against_torch_autograd(gpu_ctx, C, groups, stride, H, W):
    """amp_conv2d_grouped_wgrad and the data gradient (amp_group_dgrad_weights + the grouped forward kernel, stride 2 through the zero-spread
    map) against torch.conv2d(groups=G) autograd on the CPU, FrozenBN scale folded in as the model does; 8 / 16 / 32 / 64 channels per group."""
    import ctypes as C_
    from ampis_amd._lib import ConvDesc, check, lib, ptr
    g = torch.Generator().manual_seed(C + stride)
    B, cpg = 2, C // groups
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, cpg, 3, 3, generator=g) * 0.1
    scale = torch.rand(C, generator=g) + 0.5
    Ho, Wo = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    dy = torch.randn(B, C, Ho, Wo, generator=g) * 1e-3
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = torch.nn.functional.conv2d(xr, wr, None, stride=stride, padding=1, groups=groups) * scale[None, :, None, None]
    y.backward(dy)
    d = "cuda:0"
    x_d = x.permute(0, 2, 3, 1).contiguous().to(d)
    dy_d = dy.permute(0, 2, 3, 1).contiguous().to(d)
    w_win = torch.from_numpy(_expand_windows(w.numpy(), groups)).to(d)
    sc_d = scale.to(d)
    desc = ConvDesc(B, H, W, C, C, 3, 3, stride, 1, 0, 0, 0)
    nscr = lib().amp_grouped_wgrad_scratch_floats(C_.byref(desc))
    scratch = torch.empty(nscr, device=d)
    gw = torch.full((C, 3, 3, 64), 7.0, device=d)
    for rep in range(2):
        check(lib().amp_conv2d_grouped_wgrad(gpu_ctx.handle, C_.byref(desc), groups, ptr(x_d), ptr(dy_d), ptr(sc_d), ptr(scratch), ptr(gw)), "amp_conv2d_grouped_wgrad")
        torch.cuda.synchronize()
        if rep == 0:
            first = gw.clone()
    assert torch.equal(first, gw)                                                  # slices added in fixed order
    ref_w = _expand_windows(wr.grad.numpy(), groups)
    got_w = gw.cpu().numpy()
    assert np.abs(got_w - ref_w).max() <= 2e-5 * np.abs(ref_w).max()
    outside = _expand_windows(np.ones_like(w.numpy()), groups) == 0
    assert np.all(got_w[outside] == 0)                                             # structurally zero weights keep a zero gradient
    # data gradient
    wt = torch.empty(C, 3, 3, 64, device=d)
    check(lib().amp_group_dgrad_weights(gpu_ctx.handle, ptr(w_win), ptr(sc_d), C, 3, 3, ptr(wt)), "amp_group_dgrad_weights")
    src = dy_d
    if stride == 2:
        src = torch.zeros(B, H, W, C, device=d)
        check(lib().amp_subsample2_bwd(gpu_ctx.handle, ptr(dy_d), ptr(src), B, H, W, C), "amp_subsample2_bwd")
    dd = ConvDesc(B, H, W, C, C, 3, 3, 1, 1, 0, 0, 0)
    dx = torch.empty(B, H, W, C, device=d)
    check(lib().amp_conv2d_grouped_nhwc(gpu_ctx.handle, C_.byref(dd), groups, ptr(src), ptr(wt), None, None, None, ptr(dx)), "amp_conv2d_grouped_nhwc")
    torch.cuda.synchronize()
    ref_x = xr.grad.permute(0, 2, 3, 1).numpy()
    assert np.abs(dx.cpu().numpy() - ref_x).max() <= 2e-5 * np.abs(ref_x).max()


def test_x101_training_step_against_autograd(gpu_ctx):
    """A whole training step of X-101-32x8d-FPN (grouped conv2 with the block's stride in the 3x3): the five losses and the gradients of
    every trainable tensor against torch autograd of the training oracle (BASELINE configs[4] trains now; VERDICT r02 missing #4)."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    K, B, H, W = 2, 1, 128, 160
    imgs, gts = synth.batch(B, H, W, seed=31)
    gts = [dict(boxes=g["boxes"][:25], classes=g["classes"][:25], polygons=g["polygons"][:25]) for g in gts]
    npp = P.init_params(K, seed=4, style="spread", arch="X101")
    tp = M.to_torch_params(npp)
    names = [k for k in tp if ".norm." not in k and not k.startswith(("backbone.bottom_up.stem", "backbone.bottom_up.res2"))]
    for k in names:
        tp[k].requires_grad_(True)
    cfg = T.TrainCfg(num_classes=K, seed=3, resnet_blocks=(3, 4, 23, 3), num_groups=32, stride_in_1x1=False)
    ref = T.forward_losses(imgs, gts, tp, cfg)
    sum(ref.values()).backward()
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), arch="X101", train=True, max_gt=512, max_poly_doubles=512 * 64)
    m.load_params(npp)
    from ampis_amd import _lib
    # round 4: the default is the scaled split gradient chain for ResNeXt blocks (chain 3); the fp32-gradient chain on split activations (1)
    # is held to the same bar, and the two agree with each other far closer than either does with autograd
    grads = {}
    # third pass: the patch-staged grouped 3x3 (conv3x3_c64_kernel) forced on at this small size -- forward conv2 AND its data gradient with the
    # split ReLU mask run through it (at bench sizes it is the default; its grid rule would leave it out here)
    for gx, chain, patch in ((1, 3, 1), (0, 1, 1), (1, 3, 2)):
        _lib.lib().amp_debug_set_gx(gx)
        _lib.lib().amp_debug_set_patch_conv(patch)
        try:
            got = m.forward_losses(imgs, gts, seed=3, backward=True)
        finally:
            _lib.lib().amp_debug_set_gx(-1)
            _lib.lib().amp_debug_set_patch_conv(1)
        assert _lib.lib().amp_debug_last_backward_chain(m._h) == chain, (gx, _lib.lib().amp_debug_last_backward_chain(m._h))
        for k, v in ref.items():
            assert got[k] == pytest.approx(float(v), rel=3e-4, abs=1e-6), (k, got[k], float(v))
        bad = []
        for name in names:
            g, r = m.get_tensor(name, grad=True), tp[name].grad.numpy()
            grads.setdefault(name, []).append(g)
            err = float(np.abs(g - r).max()) / max(float(np.abs(r).max()), 1e-8)
            if err > 3e-3:
                bad.append((round(err, 5), name))
        assert not bad, f"chain {chain}: {len(bad)} of {len(names)} tensors off: {sorted(bad, reverse=True)[:8]}"
    worst = max(float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-12) for a, b, _ in grads.values())
    assert worst < 5e-4, worst
    assert all(np.array_equal(a, c) for a, _, c in grads.values()), "the patch-staged grouped 3x3 must give the implicit-GEMM kernels' gradients bit for bit"
    grouped = [n for n in names if n.endswith(".conv2.weight")]
    assert len(grouped) == 4 + 23 + 3 and all(np.abs(m.get_tensor(n, grad=True)).max() > 0 for n in grouped[:3])
    # an SGD step moves the grouped weights and leaves the structural zeros of their windows zero (weight decay on 0 is 0)
    before = m.get_tensor("backbone.bottom_up.res4.5.conv2.weight")
    m.sgd_step(0.01, 0.9, 1e-4)
    after = m.get_tensor("backbone.bottom_up.res4.5.conv2.weight")
    assert not np.array_equal(before, after)
    l2 = m.forward_losses(imgs, gts, seed=3)
    assert all(np.isfinite(v) for v in l2.values())
    m.close()


def test_r101_inference_and_training_step(gpu_ctx):
    """R101-FPN (blocks 3-4-23-3, dense 3x3, stride in the 1x1): inference against the oracle and one training step (losses vs oracle)."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as O, train as T
    from test_e2e_gpu import synth_image
    K, B, H, W, D = 2, 2, 192, 256, 60
    rng = np.random.default_rng(21)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    p = P.init_params(K, seed=6, style="spread", arch="R101")
    for k in p:     # 33 residual blocks amplify rounding noise into different NMS decisions with the 16-block damping of "spread"
        if ".conv3.norm.weight" in k:
            p[k] = p[k] * np.float32(0.5)
    cfg = O.Cfg(num_classes=K, detections_per_image=D, resnet_blocks=(3, 4, 23, 3))
    from oracle import gate
    ref, floor = gate.floor_of(lambda: O.infer(imgs, O.to_torch_params(p), cfg), (H, W))
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D, arch="R101", train=True,
                 max_gt=B * 700, max_poly_doubles=B * 700 * 64)
    m.load_params(p)
    out = m.infer(imgs)
    from test_e2e_gpu import _decode
    from oracle import gate
    st = gate.merge([gate.check_image(o, r, H, W, lambda mk: _decode(mk["counts"], H, W)) for o, r in zip(out, ref)])
    print("R101 gate:", gate.summary(st))
    assert st["instances"] > 60
    print("R101 gate |", gate.assert_floor(st, floor, sigmas=3.0, floor_sigmas=2.5))      # bound to the oracle's own noise on these images (33 blocks amplify it)
    # one training step runs and produces finite, sensible losses and gradients for a res4.22 weight
    timgs, gts = synth.batch(B, H, W, first_index=40)
    L = m.forward_losses(timgs, gts, seed=1, backward=True)
    assert all(np.isfinite(v) and v > 0 for v in L.values()), L
    g = m.get_tensor("backbone.bottom_up.res4.22.conv2.weight", grad=True)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    m.close()
