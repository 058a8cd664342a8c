"""Backward pass + SGD (amp_model_forward_backward / amp_model_sgd_step, through the C ABI) against torch autograd of the
training oracle on identical inputs, weights and sampling seed.  RoIAlign backward uses float atomics, so gradients are
compared with a relative tolerance (not bitwise)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(gpu_ctx):
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    K, B, H, W = 2, 2, 192, 256
    imgs, gts = synth.batch(B, H, W, seed=9)
    gts = [dict(boxes=g["boxes"][:40], classes=g["classes"][:40], polygons=g["polygons"][:40]) for g in gts]
    npp = P.init_params(K, seed=2, style="spread")
    tp = M.to_torch_params(npp)
    train_names = [k for k in tp if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2")]
    for k in train_names:
        tp[k].requires_grad_(True)
    cfg = T.TrainCfg(num_classes=K, seed=3)
    ref = T.forward_losses(imgs, gts, tp, cfg)
    sum(ref.values()).backward()
    ref_grads = {k: tp[k].grad.detach().numpy() for k in train_names}
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=2048, max_poly_doubles=2048 * 64)
    model.load_params(npp)
    got = model.forward_losses(imgs, gts, seed=3, backward=True)
    return dict(model=model, got=got, ref={k: float(v) for k, v in ref.items()}, ref_grads=ref_grads, names=train_names, npp=npp)


def test_losses_unchanged_by_backward(setup):
    for k, v in setup["ref"].items():
        assert setup["got"][k] == pytest.approx(v, rel=2e-4, abs=1e-6), k


def test_gradients_match_autograd(setup):
    m = setup["model"]
    worst = []
    for name in setup["names"]:
        g = m.get_tensor(name, grad=True)
        r = setup["ref_grads"][name]
        assert g.shape == r.shape, name
        scale = max(float(np.abs(r).max()), 1e-8)
        err = float(np.abs(g - r).max()) / scale
        worst.append((err, name))
    worst.sort(reverse=True)
    bad = [(e, n) for e, n in worst if e > 2e-3]
    assert not bad, f"{len(bad)} tensors off: {bad[:8]}"


def test_sgd_step_matches_torch(setup):
    """One SGD step (lr, momentum, weight decay) == torch.optim.SGD on the oracle's gradients."""
    m, npp = setup["model"], setup["npp"]
    lr, mu, wd = 0.01, 0.9, 1e-4
    m.sgd_step(lr, mu, wd)
    for name in ("backbone.fpn_output3.weight", "roi_heads.box_head.fc2.bias", "backbone.bottom_up.res4.2.conv2.weight",
                 "roi_heads.mask_head.deconv.weight", "proposal_generator.rpn_head.anchor_deltas.weight"):
        p0 = npp[name]
        g = setup["ref_grads"][name]
        expect = p0 - lr * (g + wd * p0)      # first step: momentum buffer = g'
        got = m.get_tensor(name)
        assert np.abs(got - expect).max() <= 2e-3 * lr * max(np.abs(g).max(), 1e-6) + 1e-7, name
    # frozen tensors did not move
    assert np.array_equal(m.get_tensor("backbone.bottom_up.res2.0.conv1.weight"), npp["backbone.bottom_up.res2.0.conv1.weight"])


def test_grad_arena_torch_view_is_zero_copy(setup, gpu_ctx):
    """The RCCL all-reduce runs on a torch tensor that aliases the gradient arena (no copy): writes through the view are seen by
    amp_model_get_tensor."""
    from ampis_amd.utils import comm
    m = setup["model"]
    ptr, n = m.grad_arena()
    t = comm.arena_as_tensor(ptr, n, torch.device("cuda", 0))
    assert t.is_cuda and t.dtype == torch.float32 and t.numel() == n and t.data_ptr() == ptr
    before = m.get_tensor("backbone.fpn_output3.weight", grad=True)
    t.mul_(2.0)
    torch.cuda.synchronize()
    after = m.get_tensor("backbone.fpn_output3.weight", grad=True)
    assert np.allclose(after, 2.0 * before)
    assert comm.all_reduce_gradients(m, gpu_ctx) == 1.0      # single process: no-op, scale 1


def test_gradients_single_class_ragged_batch(gpu_ctx):
    """The tutorial's training configuration in miniature: NUM_CLASSES = 1 and two differently sized images in one batch
    (amp_model_set_image_sizes).  Every trainable tensor's gradient against autograd of the oracle."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    K, B, H, W = 1, 2, 160, 224
    sizes = [(160, 224), (128, 171)]
    imgs, gts = synth.batch(B, H, W, seed=15)
    def inside(g, h, w, n):
        keep = [i for i in range(len(g["boxes"])) if g["boxes"][i][2] <= w - 1 and g["boxes"][i][3] <= h - 1][:n]
        return dict(boxes=np.asarray(g["boxes"])[keep], classes=np.zeros(len(keep), np.int64), polygons=[g["polygons"][i] for i in keep])
    gts = [inside(gts[b], *sizes[b], 25) for b in range(B)]
    assert all(len(g["boxes"]) >= 3 for g in gts)
    npp = P.init_params(K, seed=4, style="spread")
    tp = M.to_torch_params(npp)
    names = [k for k in tp if ".norm." not in k and not k.startswith("backbone.bottom_up.stem") and not k.startswith("backbone.bottom_up.res2")]
    for k in names:
        tp[k].requires_grad_(True)
    ref = T.forward_losses(imgs, gts, tp, T.TrainCfg(num_classes=K, seed=5), image_sizes=sizes)
    sum(ref.values()).backward()
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=2048, max_poly_doubles=2048 * 64)
    model.load_params(npp)
    model.set_image_sizes(sizes)
    got = model.forward_losses(imgs, gts, seed=5, backward=True)
    for k, v in ref.items():
        assert got[k] == pytest.approx(float(v.detach()), rel=2e-4, abs=1e-6), k
    bad = []
    for name in names:
        g, r = model.get_tensor(name, grad=True), tp[name].grad.detach().numpy()
        assert g.shape == r.shape, name
        err = float(np.abs(g - r).max()) / max(float(np.abs(r).max()), 1e-8)
        # 2e-3 in the default arithmetic; under AMP_CONV_MODE=f32 one tensor (res4.3.conv3.weight) sits at 2.7e-3: the fp32 MFMA's
        # re-association noise is the larger of the two and the autograd reference is fp32 itself (measured on rounds 2 and 3 alike)
        if err > (4e-3 if gpu_ctx.conv_mode == gpu_ctx.CONV_F32 else 2e-3):
            bad.append((err, name))
    model.close()
    assert not bad, f"{len(bad)} tensors off: {sorted(bad, reverse=True)[:8]}"
