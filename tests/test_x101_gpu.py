"""X-101-32x8d-FPN (BASELINE configs[4]: grouped 3x3 convs, stride in the 3x3, blocks 3-4-23-3) through the same C ABI, against
the CPU oracle on identical weights and inputs.  Same gates as tests/test_e2e_gpu.py."""
import numpy as np
import pytest

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
    yield dict(model=model, out=out, ref=ref, stages=stages, H=H, W=W, params=np_params)
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


def test_grouped_backbone_refuses_training(gpu_ctx):
    from ampis_amd import _lib
    from ampis_amd.model import MaskRCNN
    with pytest.raises(_lib.AmpError):
        MaskRCNN(gpu_ctx, 2, max_batch=1, max_h=64, max_w=64, arch="X101", train=True, max_gt=16, max_poly_doubles=1024)


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
    ref = O.infer(imgs, O.to_torch_params(p), cfg)
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D, arch="R101", train=True,
                 max_gt=B * 700, max_poly_doubles=B * 700 * 64)
    m.load_params(p)
    out = m.infer(imgs)
    from test_e2e_gpu import _decode
    from oracle import gate
    st = gate.merge([gate.check_image(o, r, H, W, lambda mk: _decode(mk["counts"], H, W)) for o, r in zip(out, ref)])
    print("R101 gate:", gate.summary(st))
    assert st["instances"] > 60
    gate.assert_bounds(st, iou_below_share=0.05, box_rel_used=0, iou_min=0.95)
    # one training step runs and produces finite, sensible losses and gradients for a res4.22 weight
    timgs, gts = synth.batch(B, H, W, first_index=40)
    L = m.forward_losses(timgs, gts, seed=1, backward=True)
    assert all(np.isfinite(v) and v > 0 for v in L.values()), L
    g = m.get_tensor("backbone.bottom_up.res4.22.conv2.weight", grad=True)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    m.close()
