"""Checkpoint reader with detectron2's DetectionCheckpointer semantics (the reference's workflow: notebook cell 20
fine-tunes the 80-class COCO zoo model with NUM_CLASSES = 1, and its cfg.MODEL.WEIGHTs typo leaves the caffe2-named ImageNet
R-50.pkl in effect).  CPU only: the files are synthetic, of the real shapes."""
import pickle

import numpy as np
import pytest
import torch

from ampis_amd import checkpoint, params as P


def _coco80_pkl(path, seed=7):
    p = P.init_params(80, seed=seed, style="spread")
    with open(path, "wb") as f:
        pickle.dump({"model": {k: v for k, v in p.items()}, "__author__": "Detectron2 Model Zoo"}, f)
    return p


def test_coco_80_class_zoo_checkpoint_into_a_1_class_model(tmp_path, caplog):
    src = _coco80_pkl(tmp_path / "model_final_f10217.pkl")
    init = P.init_params(1, seed=3, style="d2")
    with caplog.at_level("WARNING", logger="ampis_amd"):
        got, rep = checkpoint.load_checkpoint(tmp_path / "model_final_f10217.pkl", 1, init=init, with_report=True)
    skipped = {n for n, _, _ in rep["shape_mismatch"]}
    assert skipped == {"roi_heads.box_predictor.cls_score.weight", "roi_heads.box_predictor.cls_score.bias",
                       "roi_heads.box_predictor.bbox_pred.weight", "roi_heads.box_predictor.bbox_pred.bias",
                       "roi_heads.mask_head.predictor.weight", "roi_heads.mask_head.predictor.bias"}
    assert ("roi_heads.box_predictor.cls_score.weight", (81, 1024), (2, 1024)) in rep["shape_mismatch"]
    assert rep["missing"] == [] and rep["unexpected"] == [] and rep["source"] == "d2"
    for name in got:
        want = init[name] if name in skipped else src[name]
        assert np.array_equal(got[name], want), name
    assert got["roi_heads.box_predictor.cls_score.weight"].shape == (2, 1024)
    assert "Skip loading parameters with a different shape" in caplog.text and "cls_score.weight" in caplog.text
    with pytest.raises(ValueError):
        checkpoint.load_checkpoint(tmp_path / "model_final_f10217.pkl", 1, strict=True)


def _msra_r50_pkl(path, seed=5):
    """The blob dict of detectron2://ImageNetPretrained/MSRA/R-50.pkl: caffe2 names, affine-only 'bn' (scale / bias, no statistics)."""
    rng = np.random.default_rng(seed)
    blobs = {}
    def conv(name, cout, cin, k):
        blobs[name + "_w"] = rng.normal(0, 0.05, (cout, cin, k, k)).astype(np.float32)
    def bn(name, c):
        blobs[name + "_bn_s"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
        blobs[name + "_bn_b"] = rng.normal(0, 0.1, c).astype(np.float32)
    conv("conv1", 64, 3, 7); bn("res_conv1", 64)
    cin = 64
    for si, (nblk, mid, cout) in enumerate(((3, 64, 256), (4, 128, 512), (6, 256, 1024), (3, 512, 2048))):
        for b in range(nblk):
            p = f"res{si + 2}_{b}"
            if b == 0:
                conv(p + "_branch1", cout, cin, 1); bn(p + "_branch1", cout)
            conv(p + "_branch2a", mid, cin, 1); bn(p + "_branch2a", mid)
            conv(p + "_branch2b", mid, mid, 3); bn(p + "_branch2b", mid)
            conv(p + "_branch2c", cout, mid, 1); bn(p + "_branch2c", cout)
            cin = cout
    blobs["fc1000_w"] = rng.normal(0, 0.01, (1000, 2048)).astype(np.float32)
    blobs["fc1000_b"] = np.zeros(1000, np.float32)
    blobs["conv1_w_momentum"] = np.zeros((64, 3, 7, 7), np.float32)          # optimizer residue: dropped like detectron2 does
    with open(path, "wb") as f:
        pickle.dump(blobs, f)
    return blobs


def test_caffe2_named_imagenet_backbone(tmp_path):
    blobs = _msra_r50_pkl(tmp_path / "R-50.pkl")
    init = P.init_params(1, seed=1, style="d2")
    got, rep = checkpoint.load_checkpoint(tmp_path / "R-50.pkl", 1, init=init, with_report=True)
    assert rep["source"] == "caffe2" and rep["renamed"] == 161 and rep["shape_mismatch"] == []
    assert rep["unexpected"] == ["fc1000.bias", "fc1000.weight"]
    # every backbone weight and its affine came from the file, under detectron2's names
    assert np.array_equal(got["backbone.bottom_up.stem.conv1.weight"], blobs["conv1_w"])
    assert np.array_equal(got["backbone.bottom_up.stem.conv1.norm.weight"], blobs["res_conv1_bn_s"])
    assert np.array_equal(got["backbone.bottom_up.stem.conv1.norm.bias"], blobs["res_conv1_bn_b"])
    assert np.array_equal(got["backbone.bottom_up.res2.0.shortcut.weight"], blobs["res2_0_branch1_w"])
    assert np.array_equal(got["backbone.bottom_up.res4.5.conv2.weight"], blobs["res4_5_branch2b_w"])
    assert np.array_equal(got["backbone.bottom_up.res5.2.conv3.norm.bias"], blobs["res5_2_branch2c_bn_b"])
    # FrozenBatchNorm2d buffers the file does not have: mean 0, var 1 - eps  ->  scale = weight exactly (params.BN_EPS = 1e-5)
    rv = got["backbone.bottom_up.res3.1.conv1.norm.running_var"]
    assert np.all(got["backbone.bottom_up.res3.1.conv1.norm.running_mean"] == 0) and np.all(rv == np.float32(1.0) - np.float32(1e-5))
    w = got["backbone.bottom_up.res3.1.conv1.norm.weight"]
    assert np.array_equal((w * (1.0 / np.sqrt(rv + np.float32(1e-5)))).astype(np.float32), w)
    # everything outside the backbone keeps its initialisation and is reported missing
    outside = [n for n in init if not n.startswith("backbone.bottom_up.")]
    assert set(n for n in rep["missing"] if not n.startswith("backbone.bottom_up.")) == set(outside)
    assert all(np.array_equal(got[n], init[n]) for n in outside)
    assert all(n.endswith(("running_mean", "running_var")) for n in rep["missing"] if n.startswith("backbone.bottom_up."))


def test_detectron2_named_heuristics_pkl_keeps_its_bn_statistics(tmp_path):
    """A torchvision-converted ImageNet backbone (detectron2's convert-torchvision-to-d2.py output, ImageNetPretrained/torchvision/R-50.pkl):
    detectron2 module names WITHOUT the backbone prefix, `matching_heuristics: True`, author not Caffe2.  detectron2 converts names only
    for `__author__ == 'Caffe2'`; the heuristics switch only enables suffix matching, so running_mean / running_var must arrive intact."""
    full = P.init_params(1, seed=4, style="spread")
    pre = "backbone.bottom_up."
    rng = np.random.default_rng(2)
    model = {}
    for k, v in full.items():
        if k.startswith(pre):
            model[k[len(pre):]] = rng.normal(0.3, 1.0, v.shape).astype(np.float32) if k.endswith(("running_mean", "running_var")) else v
    model["stem.fc.weight"] = np.zeros((1000, 2048), np.float32)
    with open(tmp_path / "R-50.pkl", "wb") as f:
        pickle.dump({"model": model, "__author__": "torchvision", "matching_heuristics": True}, f)
    init = P.init_params(1, seed=1, style="d2")
    got, rep = checkpoint.load_checkpoint(tmp_path / "R-50.pkl", 1, init=init, with_report=True)
    assert rep["source"] == "d2-suffix" and rep["renamed"] == 0 and rep["shape_mismatch"] == []
    assert rep["unexpected"] == ["stem.fc.weight"]
    assert not [n for n in rep["missing"] if n.startswith(pre)]
    for k, v in model.items():
        if k != "stem.fc.weight":
            assert np.array_equal(got[pre + k], v), k
    assert not np.all(got[pre + "stem.conv1.norm.running_mean"] == 0)          # the statistics of the file, not the FrozenBN defaults
    # the same keys with exact names and no heuristics flag: nothing matches (detectron2 would not match them either)
    with open(tmp_path / "plain.pkl", "wb") as f:
        pickle.dump({"model": model, "__author__": "torchvision"}, f)
    _, rep2 = checkpoint.load_checkpoint(tmp_path / "plain.pkl", 1, init=init, with_report=True)
    assert rep2["source"] == "d2" and len(rep2["unexpected"]) == len(model)


def test_c2_name_conversion_table():
    ren = checkpoint.convert_c2_backbone_names(["conv1_w", "res_conv1_bn_s", "res_conv1_bn_b", "res2_0_branch2a_w", "res2_0_branch2a_bn_s",
                                                "res3_0_branch1_bn_b", "res4_22_branch2c_w", "res5_1_branch2b_bn_rm", "res5_1_branch2b_bn_riv", "fc1000_b"])
    assert ren == {"conv1_w": "stem.conv1.weight", "res_conv1_bn_s": "stem.conv1.norm.weight", "res_conv1_bn_b": "stem.conv1.norm.bias",
                   "res2_0_branch2a_w": "res2.0.conv1.weight", "res2_0_branch2a_bn_s": "res2.0.conv1.norm.weight",
                   "res3_0_branch1_bn_b": "res3.0.shortcut.norm.bias", "res4_22_branch2c_w": "res4.22.conv3.weight",
                   "res5_1_branch2b_bn_rm": "res5.1.conv2.norm.running_mean", "res5_1_branch2b_bn_riv": "res5.1.conv2.norm.running_var",
                   "fc1000_b": "fc1000.bias"}


def test_pth_round_trip_with_iteration_and_momentum(tmp_path):
    p = P.init_params(2, seed=9, style="spread")
    mom = np.random.default_rng(0).normal(size=1000).astype(np.float32)
    checkpoint.save_checkpoint(tmp_path / "model_0000041.pth", p, iteration=41, optimizer={"arena": mom})
    got, rep = checkpoint.load_checkpoint(tmp_path / "model_0000041.pth", 2, strict=True, with_report=True)
    assert all(np.array_equal(got[k], p[k]) for k in p) and rep["missing"] == [] and rep["unexpected"] == []
    assert checkpoint.checkpoint_iteration(tmp_path / "model_0000041.pth") == 41
    assert np.array_equal(checkpoint.checkpoint_momentum(tmp_path / "model_0000041.pth")["arena"], mom)
    d = torch.load(tmp_path / "model_0000041.pth", weights_only=False)
    assert set(d) >= {"model", "iteration", "optimizer"}                      # detectron2's checkpoint keys


def test_urls_are_refused_without_a_network(tmp_path):
    with pytest.raises(FileNotFoundError):
        checkpoint.load_checkpoint("detectron2://COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x/137849600/model_final_f10217.pkl", 1)


def test_a_torch_layout_optimizer_state_is_reported_not_silently_dropped(tmp_path, caplog):
    """A detectron2 / reference checkpoint stores torch's SGD.state_dict() (`state` by parameter index + `param_groups`): its momentum cannot be
    mapped to names here -- checkpoint_momentum returns nothing AND says so, while weights and iteration still load."""
    import logging
    import torch
    from ampis_amd import checkpoint, params as P
    p = P.init_params(1, seed=1, style="spread")
    path = tmp_path / "d2_style.pth"
    torch.save({"model": {k: torch.as_tensor(v) for k, v in p.items()}, "iteration": 41,
                "optimizer": {"state": {0: {"momentum_buffer": torch.zeros(3)}, 1: {"momentum_buffer": torch.ones(2)}},
                              "param_groups": [{"lr": 0.02, "momentum": 0.9, "params": [0, 1]}]}}, str(path))
    with caplog.at_level(logging.WARNING, logger="ampis_amd"):
        mom = checkpoint.checkpoint_momentum(str(path))
    assert mom == {} and any("momentum is NOT" in r.getMessage() for r in caplog.records)
    assert checkpoint.checkpoint_iteration(str(path)) == 41
    # the format written here round-trips with its buffers
    own = tmp_path / "own.pth"
    checkpoint.save_checkpoint(str(own), p, iteration=7, optimizer={"roi_heads.box_head.fc1.weight": np.ones_like(p["roi_heads.box_head.fc1.weight"])})
    assert set(checkpoint.checkpoint_momentum(str(own))) == {"roi_heads.box_head.fc1.weight"}
