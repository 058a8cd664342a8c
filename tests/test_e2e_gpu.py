"""End-to-end and stage-wise parity of the HIP Mask R-CNN path (through the C ABI: amp_model_*) against the CPU oracle
on identical weights and identical input bytes.

Tolerances (fp32, BASELINE.json north_star; the rule is oracle/gate.py): box |delta| < 1e-3 px, masks identical except threshold ties, identical
class ids.  Stage taps are compared with a relative tolerance that grows with depth (fp32 re-association through ~60
layers); selection stages (top-k / NMS) are compared on matched sets because a 1e-6 score difference may legally flip
the order of two near-tied candidates.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BOX_TOL = 1e-3


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


def _relerr(a, b):
    return float(np.abs(a - b).max() / max(1e-6, np.abs(b).max()))


def synth_image(rng, h, w):
    img = rng.normal(60, 12, (h, w))
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(12):
        cy, cx, r = rng.uniform(0, h), rng.uniform(0, w), rng.uniform(6, 40)
        d = (yy - cy) ** 2 + (xx - cx) ** 2
        img = np.where(d < r * r, rng.normal(190, 15) - 40 * d / (r * r), img)
    img = np.clip(img + rng.normal(0, 4, (h, w)), 0, 255).astype(np.uint8)
    return np.repeat(img[:, :, None], 3, axis=2)


@pytest.fixture(scope="module")
def setup(gpu_ctx):
    from ampis_amd import params as P
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as O

    K, B, H, W, D = 2, 2, 224, 288, 60
    rng = np.random.default_rng(5)
    imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
    np_params = P.init_params(K, seed=3, style="spread")
    cfg = O.Cfg(num_classes=K, detections_per_image=D)
    stages = {}
    ref = O.infer(imgs, O.to_torch_params(np_params), cfg, stages=stages)
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D)
    model.load_params(np_params)
    out = model.infer(imgs)
    return dict(model=model, out=out, ref=ref, stages=stages, cfg=cfg, B=B, H=H, W=W, K=K, D=D, imgs=imgs, np_params=np_params)


def test_backbone_fpn_taps(setup):
    m, st = setup["model"], setup["stages"]
    for i, name in enumerate(["res2", "res3", "res4", "res5"]):
        assert _relerr(m.tap(name), _nhwc(st["res"][name])) < 2e-5 * (i + 2), name
    for i, name in enumerate(["p2", "p3", "p4", "p5", "p6"]):
        assert _relerr(m.tap(name), _nhwc(st["feats"][i])) < 1e-4, name


def test_rpn_head_taps(setup):
    m, st = setup["model"], setup["stages"]
    for i, name in enumerate(["rpn_pred2", "rpn_pred3", "rpn_pred4", "rpn_pred5", "rpn_pred6"]):
        logits, deltas = st["rpn_outs"][i]
        got = m.tap(name)[:, :, :15]   # [B, HW, 16]: 3 logits, 12 deltas, 1 zero pad column
        B, HW, _ = got.shape
        ref = np.concatenate([logits.numpy().reshape(B, HW, 3), deltas.numpy().reshape(B, HW, 12)], axis=2)
        assert _relerr(got, ref) < 2e-4, name


def _match_boxes(a, b, tol):
    """fraction of rows of a that have a row of b within tol (max abs coordinate difference)."""
    if len(a) == 0:
        return 1.0
    d = np.abs(a[:, None, :] - b[None, :, :]).max(axis=2)
    return float((d.min(axis=1) < tol).mean())


def test_proposals(setup):
    m, st, B = setup["model"], setup["stages"], setup["B"]
    pb, pc = m.tap("prop_boxes"), m.tap("prop_count")
    for b in range(B):
        ref = st["props"][b][0].numpy()
        got = pb[b, : pc[b]]
        assert abs(len(ref) - len(got)) <= max(2, len(ref) // 100), (len(ref), len(got))
        assert _match_boxes(ref, got, 5e-3) > 0.99


def test_box_head_taps(setup):
    m, st, B = setup["model"], setup["stages"], setup["B"]
    # compare only rows whose proposal is identical in both paths
    pb, pc = m.tap("prop_boxes"), m.tap("prop_count")
    pooled, pred = m.tap("box_pooled"), m.tap("box_pred")
    Rcap = pb.shape[1]
    o = 0
    checked = 0
    for b in range(B):
        ref_boxes = st["props"][b][0].numpy()
        n = min(len(ref_boxes), pc[b])
        same = np.abs(ref_boxes[:n] - pb[b, :n]).max(axis=1) < 1e-4
        rp = st["pooled"][o:o + len(ref_boxes)].permute(0, 2, 3, 1).numpy()[:n][same]
        gp = pooled[b * Rcap: b * Rcap + n][same]
        assert _relerr(gp, rp) < 1e-4
        rs = np.concatenate([st["box_scores"][o:o + len(ref_boxes)].numpy(), st["box_deltas"][o:o + len(ref_boxes)].numpy()], 1)[:n][same]
        gs = pred[b * Rcap: b * Rcap + n][same][:, :rs.shape[1]]
        assert _relerr(gs, rs) < 2e-4
        checked += int(same.sum())
        o += len(ref_boxes)
    assert checked > 100


def _decode(rle_counts, h, w):
    from ampis_amd import rle
    return rle.decode({"size": [h, w], "counts": rle_counts}).astype(bool)


def test_final_outputs_match_oracle(setup):
    """The north-star gate (oracle/gate.py): per image the same number of detections; EVERY oracle instance has a HIP twin with the
    same class, box |delta| < 1e-3 px, score within 1e-4; a mask pixel may differ only where the oracle's own pasted probability
    is within 1e-4 of the 0.5 threshold (a tie), so every mask without such a tie is bit-identical (IoU 1 >= 0.999)."""
    from oracle import gate
    out, ref, H, W = setup["out"], setup["ref"], setup["H"], setup["W"]
    st = gate.merge([gate.check_image(o, r, H, W, lambda m: _decode(m["counts"], H, W)) for o, r in zip(out, ref)])
    print("e2e gate:", gate.summary(st))
    assert st["instances"] > 20 and st["identical"] + st["tie_masks"] == st["instances"]
    # what the gate relaxed, bound to the reference arithmetic's own noise on THESE images (fp32 oracle against its exact-convolution evaluation)
    from oracle import maskrcnn as O
    _, floor = gate.floor_of(lambda: O.infer(setup["imgs"], O.to_torch_params(setup["np_params"]), setup["cfg"]), (H, W))
    print("e2e gate |", gate.assert_floor(st, floor, sigmas=3.0, floor_sigmas=2.5))


def test_mask_prob_tap(setup):
    m, st, B = setup["model"], setup["stages"], setup["B"]
    got, rois = m.tap("mask_prob"), m.tap("mask_rois")
    ref_boxes = np.concatenate([d[0].numpy() for d in st["dets"]])
    ref_prob = st["mask_prob"].numpy()
    n = min(len(rois), len(ref_boxes))
    same = np.abs(rois[:n] - ref_boxes[:n]).max(axis=1) < 1e-4
    assert same.sum() > 10
    assert np.abs(got[:n][same] - ref_prob[:n][same]).max() < 2e-4


def test_rle_sizes_and_area(setup):
    from ampis_amd import rle
    for o in setup["out"]:
        for mk in o["masks"]:
            c = rle.string_to_counts(mk["counts"])
            assert int(c.sum()) == mk["size"][0] * mk["size"][1]
