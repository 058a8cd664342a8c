"""CPU tests of the rows either side of the hot path: get_ddicts / extract_boxes against the reference's own
VIA annotation data (tests/golden/via_subset.json) and synthetic label images; det_seg_scores / the matcher against the
reference's known-answer vectors (ampis/analyze.py:702-728) and the oracle; frPyObjects against the oracle's rleFrPoly."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "via_subset.json")


def test_get_ddicts_via2_on_reference_annotations(tmp_path):
    from ampis_amd import data_utils
    from ampis_amd.structures import BoxMode
    g = json.load(open(GOLD))
    via_dir = tmp_path / "via_2.0.8"
    via_dir.mkdir()
    jpath = via_dir / "via_powder_particle_masks_training.json"
    json.dump(g["via"], open(jpath, "w"))
    dd = data_utils.get_ddicts(label_fmt="via2", im_root=jpath, dataset_class="Train")
    assert [d["num_instances"] for d in dd] == g["region_counts"]["via_powder_particle_masks_training.json"][:2] == [219, 351]
    d = dd[0]
    assert (d["height"], d["width"]) == (1024, 1536) and d["mask_format"] == "polygon" and d["dataset_class"] == "Train"
    assert d["image_id"] == 0 and d["annotation_file"] == jpath.name and d["HFW"].endswith("um")
    assert os.path.normpath(d["file_name"]).endswith(os.path.join("images_png", g["via"]["_via_img_metadata"][next(iter(g["via"]["_via_img_metadata"]))]["filename"]))
    a = d["annotations"][0]
    reg = next(iter(g["via"]["_via_img_metadata"].values()))["regions"][0]["shape_attributes"]
    assert a["category_id"] == 0 and a["bbox_mode"] == BoxMode.XYXY_ABS
    assert a["bbox"].tolist() == [min(reg["all_points_x"]), min(reg["all_points_y"]), max(reg["all_points_x"]), max(reg["all_points_y"])]
    assert a["segmentation"][0][:2] == [reg["all_points_x"][0] + 0.5, reg["all_points_y"][0] + 0.5]
    assert len(a["segmentation"]) == 1 and len(a["segmentation"][0]) == 2 * len(reg["all_points_x"])
    with pytest.raises(ValueError):
        data_utils.get_ddicts("coco", jpath)


def test_get_ddicts_binary_and_label_and_rle(tmp_path):
    from PIL import Image
    from ampis_amd import data_utils, rle
    img_dir, ann_dir = tmp_path / "images", tmp_path / "annotations"
    img_dir.mkdir(); ann_dir.mkdir()
    lab = np.zeros((40, 50), np.uint8)
    lab[2:10, 3:12] = 1
    lab[20:30, 25:45] = 1
    lab[10, 12] = 1            # touches the first blob diagonally: one instance under 8-connectivity (skimage default)
    Image.fromarray(np.zeros((40, 50, 3), np.uint8)).save(img_dir / "im1.png")
    Image.fromarray(lab * 255).save(ann_dir / "im1.png")
    dd = data_utils.get_ddicts("binary", img_dir, ann_dir)
    assert len(dd) == 1 and dd[0]["num_instances"] == 2 and dd[0]["mask_format"] == "bitmask" and (dd[0]["height"], dd[0]["width"]) == (40, 50)
    assert dd[0]["annotations"][0]["bbox"].tolist() == [3, 2, 12, 10] and dd[0]["annotations"][1]["bbox"].tolist() == [25, 20, 44, 29]
    m0 = rle.decode(dd[0]["annotations"][0]["segmentation"]).astype(bool)
    assert m0.sum() == 8 * 9 + 1
    np.save(ann_dir / "im1.npy", (lab * 3).astype(np.int32))
    os.remove(ann_dir / "im1.png")
    dl = data_utils.get_ddicts("label", img_dir, ann_dir)
    assert dl[0]["num_instances"] == 1 and dl[0]["annotations"][0]["bbox"].tolist() == [3, 2, 44, 29]      # one label value = one instance
    # rle json round trip
    segs = [{"size": a["segmentation"]["size"], "counts": a["segmentation"]["counts"].decode("utf-8")} for a in dd[0]["annotations"]]
    json.dump([{"file_name": "images/im1.png", "segmentations": segs}], open(tmp_path / "ann.json", "w"))
    dr = data_utils.get_ddicts("rle", tmp_path / "ann.json")
    assert dr[0]["num_instances"] == 2 and dr[0]["annotations"][1]["bbox"].tolist() == [25, 20, 44, 29]
    with pytest.raises(AssertionError):
        os.remove(ann_dir / "im1.npy")
        data_utils.get_ddicts("binary", img_dir, ann_dir)


def test_extract_boxes_modes():
    from ampis_amd.data_utils import extract_boxes
    m = np.zeros((3, 10, 12), bool)
    m[0, 2:5, 3:9] = True
    m[2, 9, 11] = True
    assert extract_boxes(m).tolist() == [[3, 2, 8, 4], [0, 0, 0, 0], [11, 9, 11, 9]]
    assert extract_boxes(m[0]).shape == (1, 4)
    assert extract_boxes(m.transpose(1, 2, 0), "matterport", "matterport").tolist() == [[2, 5, 3, 9], [0, 1, 0, 1], [9, 10, 11, 12]]


def test_frpyobjects_matches_oracle_rlefrpoly():
    from ampis_amd import rle
    from oracle import train as T
    rng = np.random.default_rng(0)
    for t in range(120):
        k = int(rng.integers(3, 40))
        h, w = int(rng.integers(5, 200)), int(rng.integers(5, 200))
        ang = np.sort(rng.uniform(0, 2 * np.pi, k))
        r = rng.uniform(0.2, 0.7, k) * min(h, w)
        xy = np.stack([w / 2 + r * np.cos(ang) * rng.uniform(0.5, 1.5), h / 2 + r * np.sin(ang)], 1).reshape(-1)
        if t % 5 == 0:
            xy = np.round(xy)
        got = rle.string_to_counts(rle.frPyObjects(xy.tolist(), h, w)["counts"])
        assert np.array_equal(got, T.fr_poly(xy, h, w))
    many = rle.frPyObjects([[1, 1, 8, 1, 8, 8, 1, 8], [2, 2, 5, 2, 5, 5]], 10, 10)
    assert len(many) == 2 and rle.area(many[0]) == 49


def test_det_seg_scores_reference_kat_and_oracle():
    from ampis_amd import analyze, rle
    from oracle import matcher, rle as orle
    blocks = []
    for (r, c) in ((0, 0), (0, 2), (2, 0), (2, 2)):
        m = np.zeros((4, 4), bool)
        m[r:r + 2, c:c + 2] = True
        blocks.append(m)
    gt = [rle.encode(m) for m in blocks]
    pred = [gt[2], gt[1], gt[3]]
    assert np.array_equal(analyze.iou_matrix(gt, pred), np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1]], float))   # analyze.py:719-722
    res = analyze.rle_instance_matcher(gt, pred)
    assert res["tp"].tolist() == [[1, 1], [2, 0], [3, 2]] and res["fn"].tolist() == [0] and res["fp"].tolist() == []       # analyze.py:725-728
    sc = analyze.det_seg_scores(gt, pred)
    assert sc["det_precision"] == 1.0 and sc["det_recall"] == 0.75 and sc["seg_tp"].tolist() == [4, 4, 4]
    # random overlapping masks, dense inputs, vs the oracle matcher
    rng = np.random.default_rng(1)
    g = rng.random((9, 30, 40)) > 0.7
    p = np.concatenate([g[:6] ^ (rng.random((6, 30, 40)) > 0.9), rng.random((3, 30, 40)) > 0.8])
    a = analyze.det_seg_scores(g, p, iou_thresh=0.5)
    ref = matcher.piecewise_rle_match([orle.encode(m) for m in g], [orle.encode(m) for m in p], orle.iou, 0.5)
    assert a["det_tp"].tolist() == ref["tp"].tolist() and np.allclose(a["det_tp_iou"], ref["iou"])
    for (gi, pi), tp, fn, fp in zip(a["det_tp"], a["seg_tp"], a["seg_fn"], a["seg_fp"]):
        assert tp == (g[gi] & p[pi]).sum() and fn == (g[gi] & ~p[pi]).sum() and fp == (~g[gi] & p[pi]).sum()
    # exact IoU ties, shared predictions, thresholds that sit ON an IoU, more than 80 masks a side (the reference's block size):
    # the product (array semantics, one C call) against the loop-for-loop restatement of the reference in oracle/matcher.py
    for trial in range(12):
        r = np.random.default_rng(100 + trial)
        H, W = 24, 36
        cells = [(y, x) for y in range(0, H, 6) for x in range(0, W, 6)]                     # 24 disjoint 6x6 cells
        def blob(ks):
            m = np.zeros((H, W), bool)
            for k in ks:
                y, x = cells[k]
                m[y:y + 6, x:x + 6] = True
            return m
        n_g, n_p = (int(r.integers(1, 30)), int(r.integers(1, 30))) if trial < 10 else (int(r.integers(90, 130)), int(r.integers(85, 170)))
        gm = [blob(r.choice(24, size=int(r.integers(1, 4)), replace=False)) for _ in range(n_g)]
        pm_ = [blob(r.choice(24, size=int(r.integers(1, 4)), replace=False)) for _ in range(n_p)]   # unions of whole cells: IoUs are k/n, ties everywhere
        pm_ += [m.copy() for m in pm_[:3]]                                                      # exact duplicates: first-maximum rule
        ge, pe = [orle.encode(m) for m in gm], [orle.encode(m) for m in pm_]
        for thr in (0.0, 0.25, 1 / 3, 0.5, 2 / 3, 0.999, 1.0):
            got = analyze.rle_instance_matcher(ge, pe, iou_thresh=thr)
            want = matcher.piecewise_rle_match(ge, pe, orle.iou, thr)
            for k in ("tp", "fn", "fp"):
                assert got[k].tolist() == want[k].tolist(), (trial, thr, k)
            assert np.array_equal(got["iou"], want["iou"])
            assert got["tp"].shape == (len(got["iou"]), 2)
        assert np.array_equal(analyze.iou_matrix(ge, pe), matcher.piecewise_iou(ge, pe, orle.iou))
    # no predictions: every ground truth is a false negative, and the detection precision is 0/0 as in the reference
    res = analyze.rle_instance_matcher(gt, [])
    assert res["tp"].shape == (0, 2) and res["fn"].tolist() == [0, 1, 2, 3] and res["fp"].tolist() == []
    with pytest.raises(ZeroDivisionError):
        analyze.det_seg_scores(gt, [])
    # polygon ground truth needs size
    from ampis_amd.structures import PolygonMasks
    pm = PolygonMasks([[[1, 1, 8, 1, 8, 8, 1, 8]]])
    r = analyze.masks_to_rle(pm, (10, 10))
    assert rle.area(r[0]) == 49
    with pytest.raises(AssertionError):
        analyze.masks_to_rle(pm)
