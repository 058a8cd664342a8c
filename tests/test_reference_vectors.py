"""The evaluation harness against outputs OF THE REFERENCE ITSELF (tests/golden/reference_vectors.json.gz, made by
tests/golden/make_reference_vectors.py in the build container: ampis.analyze imported unmodified, run on seeded RLE sets).  Runs anywhere:
the fixture carries the inputs (counts strings) and what ampis/analyze.py:54-339 returned for them.  Both the product (ampis_amd.analyze)
and the oracle restatement (oracle/matcher.py on oracle/rle.py) must reproduce them."""
import base64
import gzip
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_vectors.json.gz")


def _cases():
    with gzip.open(GOLD, "rt") as f:
        return json.load(f)["cases"]


def _rles(case, key):
    return [{"size": list(case["size"]), "counts": base64.b64decode(s)} for s in case[key]]


def test_product_reproduces_the_reference_matcher_and_scores():
    from ampis_amd import analyze
    cases = _cases()
    assert len(cases) == 10 and max(len(c["gt"]) for c in cases) > 80 and max(len(c["pred"]) for c in cases) > 80
    for ci, c in enumerate(cases):
        g, p = _rles(c, "gt"), _rles(c, "pred")
        assert np.array_equal(analyze.iou_matrix(g, p), np.asarray(c["iou"])), ci
        for thr_s, want in c["match"].items():
            got = analyze.rle_instance_matcher(g, p, iou_thresh=float(thr_s))
            for k in ("tp", "fn", "fp"):
                assert np.asarray(got[k]).reshape(-1).tolist() == [int(v) for v in want[k]], (ci, thr_s, k)
            assert np.array_equal(np.asarray(got["iou"], float).reshape(-1), np.asarray(want["iou"], float)), (ci, thr_s)
        sc = analyze.det_seg_scores(g, p, size=tuple(c["size"]))
        assert set(sc) == set(c["scores"])
        for k, want in c["scores"].items():
            assert np.allclose(np.asarray(sc[k], float).reshape(-1), np.asarray(want, float), rtol=0, atol=0, equal_nan=True), (ci, k)


def test_oracle_restatement_reproduces_the_reference_matcher():
    from oracle import matcher, rle as orle
    for ci, c in enumerate(_cases()):
        g, p = _rles(c, "gt"), _rles(c, "pred")
        assert np.array_equal(matcher.piecewise_iou(g, p, orle.iou), np.asarray(c["iou"])), ci
        for thr_s, want in c["match"].items():
            got = matcher.piecewise_rle_match(g, p, orle.iou, float(thr_s))
            for k in ("tp", "fn", "fp"):
                assert np.asarray(got[k]).reshape(-1).tolist() == [int(v) for v in want[k]], (ci, thr_s, k)
            assert np.array_equal(np.asarray(got["iou"], float), np.asarray(want["iou"], float))


def test_product_reproduces_the_reference_via2_ingestion(tmp_path):
    """ampis.data_utils.get_ddicts('via2') as the reference ran it on the two-image cut of its own VIA project (tests/golden/via_subset.json)."""
    from ampis_amd import data_utils
    with gzip.open(GOLD, "rt") as f:
        want = json.load(f)["via2"]
    sub = json.load(open(os.path.join(os.path.dirname(GOLD), "via_subset.json")))["via"]
    (tmp_path / "via_2.0.8").mkdir()
    jp = tmp_path / "via_2.0.8" / "subset.json"
    jp.write_text(json.dumps(sub))
    cwd = os.getcwd()
    os.chdir(jp.parent)
    try:
        got = data_utils.get_ddicts("via2", "subset.json", dataset_class="Train")
    finally:
        os.chdir(cwd)
    assert len(got) == len(want) == 2
    for g, w in zip(got, want):
        assert set(g) == set(w)
        for k in w:
            if k != "annotations":
                assert g[k] == w[k], (k, g[k], w[k])
        assert len(g["annotations"]) == len(w["annotations"]) > 100
        for a, b in zip(g["annotations"], w["annotations"]):
            assert np.array_equal(np.asarray(a["bbox"], float), np.asarray(b["bbox"], float)) and int(a["bbox_mode"]) == b["bbox_mode"]
            assert int(a["category_id"]) == b["category_id"] and np.array_equal(np.asarray(a["segmentation"][0], float), np.asarray(b["segmentation"][0], float))


def test_product_reproduces_the_reference_output_container():
    """format_outputs / compress_pred of the reference (ampis/data_utils.py:255-310) on dense masks: same container, same RLE bytes."""
    import torch
    from ampis_amd import data_utils, rle
    from ampis_amd.structures import Boxes, Instances
    with gzip.open(GOLD, "rt") as f:
        want = json.load(f)["container"]
    src = want["inputs"]
    dense = np.stack([rle.decode({"size": [1024, 1536], "counts": base64.b64decode(c)}).astype(bool) for c in src["masks"]])
    inst = Instances((1024, 1536), pred_boxes=Boxes(torch.as_tensor(np.asarray(src["boxes"], np.float32))), scores=torch.as_tensor(np.asarray(src["scores"], np.float32)),
                     pred_classes=torch.as_tensor(np.asarray(src["classes"], np.int64)), pred_masks=torch.as_tensor(dense))
    res = data_utils.format_outputs("x.png", "d_Train", {"instances": inst})
    ri = res["pred"]["instances"]
    assert res["file_name"] == want["file_name"] and res["dataset"] == want["dataset"] and list(ri.image_size) == want["image_size"]
    assert sorted(ri._fields) == want["fields"]
    assert np.asarray(ri.pred_boxes).dtype == np.dtype(want["boxes_dtype"]) and np.array_equal(np.asarray(ri.pred_boxes, float), np.asarray(want["pred_boxes"]))
    assert np.array_equal(np.asarray(ri.scores, float), np.asarray(want["scores"]))
    assert np.asarray(ri.pred_classes).dtype == np.dtype(want["classes_dtype"]) and np.asarray(ri.pred_classes).tolist() == want["classes"]
    assert [{"size": list(m["size"]), "counts": base64.b64encode(m["counts"]).decode()} for m in ri.pred_masks] == want["masks"]
