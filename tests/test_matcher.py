"""The reference's only numeric known-answer test (ampis/analyze.py:702-728) with its own vectors: four 2x2 corner
blocks m1..m4 of a 4x4 image; gt = [m1, m2, m3, m4], pred = [m3, m2, m4]; expected IoU matrix analyze.py:719-722,
expected tp/fn/fp/iou analyze.py:725-728."""
import numpy as np

from oracle import matcher, rle as orle


def _blocks():
    ms = []
    for (r, c) in ((0, 0), (0, 2), (2, 0), (2, 2)):
        m = np.zeros((4, 4), bool)
        m[r:r + 2, c:c + 2] = True
        ms.append(m)
    return ms


def _check(enc, iou_fn):
    b = _blocks()
    gt = [enc(m) for m in b]
    pred = [enc(b[2]), enc(b[1]), enc(b[3])]
    iou = matcher.piecewise_iou(gt, pred, iou_fn)
    expect = np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1]], float)
    assert np.array_equal(iou, expect)
    res = matcher.piecewise_rle_match(gt, pred, iou_fn)
    assert res["tp"].tolist() == [[1, 1], [2, 0], [3, 2]]
    assert res["fn"].tolist() == [0]
    assert res["fp"].tolist() == []
    assert res["iou"].tolist() == [1.0, 1.0, 1.0]


def test_kat_with_oracle_codec():
    _check(orle.encode, orle.iou)


def test_kat_with_product_codec():
    from ampis_amd import rle as prle
    _check(prle.encode, prle.iou)


def test_interval_chunking_matches_unchunked():
    rng = np.random.default_rng(0)
    masks = [rng.random((12, 9)) > 0.6 for _ in range(23)]
    a = [orle.encode(m) for m in masks[:11]]
    b = [orle.encode(m) for m in masks[11:]]
    full = matcher.piecewise_iou(a, b, orle.iou, interval=80)
    assert np.allclose(full, matcher.piecewise_iou(a, b, orle.iou, interval=4))
