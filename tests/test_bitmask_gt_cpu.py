"""Host side of bitmask ground truth (no GPU): the mapper's INPUT.MASK_FORMAT='bitmask' transform against a dense restatement of what
detectron2 does to each instance mask (decode -> PIL NEAREST resize with the image -> mirror -> empty instances dropped), the
flattening of several-polygon and bitmask instances into amp_gt (PackedGt), and the oracle's two target builders on cases whose
answer is known in closed form."""
import numpy as np
import pytest


def _dense_reference(annos, h, w, nh, nw, flip):
    from PIL import Image
    from oracle import rle as orle
    out = []
    for a in annos:
        seg = a["segmentation"]
        assert isinstance(seg, dict)
        m = orle.decode(seg).astype(np.uint8)
        m = np.asarray(Image.fromarray(m).resize((nw, nh), Image.NEAREST))
        if flip:
            m = m[:, ::-1]
        b = np.asarray(a["bbox"], float).copy()
        b[0::2] *= nw / w; b[1::2] *= nh / h
        if flip:
            b[0], b[2] = nw - b[2], nw - b[0]
        b = np.clip(b, 0, [nw, nh, nw, nh])
        if b[2] - b[0] <= 1e-5 or b[3] - b[1] <= 1e-5 or not m.any():
            continue
        out.append((b.astype(np.float32), m.astype(bool)))
    return out


def test_bitmask_transform_matches_dense_resize_and_flip():
    from ampis_amd import data, rle
    rng = np.random.default_rng(0)
    h, w = 60, 90
    annos = []
    for i in range(7):
        m = np.zeros((h, w), bool)
        y0, x0 = rng.integers(0, h - 12), rng.integers(0, w - 12)
        m[y0:y0 + rng.integers(2, 12), x0:x0 + rng.integers(2, 12)] = True
        m &= rng.random((h, w)) > 0.2
        ys, xs = np.nonzero(m)
        annos.append({"bbox": [xs.min(), ys.min(), xs.max(), ys.max()], "bbox_mode": 0, "category_id": 0, "segmentation": rle.encode(np.asfortranarray(m))})
    annos.append({"bbox": [5, 5, 6, 6], "bbox_mode": 0, "category_id": 0, "segmentation": rle.encode(np.asfortranarray(np.zeros((h, w), bool)))})   # empty mask: dropped
    annos.append({"bbox": [5, 5, 9, 9], "bbox_mode": 0, "category_id": 0, "iscrowd": 1, "segmentation": annos[0]["segmentation"]})                    # crowd: skipped
    for (nh, nw, flip) in ((60, 90, False), (60, 90, True), (80, 120, True), (37, 55, False)):
        got = data.transform_annotations_bitmask(annos, h, w, nh, nw, flip)
        ref = _dense_reference(annos[:-1], h, w, nh, nw, flip)
        assert len(got["masks_rle"]) == len(ref) == len(got["boxes"]) and got["polygons"] == [None] * len(ref)
        for r, b, (rb, rm) in zip(got["masks_rle"], got["boxes"], ref):
            assert r["size"] == [nh, nw] and np.array_equal(rle.decode(r).astype(bool), rm) and np.array_equal(b, rb)
    # polygon segmentations under the bitmask format: rasterised at the NEW size after the vertices were scaled and mirrored
    pa = [{"bbox": [10, 10, 40, 30], "bbox_mode": 0, "category_id": 0, "segmentation": [[10, 10, 40, 10, 40, 30, 10, 30], [50, 40, 60, 40, 60, 50, 50, 50]]}]
    got = data.transform_annotations_bitmask(pa, h, w, 120, 180, True)
    m = rle.decode(got["masks_rle"][0]).astype(bool)
    assert m.shape == (120, 180) and m[40, 180 - 50] and m[90, 180 - 110] and not m[40, 50] and m.sum() == pytest.approx(60 * 40 + 20 * 20, rel=0.05)


def test_polygon_format_refuses_rle_and_takes_several_polygons():
    from ampis_amd import data, rle
    rl = rle.encode(np.asfortranarray(np.ones((8, 8), bool)))
    with pytest.raises(ValueError, match="bitmask"):
        data.transform_annotations([{"bbox": [0, 0, 8, 8], "bbox_mode": 0, "category_id": 0, "segmentation": rl}], 1.0, 1.0, False, 8, 8)
    two = [{"bbox": [0, 0, 30, 30], "bbox_mode": 0, "category_id": 0, "segmentation": [[1, 1, 9, 1, 9, 9, 1, 9], [20, 20, 29, 20, 29, 29]]},
           {"bbox": [2, 2, 6, 6], "bbox_mode": 0, "category_id": 1, "segmentation": [[2, 2, 6, 2, 6, 6, 2, 6]]}]
    g = data.transform_annotations(two, 2.0, 2.0, True, 100, 100)
    assert isinstance(g["polygons"][0], list) and len(g["polygons"][0]) == 2 and g["polygons"][0][1].tolist() == [60, 40, 42, 40, 42, 58]
    assert not isinstance(g["polygons"][1], list) and g["polygons"][1].tolist() == [96, 4, 88, 4, 88, 12, 96, 12]


def test_packed_gt_layout_for_polygon_ranges_and_runs():
    """amp_gt as include/ampis_hip.h describes it: poly_off per POLYGON + inst_poly_off per instance once an instance has != 1 polygons;
    rle_off / rle_counts / rle_hw for bitmask instances; the plain one-polygon form stays as it was (inst_poly_off NULL)."""
    from ampis_amd import rle
    from ampis_amd.model import PackedGt
    sq = lambda x, y, s: np.array([x, y, x + s, y, x + s, y + s, x, y + s], float)
    plain = PackedGt([dict(boxes=np.zeros((2, 4), np.float32), classes=np.zeros(2), polygons=[sq(0, 0, 4), sq(5, 5, 3)])])
    assert not plain.struct.inst_poly_off and not plain.struct.rle_off and [plain.struct.poly_off[i] for i in range(3)] == [0, 8, 16]
    m = np.zeros((6, 5), bool); m[1:4, 2:4] = True
    r = rle.encode(np.asfortranarray(m))
    mixed = PackedGt([dict(boxes=np.zeros((3, 4), np.float32), classes=np.zeros(3), polygons=[[sq(0, 0, 4), sq(9, 9, 2)], None, sq(1, 1, 2)], masks_rle=[None, r, None]),
                      dict(boxes=np.zeros((1, 4), np.float32), classes=np.zeros(1), polygons=[None], masks_rle=[r])])
    s = mixed.struct
    assert [s.gt_off[i] for i in range(3)] == [0, 3, 4]
    assert [s.inst_poly_off[i] for i in range(5)] == [0, 2, 2, 3, 3] and [s.poly_off[i] for i in range(4)] == [0, 8, 16, 24]
    cnt = rle.string_to_counts(r["counts"])
    assert [int(s.rle_off[i]) for i in range(5)] == [0, 0, len(cnt), len(cnt), 2 * len(cnt)]
    assert [s.rle_counts[i] for i in range(2 * len(cnt))] == cnt.tolist() * 2
    assert [s.rle_hw[i] for i in range(8)] == [0, 0, 6, 5, 0, 0, 6, 5]


def test_oracle_target_builders_on_closed_form_cases():
    """BitMasks.crop_and_resize of a box that IS the pixel grid of a 28x28 mask region returns that region; the union of two polygons is
    the OR of their rasters."""
    from oracle import train as T
    rng = np.random.default_rng(1)
    m = np.zeros((50, 70), bool)
    m[10:38, 20:48] = rng.random((28, 28)) > 0.5
    out = T.bitmask_crop_and_resize(m, [20, 10, 48, 38], 28)           # one sample per bin, at the pixel centres
    assert np.array_equal(out, m[10:38, 20:48])
    out2 = T.bitmask_crop_and_resize(m, [20, 10, 76, 66], 28)          # 2x2 samples per bin over 2x2 pixels: mean >= 0.5
    assert out2.shape == (28, 28) and not out2[20:, :].any()
    a = np.array([2, 2, 12, 2, 12, 12, 2, 12], float); b = np.array([15, 15, 25, 15, 25, 25, 15, 25], float)
    box = [0, 0, 28, 28]
    assert np.array_equal(T.rasterize_polygon_within_box([a, b], box, 28), T.rasterize_polygon_within_box(a, box, 28) | T.rasterize_polygon_within_box(b, box, 28))


def test_rle_domain_resize_is_pil_nearest_resize_and_flip():
    """amp_rle_resize_nearest against PIL itself (the call detectron2's ResizeTransform.apply_segmentation makes) on random, empty, full,
    blocky and sparse masks, up- and down-scaling with awkward ratios, 1-pixel sizes, with and without the mirror: same pixels, same
    counts string as encode(resize(decode))."""
    from PIL import Image
    from ampis_amd import rle
    rng = np.random.default_rng(0)
    for trial in range(150):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 90))
        nh, nw = int(rng.integers(1, 140)), int(rng.integers(1, 160))
        kind = trial % 5
        m = (rng.random((h, w)) > (0.5, 2.0, -1.0, 0.5, 0.9)[kind])
        if kind == 3:
            m = np.zeros((h, w), bool); m[h // 4:max(h // 4 + 1, 3 * h // 4), w // 3:max(w // 3 + 1, 2 * w // 3)] = True
        r = rle.encode(np.asfortranarray(m))
        for flip in (False, True):
            ref = np.asarray(Image.fromarray(m.astype(np.uint8)).resize((nw, nh), Image.NEAREST)).astype(bool)
            ref = ref[:, ::-1] if flip else ref
            got = rle.resize_nearest(r, nh, nw, flip)
            assert got["size"] == [nh, nw] and got["counts"] == rle.encode(np.asfortranarray(ref))["counts"], (h, w, nh, nw, flip, kind)
    m = np.zeros((1024, 1536), bool); m[300:700, 200:900] = True; m[::97, ::89] = True
    got = rle.resize_nearest(rle.encode(np.asfortranarray(m)), 800, 1200, True)
    ref = np.asarray(Image.fromarray(m.astype(np.uint8)).resize((1200, 800), Image.NEAREST)).astype(bool)[:, ::-1]
    assert np.array_equal(rle.decode(got).astype(bool), ref)
