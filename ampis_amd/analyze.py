"""The evaluation metric AMPIS defines on the hot path's output -- "mask IoU vs ref" of the benchmark metric:
mirror of ampis/analyze.py:54-112 (_piecewise_iou), :115-181 (_piecewise_rle_match), :184-223 (rle_instance_matcher) and
:226-339 (det_seg_scores), same names / arguments / result keys, running on the C-ABI RLE codec (ampis_amd.rle).  Quirks kept on
purpose: strict `>` against iou_thresh, several ground-truth instances may match the same prediction, division by zero when
there are no detections."""
import numpy as np

from . import rle
from .structures import BitMasks, PolygonMasks, RLEBitMasks


def masks_to_rle(masks, size=None):
    """list of RLE dicts | RLEBitMasks | object with .rle | PolygonMasks (needs size=(h,w)) | BitMasks / bool ndarray [N,H,W]."""
    if isinstance(masks, RLEBitMasks) or hasattr(masks, "rle"):
        return list(masks.rle)
    if isinstance(masks, (list, tuple)) and (len(masks) == 0 or isinstance(masks[0], dict)):
        return list(masks)
    if isinstance(masks, PolygonMasks):
        assert size is not None, "size=(height, width) is required for polygon masks"
        return [rle.merge(rle.frPyObjects([np.asarray(p).reshape(-1).tolist() for p in inst], size[0], size[1])) for inst in masks.polygons]
    arr = masks.tensor.numpy() if isinstance(masks, BitMasks) else np.asarray(masks)
    if arr.ndim == 3:
        return [rle.encode(np.asfortranarray(m)) for m in arr.astype(bool)]
    raise NotImplementedError(f"unsupported mask type {type(masks)}")


def _piecewise_iou(a, b, interval=80):
    imax, jmax = len(a), len(b)
    target = np.zeros((imax, jmax))
    n_a = imax // interval + int(bool(imax % interval))
    n_b = jmax // interval + int(bool(jmax % interval))
    crowd = np.zeros(interval, bool)
    for i in range(n_a):
        i1, i2 = interval * i, min(interval * i + interval, imax)
        for j in range(n_b):
            j1, j2 = interval * j, min(interval * j + interval, jmax)
            target[i1:i2, j1:j2] = rle.iou(b[j1:j2], a[i1:i2], crowd[: i2 - i1]).T
    return target


def _piecewise_rle_match(gt, pred, iou_thresh=0.5, interval=80):
    jmax = len(pred)
    tp, fn, iou = [], [], []
    matched = np.zeros(len(pred), bool)
    n_seg = jmax // interval + int(jmax % interval > 0)
    for gi, g in enumerate(gt):
        best, arg = 0.0, -1
        for j in range(n_seg):
            j0 = interval * j
            s = rle.iou(pred[j0:j0 + interval], [g], [False])[:, 0]
            k = int(np.argmax(s))
            if s[k] > best:
                best, arg = s[k], k + j0
        if best > iou_thresh:
            tp.append([gi, arg])
            iou.append(best)
            matched[arg] = True
        else:
            fn.append(gi)
    fp = np.array([x for x, mm in enumerate(matched) if not mm], int)
    return {"tp": np.asarray(tp, int), "fn": np.asarray(fn, int), "fp": fp, "iou": np.asarray(iou)}


def rle_instance_matcher(gt, pred, iou_thresh=0.5, size=None):
    return _piecewise_rle_match(masks_to_rle(gt, size), masks_to_rle(pred, size), iou_thresh)


def det_seg_scores(gt, pred, iou_thresh=0.5, size=None):
    gtm, prm = masks_to_rle(gt, size), masks_to_rle(pred, size)
    res = rle_instance_matcher(gtm, prm, iou_thresh=iou_thresh, size=size)
    matches = np.asarray(res["tp"])
    tp, fn, fp = len(matches), len(res["fn"]), len(res["fp"])
    det_precision = tp / (tp + fp)
    det_recall = tp / (tp + fn)
    g_tp = [gtm[i[0]] for i in matches]
    p_tp = [prm[i[1]] for i in matches]
    seg_tp = np.array([rle.area(rle.merge([a, b], intersect=True)) for a, b in zip(g_tp, p_tp)], np.int64)
    ga = np.array([rle.area(m) for m in g_tp], np.int64)
    pa = np.array([rle.area(m) for m in p_tp], np.int64)
    seg_fp, seg_fn = pa - seg_tp, ga - seg_tp
    return {"det_precision": det_precision, "det_recall": det_recall, "seg_precision": seg_tp / (seg_tp + seg_fp),
            "seg_recall": seg_tp / (seg_tp + seg_fn), "det_tp": matches, "det_fn": res["fn"], "det_fp": res["fp"], "seg_tp": seg_tp,
            "seg_fn": seg_fn, "seg_fp": seg_fp, "det_tp_iou": res["iou"]}
