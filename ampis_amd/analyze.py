"""Instance matching and detection / segmentation scores on RLE masks -- the "mask IoU vs ref" half of the benchmark metric
(SURVEY.md §8 f3).  Same entry points, arguments and result keys as ampis/analyze.py (`rle_instance_matcher` :184-223,
`det_seg_scores` :226-339), so notebooks call it unchanged; the computation is laid out for the C-ABI RLE codec instead of for
pycocotools:

  * ONE `amp_rle_iou_matrix` call gives every (ground truth, prediction) IoU.  (The reference walks 80-wide blocks because
    pycocotools.mask.iou has a size limit, ampis/analyze.py:54-112; the C routine has none, so there is nothing to walk.)
  * The matching rule is stated as array semantics rather than as a loop:
      - a ground-truth instance g is assigned the prediction of maximal IoU, the FIRST one among equals (lowest index);
      - g is a true positive iff that IoU is strictly greater than `iou_thresh`, otherwise a false negative;
      - assignment is not exclusive: two ground-truth instances may take the same prediction (a prediction covering two particles
        counts as two true positives -- the reference's behaviour);
      - a prediction no ground-truth instance took is a false positive.
  * Pixel-level scores come from ONE `amp_rle_pair_overlap` call over the matched pairs (|g and p|, |g minus p|, |p minus g|) instead
    of a merge + three area calls per pair.
With no prediction at all the detection precision is 0/0: like the reference this raises ZeroDivisionError.
The independent checker is oracle/matcher.py (a loop-for-loop restatement of the reference, pinned by its known-answer test)."""
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


def iou_matrix(gt, pred):
    """[len(gt), len(pred)] float64 IoU of every ground-truth / prediction pair (no crowd regions), one C call."""
    if len(gt) == 0 or len(pred) == 0:
        return np.zeros((len(gt), len(pred)))
    return np.ascontiguousarray(rle.iou(pred, gt, np.zeros(len(gt), bool)).T)


def match_instances(iou, iou_thresh=0.5):
    """The matching rule of the module docstring on a [G, P] IoU matrix -> {'tp': [n,2] (gt, pred) index pairs in gt order,
    'fn': gt indices, 'fp': pred indices, 'iou': IoU of each true positive}."""
    assert iou_thresh >= 0, "iou_thresh must not be negative"
    G, P = iou.shape
    if P == 0:
        return {"tp": np.zeros((0, 2), int), "fn": np.arange(G), "fp": np.zeros(0, int), "iou": np.zeros(0)}
    taken = iou.argmax(axis=1)                               # first maximum of each row
    best = iou[np.arange(G), taken]
    hit = best > iou_thresh                                  # strict
    unclaimed = np.ones(P, bool)
    unclaimed[taken[hit]] = False
    return {"tp": np.stack([np.flatnonzero(hit), taken[hit]], axis=1).astype(int).reshape(-1, 2), "fn": np.flatnonzero(~hit),
            "fp": np.flatnonzero(unclaimed), "iou": best[hit]}


def rle_instance_matcher(gt, pred, iou_thresh=0.5, size=None):
    gt, pred = masks_to_rle(gt, size), masks_to_rle(pred, size)
    return match_instances(iou_matrix(gt, pred), iou_thresh)


def det_seg_scores(gt, pred, iou_thresh=0.5, size=None):
    gt, pred = masks_to_rle(gt, size), masks_to_rle(pred, size)
    m = match_instances(iou_matrix(gt, pred), iou_thresh)
    n_tp, n_fn, n_fp = len(m["tp"]), len(m["fn"]), len(m["fp"])
    both, gt_only, pred_only = rle.pair_overlap(gt, pred, m["tp"])       # per matched pair: pixels in both / missed / spurious
    with np.errstate(divide="ignore", invalid="ignore"):
        seg_precision, seg_recall = both / (both + pred_only), both / (both + gt_only)
    return {"det_precision": n_tp / (n_tp + n_fp), "det_recall": n_tp / (n_tp + n_fn), "seg_precision": seg_precision, "seg_recall": seg_recall,
            "det_tp": m["tp"], "det_fn": m["fn"], "det_fp": m["fp"], "seg_tp": both, "seg_fn": gt_only, "seg_fp": pred_only, "det_tp_iou": m["iou"]}
