"""ORACLE (test infrastructure, NOT product code) — restatement of AMPIS's RLE instance matcher, the code that defines
"mask IoU vs ref" (ampis/analyze.py:54-112 `_piecewise_iou`, :115-181 `_piecewise_rle_match`).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Pinned by the reference's own known-answer test, ampis/analyze.py:702-728 (tests/test_matcher.py).
"""
import numpy as np


def piecewise_iou(a, b, iou_fn, interval=80):
    """analyze.py:54-112: len(a) x len(b) IoU matrix computed in interval x interval blocks of rle.iou(b_blk, a_blk).T."""
    imax, jmax = len(a), len(b)
    target = np.zeros((imax, jmax))
    n_a = imax // interval + int(bool(imax % interval))
    n_b = jmax // interval + int(bool(jmax % interval))
    crowd = np.zeros(interval, bool)
    for i in range(n_a):
        i1, i2 = interval * i, min(interval * i + interval, imax)
        for j in range(n_b):
            j1, j2 = interval * j, min(interval * j + interval, jmax)
            target[i1:i2, j1:j2] = iou_fn(b[j1:j2], a[i1:i2], crowd[: i2 - i1]).T
    return target


def piecewise_rle_match(gt, pred, iou_fn, iou_thresh=0.5, interval=80):
    """analyze.py:115-181: each gt takes the prediction of maximal IoU (first maximum, strict > over chunks);
    it is a match iff that IoU > iou_thresh (strict).  Several gts may share a prediction (no exclusivity)."""
    jmax = len(pred)
    tp, fn, ious = [], [], []
    matched = np.zeros(jmax, bool)
    n_seg = jmax // interval + int(jmax % interval > 0)
    for gi, g in enumerate(gt):
        best, arg = 0.0, -1
        for j in range(n_seg):
            j0 = interval * j
            s = iou_fn(pred[j0:j0 + interval], [g], [False])[:, 0]
            k = int(np.argmax(s))
            if s[k] > best:
                best, arg = s[k], k + j0
        if best > iou_thresh:
            tp.append([gi, arg])
            ious.append(best)
            matched[arg] = True
        else:
            fn.append(gi)
    fp = np.array([x for x, mm in enumerate(matched) if not mm], int)
    return {"tp": np.asarray(tp, int), "fn": np.asarray(fn, int), "fp": fp, "iou": np.asarray(ious)}
