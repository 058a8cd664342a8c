"""TEST INFRASTRUCTURE (see oracle/__init__): numpy restatement of Pillow's ImagingResample for uint8 images and the BILINEAR
filter -- what detectron2's ResizeTransform.apply_image runs under DefaultPredictor (`Image.resize((w, h), Image.BILINEAR)`,
SURVEY §8a row a7).  Pinned: bit-exact against PIL itself (tests/test_resize.py), which is importable here."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc (bilinear: support 1, widened by the down-scale factor)."""
    scale = in_size / out_size
    filterscale = max(1.0, scale)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds, kk = [], np.zeros((out_size, ksize), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        k = []
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            k.append(1.0 - t if t < 1.0 else 0.0)
        ww = sum(k)
        for x in range(xmax):
            v = k[x] / ww if ww != 0.0 else k[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds.append((xmin, xmax))
    return bounds, kk


def resize_bilinear_u8(img, h, w):
    """img uint8 [H,W,C] -> [h,w,C]: horizontal pass, then vertical pass on its uint8 result (Pillow's order)."""
    H, W = img.shape[:2]
    cur = img.astype(np.int64)
    half = 1 << (PRECISION_BITS - 1)
    if w != W:
        b, kk = coeffs(W, w)
        out = np.zeros((H, w, img.shape[2]), np.int64)
        for xx, (xmin, n) in enumerate(b):
            out[:, xx, :] = np.clip((half + (cur[:, xmin:xmin + n, :] * kk[xx, :n][None, :, None]).sum(1)) >> PRECISION_BITS, 0, 255)
        cur = out
    if h != H:
        b, kk = coeffs(H, h)
        out = np.zeros((h, cur.shape[1], img.shape[2]), np.int64)
        for yy, (ymin, n) in enumerate(b):
            out[yy] = np.clip((half + (cur[ymin:ymin + n, :, :] * kk[yy, :n][:, None, None]).sum(0)) >> PRECISION_BITS, 0, 255)
        cur = out
    return cur.astype(np.uint8)
