"""COCO RLE codec on the C ABI (amp_rle_* of libampis_hip.so, host C++): the drop-in for the pycocotools.mask
calls AMPIS makes around the hot path (ampis/data_utils.py:275; ampis/analyze.py:108,158,315-321;
ampis/structures.py:465-468,568,752).  RLE dicts are pycocotools' compressed form {'size':[h,w], 'counts': bytes}."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def counts_to_string(cnts):
    cnts = np.ascontiguousarray(cnts, dtype=np.uint32)
    cap = 7 * len(cnts) + 8
    buf = C.create_string_buffer(cap)
    n = C.c_size_t()
    check(lib().amp_rle_to_string(cnts.ctypes.data_as(C.c_void_p), len(cnts), buf, cap, C.byref(n)), "amp_rle_to_string")
    return buf.raw[: n.value]


def counts_to_strings(pool, off, ln):
    """RLE strings of many masks in one call: mask i = pool[off[i] : off[i] + ln[i]] (uint32 run lengths)."""
    n = len(off)
    if n == 0:
        return []
    pool = np.ascontiguousarray(pool, dtype=np.uint32)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    ln = np.ascontiguousarray(ln, dtype=np.int32)
    cap = 7 * int(ln.sum()) + 8 * n + 8
    buf = C.create_string_buffer(cap)
    so = np.zeros(n + 1, dtype=np.uintp)
    check(lib().amp_rle_to_strings(pool.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), n,
                                   buf, cap, so.ctypes.data_as(C.c_void_p)), "amp_rle_to_strings")
    raw = buf.raw
    so = so.tolist()
    return [raw[so[i]: so[i + 1]] for i in range(n)]


def string_to_counts(s):
    if isinstance(s, str):
        s = s.encode("ascii")
    out = np.empty(len(s) + 1, dtype=np.uint32)
    m = C.c_int()
    check(lib().amp_rle_from_string(C.c_char_p(s), len(s), out.ctypes.data_as(C.c_void_p), len(out), C.byref(m)),
          "amp_rle_from_string")
    return out[: m.value].copy()


def _counts(r):
    c = r["counts"]
    return string_to_counts(c) if isinstance(c, (bytes, str)) else np.ascontiguousarray(c, dtype=np.uint32)


def encode(mask):
    """mask: [H,W] (or [H,W,N]) bool/uint8 -> RLE dict (or list), like pycocotools.mask.encode."""
    mask = np.asarray(mask)
    if mask.ndim == 3:
        return [encode(mask[:, :, i]) for i in range(mask.shape[2])]
    h, w = mask.shape
    f = np.asfortranarray(mask.astype(np.uint8))
    cap = h * w + 2
    out = np.empty(cap, dtype=np.uint32)
    m = C.c_int()
    check(lib().amp_rle_encode(f.ctypes.data_as(C.c_void_p), h, w, out.ctypes.data_as(C.c_void_p), cap, C.byref(m)),
          "amp_rle_encode")
    return {"size": [h, w], "counts": counts_to_string(out[: m.value])}


def decode(r):
    if isinstance(r, (list, tuple)):
        return np.stack([decode(x) for x in r], axis=2)
    h, w = r["size"]
    c = _counts(r)
    out = np.empty((h, w), dtype=np.uint8, order="F")
    check(lib().amp_rle_decode(c.ctypes.data_as(C.c_void_p), len(c), h, w, out.ctypes.data_as(C.c_void_p)), "amp_rle_decode")
    return out


def area(r):
    if isinstance(r, (list, tuple)):
        return np.array([area(x) for x in r], dtype=np.uint32)
    c = _counts(r)
    a = C.c_ulonglong()
    check(lib().amp_rle_area(c.ctypes.data_as(C.c_void_p), len(c), C.byref(a)), "amp_rle_area")
    return int(a.value)


def _pool(counts_list):
    off = np.zeros(len(counts_list), dtype=np.uint64)
    ln = np.asarray([len(c) for c in counts_list], dtype=np.int32)
    if len(counts_list):
        off[1:] = np.cumsum(ln[:-1], dtype=np.uint64)
    pool = np.ascontiguousarray(np.concatenate(counts_list) if len(counts_list) else np.zeros(1, np.uint32), dtype=np.uint32)
    return pool, off, ln


def iou(dt, gt, iscrowd):
    """len(dt) x len(gt) float64 matrix, like pycocotools.mask.iou on RLE lists (one C call: amp_rle_iou_matrix)."""
    out = np.zeros((len(dt), len(gt)), dtype=np.float64)
    if len(dt) == 0 or len(gt) == 0:
        return out
    dp, do, dl = _pool([_counts(x) for x in dt])
    gp, go, gl = _pool([_counts(x) for x in gt])
    crowd = np.ascontiguousarray([int(bool(c)) for c in iscrowd], dtype=np.uint8) if len(iscrowd) else None
    assert crowd is None or len(crowd) == len(gt)
    h = int(dt[0]["size"][0]) if isinstance(dt[0], dict) and "size" in dt[0] else 0
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    check(lib().amp_rle_iou_matrix(vp(dp), vp(do), vp(dl), len(dt), vp(gp), vp(go), vp(gl), len(gt),
                                   vp(crowd) if crowd is not None else None, h, vp(out)), "amp_rle_iou_matrix")
    return out


def resize_nearest(r, new_h, new_w, flip=False):
    """RLE of flip(PIL.Image.resize(decode(r), (new_w, new_h), NEAREST)) computed on the runs (amp_rle_resize_nearest): what detectron2 does to a
    bitmask annotation under ResizeShortestEdge + RandomFlip, without decoding."""
    h, w = int(r["size"][0]), int(r["size"][1])
    c = _counts(r)
    cap = 2 * int(new_w) + 2 * len(c) * max(1, -(-int(new_w) // max(w, 1))) + 8      # every source transition repeats for each output column that reads its column
    while True:
        out = np.empty(cap, dtype=np.uint32)
        m = C.c_int()
        st = lib().amp_rle_resize_nearest(c.ctypes.data_as(C.c_void_p), len(c), h, w, int(new_h), int(new_w), int(bool(flip)),
                                          out.ctypes.data_as(C.c_void_p), cap, C.byref(m))
        if st == 0:
            return {"size": [int(new_h), int(new_w)], "counts": counts_to_string(out[: m.value])}
        if cap >= int(new_h) * int(new_w) + 2:
            check(st, "amp_rle_resize_nearest")
        cap = min(cap * 4, int(new_h) * int(new_w) + 2)


def pair_overlap(a, b, pairs):
    """For index pairs (i, j): |a[i] AND b[j]|, |a[i] minus b[j]|, |b[j] minus a[i]| as three int64 arrays, one C call
    (amp_rle_pair_overlap)."""
    pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
    n = len(pairs)
    out = [np.zeros(n, dtype=np.uint64) for _ in range(3)]
    if n:
        ap, ao, al = _pool([_counts(x) for x in a])
        bp, bo, bl = _pool([_counts(x) for x in b])
        assert pairs[:, 0].max() < len(a) and pairs[:, 1].max() < len(b) and pairs.min() >= 0
        pa, pb = np.ascontiguousarray(pairs[:, 0]), np.ascontiguousarray(pairs[:, 1])
        vp = lambda x: x.ctypes.data_as(C.c_void_p)
        check(lib().amp_rle_pair_overlap(vp(ap), vp(ao), vp(al), vp(bp), vp(bo), vp(bl), vp(pa), vp(pb), n, vp(out[0]), vp(out[1]), vp(out[2])),
              "amp_rle_pair_overlap")
    return tuple(o.astype(np.int64) for o in out)


def merge(rles, intersect=False):
    assert len(rles) >= 1
    h, w = rles[0]["size"]
    cur = _counts(rles[0])
    for r in rles[1:]:
        b = _counts(r)
        cap = len(cur) + len(b) + 2
        out = np.empty(cap, dtype=np.uint32)
        m = C.c_int()
        check(lib().amp_rle_merge2(cur.ctypes.data_as(C.c_void_p), len(cur), b.ctypes.data_as(C.c_void_p), len(b),
                                   int(bool(intersect)), out.ctypes.data_as(C.c_void_p), cap, C.byref(m)), "amp_rle_merge2")
        cur = out[: m.value].copy()
    return {"size": [h, w], "counts": counts_to_string(cur)}


def frPyObjects(polys, h, w):
    """Polygons -> RLE, like pycocotools.mask.frPyObjects for polygon input: `polys` is a list of flat [x0,y0,x1,y1,...]
    polygons (or one such flat list); returns a list of RLE dicts (or one dict)."""
    single = len(polys) > 0 and np.isscalar(polys[0])
    plist = [polys] if single else polys
    out = []
    for p in plist:
        xy = np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape(-1))
        k = len(xy) // 2
        cap = 2 * 5 * (4 * (h + w) + 8 * k) + 16
        buf = np.empty(cap, dtype=np.uint32)
        m = C.c_int()
        check(lib().amp_rle_from_polygon(xy.ctypes.data_as(C.c_void_p), k, int(h), int(w), buf.ctypes.data_as(C.c_void_p), cap, C.byref(m)),
              "amp_rle_from_polygon")
        out.append({"size": [int(h), int(w)], "counts": counts_to_string(buf[: m.value])})
    return out[0] if single else out
