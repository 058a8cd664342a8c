"""ORACLE (test infrastructure, NOT product code) — numpy/pure-Python restatement of the COCO RLE codec.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Follows pycocotools 2.0.4 common/maskApi.c (docker/env.yml:21; NOT vendored in /root/reference) as called from
ampis/data_utils.py:275 (`RLE.encode(np.asfortranarray(x))`), ampis/analyze.py:108,158 (`rle.iou`), :315-321
(`rle.merge`, `rle.area`) and ampis/structures.py:752 (`RLE.decode`).  Pinned by the reference's own data: every one of
the 6 012 masks in its five result pickles must decode to h*w pixels and re-encode to the identical `counts` bytes
(tests/test_rle_golden.py, fixture tests/golden/rle_pickles.json.gz).
"""
import numpy as np


def encode_counts(mask):
    """mask [H,W] bool -> list of run lengths, column-major, first run counts zeros (maskApi.c rleEncode)."""
    flat = np.asarray(mask, dtype=bool).reshape(-1, order="F")
    if flat.size == 0:
        return np.zeros(1, dtype=np.uint32)
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    runs = np.diff(bounds)
    if flat[0]:
        runs = np.concatenate(([0], runs))
    return runs.astype(np.uint32)


def decode_counts(counts, h, w):
    """run lengths -> [H,W] bool (maskApi.c rleDecode)."""
    counts = np.asarray(counts, dtype=np.int64)
    assert counts.sum() == h * w, (int(counts.sum()), h * w)
    vals = (np.arange(len(counts)) % 2).astype(bool)
    return np.repeat(vals, counts).reshape((h, w), order="F")


def counts_to_string(counts):
    """maskApi.c rleToString: 5-bit groups, LSB first, +48, 0x20 = continuation, deltas against run i-2 for i > 2."""
    out = bytearray()
    c = [int(v) for v in counts]
    for i, x in enumerate(c):
        if i > 2:
            x -= c[i - 2]
        more = True
        while more:
            ch = x & 0x1F
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def string_to_counts(s):
    """maskApi.c rleFrString."""
    if isinstance(s, str):
        s = s.encode("ascii")
    counts = []
    p = 0
    while p < len(s):
        x = 0
        k = 0
        more = True
        while more:
            ch = s[p] - 48
            x |= (ch & 0x1F) << (5 * k)
            more = bool(ch & 0x20)
            p += 1
            k += 1
            if not more and (ch & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return np.asarray(counts, dtype=np.uint32)


def encode(mask):
    h, w = mask.shape
    return {"size": [h, w], "counts": counts_to_string(encode_counts(mask))}


def decode(rle):
    h, w = rle["size"]
    c = rle["counts"]
    c = string_to_counts(c) if isinstance(c, (bytes, str)) else c
    return decode_counts(c, h, w)


def area(rle):
    c = rle["counts"]
    c = string_to_counts(c) if isinstance(c, (bytes, str)) else np.asarray(c)
    return int(c[1::2].sum())


def iou(dt, gt, iscrowd):
    """pycocotools.mask.iou on RLE lists: [len(dt), len(gt)] float64; union -> area(dt) for crowd gt; 0 if disjoint."""
    out = np.zeros((len(dt), len(gt)))
    dm = [decode(d) for d in dt]
    for j, g in enumerate(gt):
        gm = decode(g)
        for i, d in enumerate(dm):
            inter = int((d & gm).sum())
            if inter == 0:
                out[i, j] = 0.0
            else:
                u = int(d.sum()) if (len(iscrowd) and iscrowd[j]) else int((d | gm).sum())
                out[i, j] = inter / u
    return out


def merge(rles, intersect=False):
    m = decode(rles[0])
    for r in rles[1:]:
        m = (m & decode(r)) if intersect else (m | decode(r))
    return encode(m)
