"""GPU tests of the device-side COCO counts-string encoder (csrc/rle_string.hip): byte-identical to the reference's
6 012 pycocotools strings (tests/golden/rle_pickles.json.gz) and to the host codec on the masks of a model run."""
import base64
import gzip
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "rle_pickles.json.gz")


def test_device_strings_equal_the_reference_bytes(gpu_ctx):
    from ampis_amd import ops, rle as RLE
    with gzip.open(GOLD, "rt") as f:
        gold = json.load(f)
    strings = [base64.b64decode(c) for fl in gold["files"] for im in fl["images"] for c in im["counts_b64"]]
    assert len(strings) == 6012
    counts = [RLE.string_to_counts(s) for s in strings]
    ln = np.array([len(c) for c in counts], np.int64)
    off = np.concatenate([[0], np.cumsum(ln)[:-1]])
    pool = torch.from_numpy(np.concatenate(counts).astype(np.int64).astype(np.uint32).view(np.int32)).cuda()
    got = ops.rle_strings_device(gpu_ctx, pool, torch.from_numpy(off), torch.from_numpy(ln))
    assert got == strings


def test_device_strings_edge_cases(gpu_ctx):
    """Empty run lists, single runs, runs at the 32-bit limit (7 characters), negative differences, more than 64 runs per mask and
    more masks than one scan pass (1024) holds."""
    from ampis_amd import ops, rle as RLE
    rng = np.random.default_rng(5)
    masks = [np.zeros(0, np.uint32), np.array([0], np.uint32), np.array([7], np.uint32), np.array([0xFFFFFFFF], np.uint32),
             np.array([0, 0xFFFFFFFF, 0, 1, 0xFFFFFFFF, 0, 5], np.uint32), np.array([5, 4, 3, 2, 1, 0, 0, 0, 1 << 31, 1, 1 << 31], np.uint32)]
    for n in (63, 64, 65, 129, 1000):
        masks.append(rng.integers(0, 1 << int(rng.integers(1, 32)), size=n).astype(np.uint32))
    masks += [rng.integers(0, 300, size=int(rng.integers(0, 40))).astype(np.uint32) for _ in range(2100)]
    ln = np.array([len(c) for c in masks], np.int64)
    off = np.concatenate([[0], np.cumsum(ln)[:-1]])
    pool = torch.from_numpy(np.concatenate(masks).view(np.int32)).cuda()
    got = ops.rle_strings_device(gpu_ctx, pool, torch.from_numpy(off), torch.from_numpy(ln))
    want = [RLE.counts_to_strings(c, np.array([0]), np.array([len(c)]))[0] if len(c) else b"" for c in masks]
    assert got == want


def test_model_strings_equal_host_encoding_of_its_run_lengths():
    """amp_model_set_rle_output(AMP_RLE_BOTH): every detection's device-encoded string equals the host encoding of the run lengths the
    same call returned; AMP_RLE_STRINGS returns the same strings without the run lengths; MaskRCNN.infer(rle='bytes') hands them out."""
    from ampis_amd import _lib, params as P, synth, rle as RLE
    from ampis_amd.model import MaskRCNN, RLE_BOTH, RLE_STRINGS
    import ctypes as C
    ctx = _lib.Context(0)
    B, S, D = 2, 256, 50
    m = MaskRCNN(ctx, 2, max_batch=B, max_h=S, max_w=S, max_out_hw=S, detections_per_image=D)
    m.load_params(P.init_params(2, seed=0, style="spread"))
    imgs, _ = synth.batch(B, S, S, first_index=0)
    m.set_rle_output(RLE_BOTH)
    d = m.infer_raw(imgs)
    n = [d.n[b] for b in range(B)]
    assert sum(n) > 10
    both = []
    for b in range(B):
        for i in range(n[b]):
            o, l = d.rle_off[b * D + i], d.rle_len[b * D + i]
            cnt = np.ctypeslib.as_array(d.rle_counts, (o + l,))[o:o + l].copy()
            so, sl = d.rle_str_off[b * D + i], d.rle_str_len[b * D + i]
            s = C.string_at(d.rle_str + so, sl)
            assert s == RLE.counts_to_strings(cnt, np.array([0]), np.array([l]))[0]
            assert int(cnt.sum()) == S * S
            both.append(s)
    m.set_rle_output(RLE_STRINGS)
    d = m.infer_raw(imgs)
    assert not d.rle_counts
    only = [C.string_at(d.rle_str + d.rle_str_off[b * D + i], d.rle_str_len[b * D + i]) for b in range(B) for i in range(d.n[b])]
    assert only == both
    out = m.infer(imgs)
    assert [mm["counts"] for o in out for mm in o["masks"]] == both
    out_c = m.infer(imgs, rle="counts")
    assert [RLE.counts_to_strings(mm["counts"], np.array([0]), np.array([len(mm["counts"])]))[0] for o in out_c for mm in o["masks"]] == both
    m.close()
    ctx.close()
