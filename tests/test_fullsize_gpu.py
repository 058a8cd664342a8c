"""BASELINE-size invariants (batch of 1024x1024 synthetic micrographs, 200 detections / image): properties that hold at any size
and need no oracle run -- sortedness, RLE well-formedness, masks inside their boxes, the NMS invariant, bitwise determinism."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FULLSIZE_TIE_SHARE_CAP, FULLSIZE_TIE_PIXELS_CAP, FULLSIZE_BOX_REL_CAP, FULLSIZE_IOU_MIN = 0.40, 20, 6, 0.99      # a little above the measured level (see the test)


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
    stats = []
    for b in (0, 5):
        ref = O.infer(imgs[b:b + 1], tp, O.Cfg(num_classes=K, detections_per_image=D))[0]
        assert len(ref["boxes"]) == D
        # fp32 MFMA sums two products per instruction, the split arithmetic 16: its re-association noise is the LARGER one (DESIGN §4.1,
        # tests/test_conv_modes_gpu.py), measured 3.3 ppm of a 393-px side against 1.9 ppm -> its relative term is 4e-6, not 3e-6
        stats.append(gate.check_image(out[b], ref, S, S, lambda mk: _decode(mk["counts"], S, S), box_rel=4e-6 if mode == "f32" else gate.BOX_REL))
    st = gate.merge(stats)
    print(f"full-size gate [{mode}]:", gate.summary(st))
    assert st["instances"] == 2 * D and st["identical"] + st["tie_masks"] == st["instances"]
    # caps a little above the measured level (round 3: f16x3 115 of 400 masks with ties, at most 8 pixels in one, 6 masks below IoU 0.999
    # -- the largest 1158 px, lowest IoU 0.9969 -- and 2 boxes of ~740 px inside the relative term; f32 124 / 12 / 6 / 0.9974 / 5):
    # a regression shows here even while the per-instance rule holds
    gate.assert_bounds(st, tie_mask_share=FULLSIZE_TIE_SHARE_CAP, max_tie_pixels=FULLSIZE_TIE_PIXELS_CAP, iou_min=FULLSIZE_IOU_MIN, box_rel_used=FULLSIZE_BOX_REL_CAP)
    if mode == "f32":
        return
    # where 1e-3 px is below the reference's own fp32 noise (boxes of several hundred px): against an exact-convolution evaluation
    # of the same network (oracle/exact.py) the HIP path must be as close as the fp32 oracle is
    from oracle import exact
    with exact.exact_convs():
        rex = O.infer(imgs[5:6], tp, O.Cfg(num_classes=K, detections_per_image=D))[0]["boxes"].numpy()
    r32 = O.infer(imgs[5:6], tp, O.Cfg(num_classes=K, detections_per_image=D))[0]["boxes"].numpy()
    dist = lambda a: max(float(np.abs(a - rex[i]).max(axis=1).min()) for i in range(len(rex)))
    d_hip, d_ref = dist(out[5]["boxes"]), dist(r32)
    print(f"worst box distance from the exact-convolution oracle: HIP {d_hip:.2e} px, fp32 oracle {d_ref:.2e} px")
    assert d_hip <= max(1e-3, 1.5 * d_ref), (d_hip, d_ref)


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
