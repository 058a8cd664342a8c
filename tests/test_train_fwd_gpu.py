"""Training-mode forward (amp_model_forward_losses, through the C ABI) against oracle/train.py on identical inputs, weights
and sampling seed: anchor labels, sampled anchor / RoI sets, polygon mask targets, and the five losses."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup(gpu_ctx):
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    K, B, H, W = 2, 2, 256, 320
    imgs, gts = synth.batch(B, H, W, seed=5)
    gts = [dict(boxes=g["boxes"][:60], classes=g["classes"][:60], polygons=g["polygons"][:60]) for g in gts]
    npp = P.init_params(K, seed=1, style="spread")
    cfg = T.TrainCfg(num_classes=K, seed=7)
    st = {}
    ref = T.forward_losses(imgs, gts, M.to_torch_params(npp), cfg, stages=st)
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=4096, max_poly_doubles=4096 * 64)
    model.load_params(npp)
    got = model.forward_losses(imgs, gts, seed=7)
    return dict(model=model, got=got, ref={k: float(v) for k, v in ref.items()}, st=st, gts=gts, B=B, K=K)


def test_anchor_labels_and_rpn_samples(setup):
    from oracle import train as T
    m, st, B = setup["model"], setup["st"], setup["B"]
    label, sampled, counts = m.tap("rpn_label"), m.tap("rpn_sampled"), m.tap("rpn_counts")
    anchors = st["anchors"]
    for b in range(B):
        gtb = torch.as_tensor(setup["gts"][b]["boxes"], dtype=torch.float32)
        _, ml = T.matcher(T.pairwise_iou(gtb, anchors), (0.3, 0.7), (0, -1, 1), True)
        assert np.array_equal(label[b], ml.numpy())                       # exact labels for all 60k+ anchors
        pos, neg, _ = st["rpn_samples"][b]
        assert counts[b, 0] == len(pos) and counts[b, 1] == len(neg)
        assert np.array_equal(sampled[b, :len(pos)], pos.numpy())          # same seeded subset, same order
        assert np.array_equal(sampled[b, len(pos):len(pos) + len(neg)], neg.numpy())


def test_roi_samples_and_mask_targets(setup):
    m, st, B, K = setup["model"], setup["st"], setup["B"], setup["K"]
    rois, cls, gti, counts = m.tap("train_rois"), m.tap("train_roi_cls"), m.tap("train_roi_gti"), m.tap("train_roi_counts")
    exact = 0
    for b in range(B):
        rc = st["roi_cls"][b].numpy()
        n = len(rc)
        assert counts[b, 0] + counts[b, 1] == n
        assert counts[b, 0] == int((rc != K).sum())
        assert np.array_equal(cls[b, :n], rc)
        d = np.abs(rois[b, :n] - st["rois"][b].numpy()).max()
        assert d < 5e-3                                                    # proposals differ by fp32 noise only
        exact += int(np.array_equal(gti[b, :n][rc != K], st["roi_gtidx"][b].numpy()[rc != K]))
    assert exact == B
    tg = m.tap("train_mask_targets")
    ref = st["mask_targets"].numpy().astype(np.uint8)
    assert tg.shape == ref.shape
    # targets are rasterised from the sampled RoI boxes: a box differing by 1e-4 px may move a polygon vertex across a
    # rounding boundary of rleFrPoly's 5x grid; allow a handful of pixels over the whole batch
    assert int((tg != ref).sum()) <= 0.002 * ref.size


def test_losses_match_oracle(setup):
    got, ref = setup["got"], setup["ref"]
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask"):
        assert got[k] == pytest.approx(ref[k], rel=2e-4, abs=1e-6), (k, got[k], ref[k])


def test_polygon_rasteriser_exact_on_identical_boxes(gpu_ctx):
    """amp_mask_target_loss on boxes given bit-identically to the oracle's rasterize_polygon_within_box: exact targets."""
    import ctypes as C
    from ampis_amd._lib import check, lib, ptr
    from oracle import train as T
    rng = np.random.default_rng(3)
    N, K = 64, 1
    th = np.linspace(0, 2 * np.pi, 33)[:-1]
    polys, boxes = [], []
    for i in range(N):
        cx, cy, r = rng.uniform(30, 200), rng.uniform(30, 200), rng.uniform(3, 60)
        p = np.stack([cx + r * np.cos(th) * rng.uniform(0.6, 1.0), cy + r * np.sin(th)], 1).reshape(-1)
        if i % 7 == 0:
            p = np.array([cx - r, cy - r, cx + r, cy - r, cx + r, cy + r, cx - r, cy + r], float)   # axis-aligned square
        polys.append(p)
        j = rng.uniform(-0.3, 0.3, 4) * r
        boxes.append([cx - r + j[0], cy - r + j[1], cx + r + j[2], cy + r + j[3]])
    boxes = np.asarray(boxes, np.float32)
    poff = np.zeros(N + 1, np.int32); poff[1:] = np.cumsum([len(p) for p in polys])
    d = "cuda:0"
    t_boxes = torch.from_numpy(boxes).to(d)
    t_cls = torch.zeros(N, dtype=torch.int32, device=d)
    t_pid = torch.arange(N, dtype=torch.int32, device=d)
    t_pxy = torch.from_numpy(np.concatenate(polys)).to(d)
    t_poff = torch.from_numpy(poff).to(d)
    logits = torch.randn(N, 28, 28, K, device=d)
    part = torch.zeros(N, device=d)
    tgt = torch.zeros(N, 784, dtype=torch.uint8, device=d)
    check(lib().amp_mask_target_loss(gpu_ctx.handle, N, K, ptr(logits), None, ptr(t_boxes), ptr(t_cls), ptr(t_pid), ptr(t_pxy), ptr(t_poff),
                                     ptr(part), ptr(tgt)))
    torch.cuda.synchronize()
    got = tgt.cpu().numpy().reshape(N, 28, 28)
    for i in range(N):
        ref = T.rasterize_polygon_within_box(polys[i], boxes[i], 28)
        assert np.array_equal(got[i].astype(bool), ref), i
    # per-RoI BCE sums
    lg = logits.cpu()[..., 0]
    ref_l = torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.from_numpy(got.astype(np.float32)), reduction="none").flatten(1).sum(1)
    assert torch.allclose(part.cpu(), ref_l, rtol=1e-5, atol=1e-4)


def test_ragged_batch_losses_match_oracle(gpu_ctx):
    """Two images of different size in one batch (what the trainer's ResizeShortestEdge batches look like): the smaller one sits
    top-left in the common frame, its padding is zero AFTER normalisation whatever bytes the frame holds there, and its proposals are
    clipped to its own size.  One of the images has a single GT instance."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    K, B, H, W = 2, 2, 256, 320
    sizes = [(256, 320), (200, 250)]
    imgs, gts = synth.batch(B, H, W, seed=9)
    imgs = imgs.copy()
    imgs[1, 200:, :, :] = 255          # garbage in the padding of the smaller image: must not matter
    imgs[1, :, 250:, :] = 17
    def inside(g, h, w, n):
        keep = [i for i in range(len(g["boxes"])) if g["boxes"][i][2] <= w - 1 and g["boxes"][i][3] <= h - 1][:n]
        return dict(boxes=np.asarray(g["boxes"])[keep], classes=np.asarray(g["classes"])[keep], polygons=[g["polygons"][i] for i in keep])
    gts = [inside(gts[0], 256, 320, 1), inside(gts[1], 200, 250, 40)]
    assert len(gts[0]["boxes"]) == 1 and len(gts[1]["boxes"]) >= 5
    npp = P.init_params(K, seed=1, style="spread")
    cfg = T.TrainCfg(num_classes=K, seed=3)
    ref = T.forward_losses(imgs, gts, M.to_torch_params(npp), cfg, image_sizes=sizes)
    ref_full = T.forward_losses(imgs, gts, M.to_torch_params(npp), cfg)
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=4096, max_poly_doubles=4096 * 64)
    model.load_params(npp)
    model.set_image_sizes(sizes)
    got = model.forward_losses(imgs, gts, seed=3)
    model.set_image_sizes(None)
    model.close()
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask"):
        assert got[k] == pytest.approx(float(ref[k]), rel=2e-4, abs=1e-6), (k, got[k], float(ref[k]))
    # the sizes matter: treating the frame as one full image gives different losses (the test would pass vacuously otherwise)
    assert any(abs(float(ref[k]) - float(ref_full[k])) > 1e-3 * abs(float(ref[k])) for k in ref)


@pytest.mark.parametrize("K", [1, 5])
def test_losses_other_class_counts(gpu_ctx, K):
    """The tutorial trains NUM_CLASSES = 1 (particles); 5 exercises an odd predictor width (6 + 20 outputs, 5 mask channels).  One
    image of the batch has no annotation at all (detectron2 would filter it out of a dataset, but the path must not depend on that)."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    B, H, W = 2, 224, 288
    imgs, gts = synth.batch(B, H, W, seed=12)
    gts = [dict(boxes=np.asarray(g["boxes"])[:30], classes=np.asarray(g["classes"])[:30] % K, polygons=g["polygons"][:30]) for g in gts]
    gts[1] = dict(boxes=np.zeros((0, 4), np.float32), classes=np.zeros(0, np.int64), polygons=[])
    npp = P.init_params(K, seed=2, style="spread")
    cfg = T.TrainCfg(num_classes=K, seed=11)
    ref = T.forward_losses(imgs, gts, M.to_torch_params(npp), cfg)
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=4096, max_poly_doubles=4096 * 64)
    model.load_params(npp)
    got = model.forward_losses(imgs, gts, seed=11, backward=True)
    model.close()
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask"):
        assert got[k] == pytest.approx(float(ref[k]), rel=2e-4, abs=1e-6), (k, got[k], float(ref[k]))


@pytest.mark.parametrize("case", ["untouched_gt", "two_lds_tiles", "no_gt_image"])
def test_anchor_labels_edge_cases(gpu_ctx, case):
    """amp_anchor_labels through the C ABI against the oracle's pairwise_iou + Matcher on the cases the sweep's culling must not change
    (the kernel evaluates only GT boxes that overlap a wave's bounding box): a GT box no anchor touches and a zero-area one -- their best
    IoU is 0, which "equals" the IoU 0 of EVERY anchor, so set_low_quality_matches_ marks them all (detectron2 matcher.py semantics) --,
    more than 256 GT boxes (two LDS tiles), and an image without GT between two that have some."""
    import ctypes as C
    from ampis_amd import _lib, ops
    from oracle import maskrcnn as M, train as T
    rng = np.random.default_rng(3)
    shapes = [(64, 80), (32, 40), (16, 20), (8, 10), (4, 5)]
    H, W = 256, 320

    def boxes(n):
        c = rng.uniform([0, 0], [W, H], size=(n, 2)); s = rng.uniform(6, 120, size=(n, 2))
        return np.concatenate([np.clip(c - s / 2, 0, None), np.minimum(c + s / 2, [W, H])], axis=1).astype(np.float32)

    if case == "untouched_gt":
        per_image = [np.concatenate([boxes(20), np.array([[5000, 5000, 5100, 5100], [40, 40, 40, 90]], np.float32)]), boxes(7)]
    elif case == "two_lds_tiles":
        per_image = [boxes(300), boxes(257)]
    else:
        per_image = [boxes(12), np.zeros((0, 4), np.float32), boxes(5)]
    B = len(per_image)
    anchors = torch.cat([M.grid_anchors(h, w, M.STRIDES[l], M.ANCHOR_SIZES[l]) for l, (h, w) in enumerate(shapes)])
    A = anchors.shape[0]
    gt_all = np.concatenate(per_image).astype(np.float32)
    off = np.concatenate([[0], np.cumsum([len(p) for p in per_image])]).astype(np.int32)
    dev = "cuda:0"
    d_gt = torch.from_numpy(gt_all if len(gt_all) else np.zeros((1, 4), np.float32)).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    dummy = [torch.zeros((B, h * w, 16), device=dev) for h, w in shapes]
    lv = ops.make_rpn_levels(dummy, shapes)
    mv = torch.empty((B, A), device=dev); mi = torch.empty((B, A), dtype=torch.int32, device=dev)
    best = torch.zeros((max(len(gt_all), 1),), dtype=torch.int32, device=dev)
    lab = torch.empty((B, A), dtype=torch.int8, device=dev)
    ops.check(_lib.lib().amp_anchor_labels(gpu_ctx.handle, C.byref(lv), B, ops.ptr(d_gt), ops.ptr(d_off), int(len(gt_all)), 0.3, 0.7,
                                           ops.ptr(mv), ops.ptr(mi), ops.ptr(best), ops.ptr(lab)), "amp_anchor_labels")
    torch.cuda.synchronize()
    for b in range(B):
        gtb = torch.from_numpy(per_image[b])
        if len(gtb) == 0:
            assert int(lab[b].abs().sum()) == 0 and float(mv[b].abs().sum()) == 0.0
            continue
        mq = T.pairwise_iou(gtb, anchors)
        matches, ml = T.matcher(mq, (0.3, 0.7), (0, -1, 1), True)
        assert np.array_equal(lab[b].cpu().numpy(), ml.numpy()), case
        assert np.array_equal(mv[b].cpu().numpy(), mq.max(dim=0)[0].numpy())
        assert np.array_equal(mi[b].cpu().numpy(), matches.numpy().astype(np.int32))
        assert np.array_equal(best[off[b]:off[b + 1]].cpu().numpy().view(np.float32), mq.max(dim=1)[0].numpy())
    if case == "untouched_gt":
        assert int((lab[0] == 1).sum()) == A and int((lab[1] == 1).sum()) < A
