"""ORACLE (test infrastructure, NOT product code) — training-mode forward of Mask R-CNN R50-FPN and its five losses.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Restates what `model(data)` returns in training mode, the dict AMPIS's LossEvalHook iterates
(ampis/data_utils.py:111-122: `metrics_dict = self._model(data)`; keys loss_cls, loss_box_reg, loss_mask, loss_rpn_cls,
loss_rpn_loc) and DefaultTrainer.run_step differentiates (notebook cell 22).  The arithmetic lives in detectron2 (not
vendored): proposal_generator/rpn.py (label_and_sample_anchors, losses), matcher.py, sampling.py,
roi_heads/roi_heads.py (label_and_sample_proposals), fast_rcnn.py (losses), mask_head.py (mask_rcnn_loss),
structures/masks.py (rasterize_polygons_within_box) and pycocotools maskApi.c rleFrPoly.  PARITY UNPINNED (no reference
test pins these numerics); everything is written in differentiable torch so autograd gives the reference gradients.

Random sub-sampling: detectron2 uses torch.randperm on the device, which cannot be reproduced; SURVEY §7.2 defines training
parity on fixed index sets.  Both this oracle and the HIP path therefore draw the SAME subset from a counter-based hash:
among the candidates, take the k smallest values of hash32(seed, image, stream, index) (ties by index).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import maskrcnn as M


# ------------------------------------------------------------------------------------------------ sampling RNG
def hash32(seed, image, stream, idx):
    """murmur3-style finaliser over (seed, image, stream, idx) -> uint32 (numpy, vectorised over idx)."""
    idx = np.asarray(idx, dtype=np.uint64)
    x = (np.uint64(seed) * np.uint64(0x9E3779B1) + np.uint64(image) * np.uint64(0x85EBCA77) +
         np.uint64(stream) * np.uint64(0xC2B2AE3D) + idx * np.uint64(0x27D4EB2F)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def sample_k(cand_idx, k, seed, image, stream, ident=None):
    """the k candidates with the largest key = max(0xffffffff - hash32(seed, image, stream, identity), 1) (ties by index),
    in descending-key order.  `ident[i]` is the hashed identity of candidate index i (default: the index itself)."""
    cand_idx = np.asarray(cand_idx, dtype=np.int64)
    if k <= 0 or len(cand_idx) == 0:
        return np.zeros(0, dtype=np.int64)
    idv = cand_idx if ident is None else np.asarray(ident, dtype=np.int64)[cand_idx]
    key = np.maximum(np.uint64(0xFFFFFFFF) - hash32(seed, image, stream, idv).astype(np.uint64), np.uint64(1))
    order = np.lexsort((cand_idx, np.uint64(0xFFFFFFFF) - key))
    return cand_idx[order[:k]]


# ------------------------------------------------------------------------------------------------ matching
def pairwise_iou(b1, b2):
    """detectron2 structures/boxes.py pairwise_iou (fp32): [N,M]; 0 where the intersection is empty."""
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


def matcher(mq, thresholds, labels, allow_low_quality):
    """detectron2 modeling/matcher.py Matcher.__call__. mq [G, N]. Returns (matches [N], match_labels [N] int8)."""
    n = mq.shape[1]
    if mq.numel() == 0:
        return torch.zeros(n, dtype=torch.int64), torch.full((n,), labels[0], dtype=torch.int8)
    vals, matches = mq.max(dim=0)
    th = [-float("inf")] + list(thresholds) + [float("inf")]
    ml = torch.full((n,), 1, dtype=torch.int8)
    for l, lo, hi in zip(labels, th[:-1], th[1:]):
        ml[(vals >= lo) & (vals < hi)] = l
    if allow_low_quality:
        best_per_gt, _ = mq.max(dim=1)
        _, pred_inds = torch.nonzero(mq == best_per_gt[:, None], as_tuple=True)
        ml[pred_inds] = 1
    return matches, ml


def get_deltas(src, tgt, weights):
    """Box2BoxTransform.get_deltas."""
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
    tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
    wx, wy, ww, wh = weights
    return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)


# ------------------------------------------------------------------------------------------------ mask targets
def fr_poly(xy, h, w):
    """pycocotools maskApi.c rleFrPoly: polygon (flat x0,y0,x1,y1,...) -> run lengths (column-major) of an h x w mask."""
    k = len(xy) // 2
    scale = 5.0
    x = [int(scale * xy[2 * j] + 0.5) for j in range(k)]
    y = [int(scale * xy[2 * j + 1] + 0.5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        if dx >= dy:
            s = (ye - ys) / dx if dx else 0.0
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + s * t + 0.5))
        else:
            s = (xe - xs) / dy
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + s * t + 0.5))
    xs_, ys_ = [], []
    for j in range(1, len(u)):
        if u[j] != u[j - 1]:
            xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
            xd = (xd + 0.5) / scale - 0.5
            if math.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
            yd = (yd + 0.5) / scale - 0.5
            yd = 0.0 if yd < 0 else (float(h) if yd > h else yd)
            yd = math.ceil(yd)
            xs_.append(int(xd))
            ys_.append(int(yd))
    a = sorted([xv * h + yv for xv, yv in zip(xs_, ys_)] + [h * w])
    p = 0
    for j in range(len(a)):
        t = a[j]
        a[j] -= p
        p = t
    b = [a[0]]
    j = 1
    while j < len(a):
        if a[j] > 0:
            b.append(a[j])
            j += 1
        else:
            j += 1
            if j < len(a):
                b[-1] += a[j]
                j += 1
    return np.asarray(b, dtype=np.uint32)


def rasterize_polygon_within_box(poly, box, size):
    """detectron2 structures/masks.py rasterize_polygons_within_box: the polygons of ONE instance (a flat xy array, or a list of
    them) are moved into the box frame, scaled to size x size and turned into a bitmask by polygons_to_bitmask = frPyObjects of each
    polygon, merge (union), decode.  -> bool [size,size]."""
    from . import rle as R
    polys = poly if isinstance(poly, (list, tuple)) and len(poly) and not np.isscalar(poly[0]) else [poly]
    w, h = float(box[2] - box[0]), float(box[3] - box[1])
    rh, rw = size / max(h, 0.1), size / max(w, 0.1)
    out = np.zeros((size, size), dtype=bool)
    for q in polys:
        p = np.asarray(q, dtype=np.float64).copy()
        p[0::2] -= float(box[0])
        p[1::2] -= float(box[1])
        if rh == rw:
            p *= rh
        else:
            p[0::2] *= rw
            p[1::2] *= rh
        out |= R.decode_counts(fr_poly(p, size, size), size, size).astype(bool)
    return out


def bitmask_crop_and_resize(mask, box, size):
    """detectron2 structures/masks.py BitMasks.crop_and_resize for ONE instance: ROIAlign((size, size), 1.0, 0, aligned=True) of the
    0/1 mask as a one-channel fp32 map, then >= 0.5.  mask: bool [H,W] at network-input resolution.  -> bool [size,size]."""
    m = torch.from_numpy(np.ascontiguousarray(mask, dtype=np.float32))[None]
    out = M.roi_align(m, torch.as_tensor(np.asarray(box, dtype=np.float32).reshape(1, 4)), size, 1.0)
    return (out[0, 0].numpy() >= 0.5)


def instance_mask_target(g, gi, box, size=28):
    """Mask target of instance gi of image-gt dict g for proposal `box`: from its bitmask when g['masks_rle'][gi] is given
    (INPUT.MASK_FORMAT = 'bitmask'), else from its polygon(s)."""
    from . import rle as R
    mr = g.get("masks_rle")
    if mr is not None and mr[gi] is not None:
        return bitmask_crop_and_resize(R.decode(mr[gi]).astype(bool), box, size)
    return rasterize_polygon_within_box(g["polygons"][gi], box, size)


# ------------------------------------------------------------------------------------------------ differentiable RoIAlign
def roi_align_torch(feat, rois, out_size, spatial_scale):
    """Same arithmetic as M.roi_align (torchvision roi_align, aligned=True, sampling_ratio=0) in torch ops, so autograd gives
    the reference gradient w.r.t. `feat` [C,H,W] (what roi_align_backward_kernel computes). rois [R,4] are constants."""
    C, H, W = feat.shape
    outs = []
    f32 = torch.float32
    for r in rois.detach():
        sw, sh = r[0] * spatial_scale - 0.5, r[1] * spatial_scale - 0.5
        rw, rh = (r[2] * spatial_scale - 0.5) - sw, (r[3] * spatial_scale - 0.5) - sh
        bh, bw = rh / out_size, rw / out_size
        gh, gw = int(torch.ceil(bh)), int(torch.ceil(bw))
        if gh <= 0 or gw <= 0:
            outs.append(feat.new_zeros((C, out_size, out_size)))
            continue
        ph = torch.arange(out_size, dtype=f32)

        def prep(start, b, g, size):
            v = start + ph[:, None] * b + (torch.arange(g, dtype=f32)[None, :] + 0.5) * b / g      # [P, g]
            bad = (v < -1.0) | (v > size)
            v = torch.where(v <= 0, torch.zeros_like(v), v)
            lo = v.to(torch.int64)
            clip = lo >= size - 1
            lo = torch.where(clip, torch.full_like(lo, size - 1), lo)
            hi = torch.where(clip, torch.full_like(lo, size - 1), lo + 1)
            v = torch.where(clip, lo.to(f32), v)
            l = v - lo.to(f32)
            return bad, lo, hi, l, 1.0 - l

        ybad, ylo, yhi, ly, hy = prep(sh, bh, gh, H)
        xbad, xlo, xhi, lx, hx = prep(sw, bw, gw, W)
        acc = feat.new_zeros((C, out_size, out_size))
        for iy in range(gh):
            for ix in range(gw):
                w1 = (hy[:, iy, None] * hx[None, :, ix])
                w2 = (hy[:, iy, None] * lx[None, :, ix])
                w3 = (ly[:, iy, None] * hx[None, :, ix])
                w4 = (ly[:, iy, None] * lx[None, :, ix])
                v1 = feat[:, ylo[:, iy][:, None], xlo[:, ix][None, :]]
                v2 = feat[:, ylo[:, iy][:, None], xhi[:, ix][None, :]]
                v3 = feat[:, yhi[:, iy][:, None], xlo[:, ix][None, :]]
                v4 = feat[:, yhi[:, iy][:, None], xhi[:, ix][None, :]]
                val = ((w1 * v1 + w2 * v2) + w3 * v3) + w4 * v4
                ok = ~(ybad[:, iy][:, None] | xbad[:, ix][None, :])
                acc = acc + val * ok
        outs.append(acc / float(max(gh * gw, 1)))
    return torch.stack(outs) if outs else feat.new_zeros((0, C, out_size, out_size))


def roi_pool_torch(features, boxes_per_img, out_size):
    """M.roi_pool with the differentiable RoIAlign."""
    boxes = torch.cat(boxes_per_img) if boxes_per_img else torch.zeros(0, 4)
    bidx = torch.cat([torch.full((len(b),), i, dtype=torch.int64) for i, b in enumerate(boxes_per_img)])
    lv = M.assign_levels(boxes)
    C = features[0].shape[1]
    out = [None] * len(boxes)
    for l in range(4):
        for b in range(len(boxes_per_img)):
            sel = torch.nonzero((lv == l) & (bidx == b)).squeeze(1)
            if len(sel):
                o = roi_align_torch(features[l][b], boxes[sel], out_size, 1.0 / M.STRIDES[l])
                for j, i in enumerate(sel.tolist()):
                    out[i] = o[j]
    return torch.stack(out) if out else features[0].new_zeros((0, C, out_size, out_size))


# ------------------------------------------------------------------------------------------------ the forward
class TrainCfg(M.Cfg):
    def __init__(self, **kw):
        M.Cfg.__init__(self)
        self.pre_nms_topk = 2000          # MODEL.RPN.PRE_NMS_TOPK_TRAIN
        self.post_nms_topk = 1000         # MODEL.RPN.POST_NMS_TOPK_TRAIN
        self.rpn_batch = 256
        self.rpn_pos_frac = 0.5
        self.rpn_iou = (0.3, 0.7)
        self.roi_batch = 512
        self.roi_pos_frac = 0.25
        self.roi_iou = 0.5
        self.seed = 0
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def forward_losses(images_u8, gt, params, cfg, stages=None, image_sizes=None):
    """GeneralizedRCNN.forward in training mode.  gt: per image dict(boxes f32 [G,4], classes i64 [G], polygons list[G] of
    flat xy float64 arrays).  Returns dict of the 5 loss tensors (differentiable w.r.t. `params`).
    image_sizes: per image (h, w) of the valid top-left part of the common frame (a batch of differently sized images as
    ImageList.from_tensors stacks it: the NORMALISED image is padded with 0, and find_top_rpn_proposals clips each image's proposals
    to its own size); None = every image fills the frame."""
    B, H, W, _ = images_u8.shape
    K = cfg.num_classes
    x = M.preprocess(images_u8, cfg)
    if image_sizes is not None:
        x = x.clone()
        for b, (h_b, w_b) in enumerate(image_sizes):
            x[b, :, h_b:, :] = 0.0
            x[b, :, :, w_b:] = 0.0
    res = M.resnet50(x, params, cfg)
    feats = M.fpn(res, params)
    rpn_outs = M.rpn_head(feats, params)
    shapes = [(f.shape[2], f.shape[3]) for f in feats]
    anchors = torch.cat([M.grid_anchors(h, w, M.STRIDES[l], M.ANCHOR_SIZES[l]) for l, (h, w) in enumerate(shapes)])
    logits = torch.cat([o[0] for o in rpn_outs], dim=1)           # [B, A]
    deltas = torch.cat([o[1] for o in rpn_outs], dim=1)           # [B, A, 4]

    # ---- RPN.label_and_sample_anchors + losses ----
    loss_cls = logits.new_zeros(())
    loss_loc = logits.new_zeros(())
    rpn_samples = []
    for b in range(B):
        gtb = torch.as_tensor(gt[b]["boxes"], dtype=torch.float32).reshape(-1, 4)
        mq = pairwise_iou(gtb, anchors)
        matches, ml = matcher(mq, cfg.rpn_iou, (0, -1, 1), True)
        pos_all = torch.nonzero(ml == 1).squeeze(1).numpy()
        neg_all = torch.nonzero(ml == 0).squeeze(1).numpy()
        npos = min(len(pos_all), int(cfg.rpn_batch * cfg.rpn_pos_frac))
        nneg = min(len(neg_all), cfg.rpn_batch - npos)
        pos = torch.from_numpy(sample_k(pos_all, npos, cfg.seed, b, 0))
        neg = torch.from_numpy(sample_k(neg_all, nneg, cfg.seed, b, 1))
        rpn_samples.append((pos, neg, matches))
        sel = torch.cat([pos, neg])
        tgt = torch.cat([torch.ones(len(pos)), torch.zeros(len(neg))])
        loss_cls = loss_cls + F.binary_cross_entropy_with_logits(logits[b][sel], tgt, reduction="sum")
        if len(pos):
            gd = get_deltas(anchors[pos], gtb[matches[pos]], (1.0, 1.0, 1.0, 1.0))
            loss_loc = loss_loc + (deltas[b][pos] - gd).abs().sum()        # smooth_l1 with beta = 0
    norm = float(cfg.rpn_batch * B)
    losses = {"loss_rpn_cls": loss_cls / norm, "loss_rpn_loc": loss_loc / norm}

    # ---- proposals (detached), GT appended, matched, sampled ----
    with torch.no_grad():
        cands = M.rpn_select_candidates([(l.detach(), d.detach()) for l, d in rpn_outs], shapes, cfg)
        props, prop_ids = [], []
        lvl_off = np.concatenate([[0], np.cumsum([h * w * 3 for h, w in shapes])])
        for c in cands:
            # same steps as M.rpn_proposals_from_candidates, also tracking each proposal's global anchor index
            boxes, lg, lvl, idx = c
            aid = torch.as_tensor(lvl_off[:-1])[lvl] + idx
            valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(lg)
            boxes, lg, lvl, aid = boxes[valid], lg[valid], lvl[valid], aid[valid]
            boxes = M.clip_boxes(boxes, *((H, W) if image_sizes is None else image_sizes[len(props)]))
            keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
            boxes, lg, lvl, aid = boxes[keep], lg[keep], lvl[keep], aid[keep]
            order = M.sort_desc_stable(lg)
            boxes, lvl, aid = boxes[order], lvl[order], aid[order]
            kept = M.nms_sorted(boxes, lvl, cfg.rpn_nms_thresh, max_keep=cfg.post_nms_topk)
            props.append(boxes[kept])
            prop_ids.append(aid[kept].numpy())
        num_anchors = int(lvl_off[-1])
    rois, roi_cls, roi_gt, roi_gtidx = [], [], [], []
    for b in range(B):
        gtb = torch.as_tensor(gt[b]["boxes"], dtype=torch.float32).reshape(-1, 4)
        gtc = torch.as_tensor(gt[b]["classes"], dtype=torch.int64)
        pb = torch.cat([props[b], gtb])                      # add_ground_truth_to_proposals: GT after the proposals
        mq = pairwise_iou(gtb, pb)
        matches, ml = matcher(mq, (cfg.roi_iou,), (0, 1), False)
        if len(gtb):
            cls = gtc[matches].clone()
            cls[ml == 0] = K
        else:
            cls = torch.full((len(pb),), K, dtype=torch.int64)
        fg_all = torch.nonzero(cls != K).squeeze(1).numpy()
        bg_all = torch.nonzero(cls == K).squeeze(1).numpy()
        nfg = min(len(fg_all), int(cfg.roi_batch * cfg.roi_pos_frac))
        nbg = min(len(bg_all), cfg.roi_batch - nfg)
        ident = np.concatenate([prop_ids[b], num_anchors + np.arange(len(gtb))])   # order-independent identity of each candidate
        sel = torch.from_numpy(np.concatenate([sample_k(fg_all, nfg, cfg.seed, b, 2, ident), sample_k(bg_all, nbg, cfg.seed, b, 3, ident)]))
        rois.append(pb[sel])
        roi_cls.append(cls[sel])
        roi_gtidx.append(matches[sel])
        roi_gt.append(gtb[matches[sel]] if len(gtb) else pb[sel])
    # ---- box head + losses (FastRCNNOutputLayers.losses) ----
    diff = any(getattr(v, 'requires_grad', False) for v in params.values())
    pooled = roi_pool_torch(feats[:4], rois, 7) if diff else M.roi_pool(feats[:4], rois, 7)[0]
    scores, bdeltas = M.box_head(pooled, params)
    gcls = torch.cat(roi_cls)
    losses["loss_cls"] = F.cross_entropy(scores, gcls, reduction="mean")
    fg = torch.nonzero((gcls >= 0) & (gcls < K)).squeeze(1)
    pboxes, gboxes = torch.cat(rois), torch.cat(roi_gt)
    fg_pred = bdeltas.view(-1, K, 4)[fg, gcls[fg]]
    gd = get_deltas(pboxes[fg], gboxes[fg], cfg.bbox_reg_weights)
    losses["loss_box_reg"] = (fg_pred - gd).abs().sum() / max(gcls.numel(), 1.0)
    # ---- mask head + loss (mask_rcnn_loss) on the foreground rois ----
    fg_rois = [r[c != K] for r, c in zip(rois, roi_cls)]
    mpooled = roi_pool_torch(feats[:4], fg_rois, 14) if diff else M.roi_pool(feats[:4], fg_rois, 14)[0]
    h = mpooled
    pre = "roi_heads.mask_head."
    for i in range(1, 5):
        h = F.relu(F.conv2d(h, params[f"{pre}mask_fcn{i}.weight"], params[f"{pre}mask_fcn{i}.bias"], padding=1))
    h = F.relu(F.conv_transpose2d(h, params[pre + "deconv.weight"], params[pre + "deconv.bias"], stride=2))
    mlogits = F.conv2d(h, params[pre + "predictor.weight"], params[pre + "predictor.bias"])
    targets = []
    for b in range(B):
        keep = (roi_cls[b] != K).numpy()
        for r, gi in zip(rois[b][keep], roi_gtidx[b][keep]):
            targets.append(torch.from_numpy(instance_mask_target(gt[b], int(gi), r.numpy(), 28)))
    fgc = torch.cat([c[c != K] for c in roi_cls])
    if len(targets):
        tg = torch.stack(targets).to(torch.float32)
        ml_sel = mlogits[torch.arange(len(fgc)), fgc]
        losses["loss_mask"] = F.binary_cross_entropy_with_logits(ml_sel, tg, reduction="mean")
    else:
        tg = torch.zeros(0, 28, 28)
        losses["loss_mask"] = mlogits.sum() * 0
    if stages is not None:
        stages.update(anchors=anchors, rpn_samples=rpn_samples, props=props, rois=rois, roi_cls=roi_cls, roi_gtidx=roi_gtidx,
                      box_scores=scores, box_deltas=bdeltas, mask_logits=mlogits, mask_targets=tg, feats=feats, rpn_outs=rpn_outs)
    return losses
