"""ORACLE (test infrastructure, NOT product code) — CPU restatement of Mask R-CNN R50-FPN inference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

What it restates: the computation `DefaultPredictor.__call__` performs for AMPIS (notebook cells 24-28,
GETTING_STARTED.md:40; output consumed at ampis/data_utils.py:275-278).  That arithmetic lives in third-party
code that is NOT vendored in /root/reference and NOT installed here: detectron2 (un-pinned, nominally v0.5/v0.6,
docker/Dockerfile:96-97), torchvision 0.10.0 ops.roi_align / ops.nms (docker/env.yml:11) and torch 1.9.0
(docker/env.yml:10).  It is restated from the published algorithm ([D2-KNOWLEDGE]); each
function names the detectron2/torchvision routine it follows.

PARITY UNPINNED: the reference ships no test, golden vector or weights that pin this network's numerics, and
detectron2 cannot be imported here (ordinary ModuleNotFoundError).  What *is* pinned by reference
assets is checked elsewhere: the COCO-RLE byte format (oracle/rle.py vs the five result pickles), the RLE-IoU
matcher known answers (oracle/matcher.py vs ampis/analyze.py:702-728) and the output container contract.

Conventions: tensors are torch CPU, activations NCHW fp32 (detectron2's layout); `params` maps detectron2
state_dict names to tensors.  Where torch/torchvision leave an order unspecified (ties in topk / sort), this
oracle fixes: descending score, ties by ascending index.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

RES_STAGES = (("res2", 3, 1), ("res3", 4, 2), ("res4", 6, 2), ("res5", 3, 2))
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ANCHOR_RATIOS = (0.5, 1.0, 2.0)
STRIDES = (4, 8, 16, 32, 64)
SCALE_CLAMP = math.log(1000.0 / 16)
BN_EPS = 1e-5


class Cfg:
    """The cfg values the inference path reads (detectron2 defaults)."""

    def __init__(self, **kw):
        self.num_classes = 80
        self.pixel_mean = (103.530, 116.280, 123.675)
        self.pixel_std = (1.0, 1.0, 1.0)
        self.size_divisibility = 32
        self.pre_nms_topk = 1000
        self.post_nms_topk = 1000
        self.rpn_nms_thresh = 0.7
        self.score_thresh = 0.05
        self.nms_thresh = 0.5
        self.detections_per_image = 100
        self.bbox_reg_weights = (10.0, 10.0, 5.0, 5.0)
        self.mask_threshold = 0.5
        # MODEL.RESNETS.*: blocks per stage, NUM_GROUPS (conv2), STRIDE_IN_1X1 -- R50 defaults; X101-32x8d = (3,4,23,3), 32, False
        self.resnet_blocks = (3, 4, 6, 3)
        self.num_groups = 1
        self.stride_in_1x1 = True
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


def to_torch_params(np_params, dtype=torch.float32):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in np_params.items()}


# --------------------------------------------------------------------------------------------- backbone
def frozen_bn(x, p, prefix):
    """detectron2 layers/batch_norm.py FrozenBatchNorm2d.forward, inference branch (x.requires_grad False):
    F.batch_norm(x, running_mean, running_var, weight, bias, training=False, eps=1e-5)."""
    return F.batch_norm(x, p[prefix + ".running_mean"], p[prefix + ".running_var"], p[prefix + ".weight"],
                        p[prefix + ".bias"], training=False, eps=BN_EPS)


def conv_bn(x, p, prefix, stride=1, padding=0, relu=False, groups=1):
    y = F.conv2d(x, p[prefix + ".weight"], None, stride=stride, padding=padding, groups=groups)
    y = frozen_bn(y, p, prefix + ".norm")
    return F.relu_(y) if relu else y


def resnet50(x, p, cfg=None):
    """detectron2 modeling/backbone/resnet.py: BasicStem + BottleneckBlock x (3,4,6,3), STRIDE_IN_1X1=True (R50); with cfg:
    blocks per stage, conv2 groups and the stride in conv1 or conv2 as BottleneckBlock.__init__ places them (ResNeXt)."""
    blocks = cfg.resnet_blocks if cfg is not None else (3, 4, 6, 3)
    groups = cfg.num_groups if cfg is not None else 1
    s1x1 = cfg.stride_in_1x1 if cfg is not None else True
    x = conv_bn(x, p, "backbone.bottom_up.stem.conv1", stride=2, padding=3, relu=True)
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    outs = {}
    for (name, _, stride), nblk in zip(RES_STAGES, blocks):
        for b in range(nblk):
            pre = f"backbone.bottom_up.{name}.{b}"
            s = stride if b == 0 else 1
            shortcut = conv_bn(x, p, pre + ".shortcut", stride=s) if (pre + ".shortcut.weight") in p else x
            y = conv_bn(x, p, pre + ".conv1", stride=s if s1x1 else 1, relu=True)
            y = conv_bn(y, p, pre + ".conv2", stride=1 if s1x1 else s, padding=1, relu=True, groups=groups)
            y = conv_bn(y, p, pre + ".conv3")
            y = y + shortcut
            x = F.relu_(y)
        outs[name] = x
    return outs


def fpn(res, p):
    """detectron2 modeling/backbone/fpn.py FPN.forward + LastLevelMaxPool. Returns [p2, p3, p4, p5, p6]."""
    lat = lambda l, t: F.conv2d(t, p[f"backbone.fpn_lateral{l}.weight"], p[f"backbone.fpn_lateral{l}.bias"])
    out = lambda l, t: F.conv2d(t, p[f"backbone.fpn_output{l}.weight"], p[f"backbone.fpn_output{l}.bias"], padding=1)
    prev = lat(5, res["res5"])
    results = {5: out(5, prev)}
    for l in (4, 3, 2):
        top_down = F.interpolate(prev, scale_factor=2.0, mode="nearest")
        prev = lat(l, res[f"res{l}"]) + top_down
        results[l] = out(l, prev)
    p6 = F.max_pool2d(results[5], kernel_size=1, stride=2, padding=0)
    return [results[2], results[3], results[4], results[5], p6]


def preprocess(images_u8, cfg):
    """GeneralizedRCNN.preprocess_image + ImageList.from_tensors: images_u8 [B,H,W,3] uint8 BGR ->
    float NCHW, (x - mean) / std, zero-padded bottom/right to a multiple of size_divisibility."""
    x = torch.as_tensor(np.ascontiguousarray(images_u8)).permute(0, 3, 1, 2).to(torch.float32)
    mean = torch.tensor(cfg.pixel_mean, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(cfg.pixel_std, dtype=torch.float32).view(1, 3, 1, 1)
    x = (x - mean) / std
    B, _, H, W = x.shape
    d = cfg.size_divisibility
    Hp, Wp = (H + d - 1) // d * d, (W + d - 1) // d * d
    if (Hp, Wp) != (H, W):
        x = F.pad(x, (0, Wp - W, 0, Hp - H), value=0.0)
    return x


# --------------------------------------------------------------------------------------------------- RPN
def cell_anchors(size):
    """detectron2 anchor_generator.py generate_cell_anchors: one size, ratios (0.5, 1, 2); python-float math,
    stored as fp32."""
    rows = []
    area = float(size) ** 2
    for r in ANCHOR_RATIOS:
        w = math.sqrt(area / r)
        h = r * w
        rows.append([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0])
    return torch.tensor(rows, dtype=torch.float32)


def grid_anchors(h, w, stride, size):
    """DefaultAnchorGenerator._grid_anchors with offset 0: flattened order (H, W, A)."""
    sx = torch.arange(0, w * stride, step=stride, dtype=torch.float32)
    sy = torch.arange(0, h * stride, step=stride, dtype=torch.float32)
    yy, xx = torch.meshgrid(sy, sx, indexing="ij")
    shifts = torch.stack((xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)), dim=1)
    return (shifts.view(-1, 1, 4) + cell_anchors(size).view(1, -1, 4)).reshape(-1, 4)


def rpn_head(features, p):
    """StandardRPNHead.forward; returns per level (logits [B, H*W*A], deltas [B, H*W*A, 4]) in (H, W, A) order."""
    pre = "proposal_generator.rpn_head."
    outs = []
    for x in features:
        t = F.relu_(F.conv2d(x, p[pre + "conv.weight"], p[pre + "conv.bias"], padding=1))
        logits = F.conv2d(t, p[pre + "objectness_logits.weight"], p[pre + "objectness_logits.bias"])
        deltas = F.conv2d(t, p[pre + "anchor_deltas.weight"], p[pre + "anchor_deltas.bias"])
        B, A, H, W = logits.shape
        logits = logits.permute(0, 2, 3, 1).flatten(1)
        deltas = deltas.view(B, A, 4, H, W).permute(0, 3, 4, 1, 2).flatten(1, -2)
        outs.append((logits, deltas))
    return outs


def apply_deltas(deltas, boxes, weights):
    """detectron2 box_regression.py Box2BoxTransform.apply_deltas. deltas [N, k*4], boxes [N, 4] -> [N, k*4]."""
    deltas = deltas.float()
    boxes = boxes.to(deltas.dtype)
    widths = boxes[:, 2] - boxes[:, 0]
    heights = boxes[:, 3] - boxes[:, 1]
    ctr_x = boxes[:, 0] + 0.5 * widths
    ctr_y = boxes[:, 1] + 0.5 * heights
    wx, wy, ww, wh = weights
    dx = deltas[:, 0::4] / wx
    dy = deltas[:, 1::4] / wy
    dw = deltas[:, 2::4] / ww
    dh = deltas[:, 3::4] / wh
    dw = torch.clamp(dw, max=SCALE_CLAMP)
    dh = torch.clamp(dh, max=SCALE_CLAMP)
    pred_ctr_x = dx * widths[:, None] + ctr_x[:, None]
    pred_ctr_y = dy * heights[:, None] + ctr_y[:, None]
    pred_w = torch.exp(dw) * widths[:, None]
    pred_h = torch.exp(dh) * heights[:, None]
    x1 = pred_ctr_x - 0.5 * pred_w
    y1 = pred_ctr_y - 0.5 * pred_h
    x2 = pred_ctr_x + 0.5 * pred_w
    y2 = pred_ctr_y + 0.5 * pred_h
    return torch.stack((x1, y1, x2, y2), dim=-1).reshape(deltas.shape)


def clip_boxes(boxes, h, w):
    """Boxes.clip: x to [0, w], y to [0, h]."""
    b = boxes.clone()
    b[..., 0].clamp_(min=0, max=w)
    b[..., 1].clamp_(min=0, max=h)
    b[..., 2].clamp_(min=0, max=w)
    b[..., 3].clamp_(min=0, max=h)
    return b


def sort_desc_stable(scores):
    """Descending order, ties by ascending index (the order this oracle fixes for topk / sort)."""
    s = scores.detach().cpu().numpy()
    return torch.from_numpy(np.argsort(-s, kind="stable"))


def box_iou_1toN(box, boxes):
    """torchvision nms_kernel IoU, fp32, op order: inter / (area_i + area_j - inter)."""
    area_i = (box[2] - box[0]) * (box[3] - box[1])
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    xx1 = torch.maximum(box[0], boxes[:, 0])
    yy1 = torch.maximum(box[1], boxes[:, 1])
    xx2 = torch.minimum(box[2], boxes[:, 2])
    yy2 = torch.minimum(box[3], boxes[:, 3])
    w = torch.clamp(xx2 - xx1, min=0)
    h = torch.clamp(yy2 - yy1, min=0)
    inter = w * h
    return inter / (area_i + areas - inter)


def nms_sorted(boxes, cats, thresh, max_keep=None):
    """Greedy NMS over boxes ALREADY in descending-score order; boxes of different `cats` never suppress each
    other (torchvision batched_nms, `_batched_nms_vanilla` semantics: exact coordinates, no offset trick).
    Suppress when IoU > thresh (strict), torchvision nms_kernel. Returns kept positions (ascending)."""
    n = boxes.shape[0]
    if n == 0:
        return torch.zeros(0, dtype=torch.int64)
    b = boxes.float().numpy()
    c = cats.numpy()
    areas = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    th = np.float32(thresh)
    for i in range(n):
        if suppressed[i]:
            continue
        keep.append(i)
        if max_keep is not None and len(keep) >= max_keep:
            break
        j = slice(i + 1, n)
        xx1 = np.maximum(b[i, 0], b[j, 0])
        yy1 = np.maximum(b[i, 1], b[j, 1])
        xx2 = np.minimum(b[i, 2], b[j, 2])
        yy2 = np.minimum(b[i, 3], b[j, 3])
        w = np.maximum(xx2 - xx1, np.float32(0))
        h = np.maximum(yy2 - yy1, np.float32(0))
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (areas[i] + areas[j] - inter)
        suppressed[j] |= (iou > th) & (c[j] == c[i])
    return torch.tensor(keep, dtype=torch.int64)


def rpn_select_candidates(rpn_outs, feat_shapes, cfg):
    """First half of find_top_rpn_proposals (detectron2 proposal_utils.py): per level top-k by logit, decode.
    Returns per image: (boxes [n,4] unclipped, logits [n], level [n]) concatenated over levels."""
    B = rpn_outs[0][0].shape[0]
    per_img = [[] for _ in range(B)]
    for lvl, ((logits, deltas), (h, w)) in enumerate(zip(rpn_outs, feat_shapes)):
        anchors = grid_anchors(h, w, STRIDES[lvl], ANCHOR_SIZES[lvl])
        k = min(cfg.pre_nms_topk, logits.shape[1])
        for n in range(B):
            order = sort_desc_stable(logits[n])[:k]
            boxes = apply_deltas(deltas[n][order], anchors[order], (1.0, 1.0, 1.0, 1.0))
            per_img[n].append((boxes, logits[n][order], torch.full((k,), lvl, dtype=torch.int64), order))
    out = []
    for n in range(B):
        out.append(tuple(torch.cat([t[i] for t in per_img[n]]) for i in range(4)))
    return out


def rpn_proposals_from_candidates(cands, image_size, cfg):
    """Second half of find_top_rpn_proposals for one image: finite filter, clip, nonempty(0), per-level NMS,
    first post_nms_topk. Returns (proposal_boxes [n,4], logits [n])."""
    boxes, logits, lvl = cands[0], cands[1], cands[2]
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(logits)
    boxes, logits, lvl = boxes[valid], logits[valid], lvl[valid]
    boxes = clip_boxes(boxes, image_size[0], image_size[1])
    keep = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    boxes, logits, lvl = boxes[keep], logits[keep], lvl[keep]
    order = sort_desc_stable(logits)
    boxes, logits, lvl = boxes[order], logits[order], lvl[order]
    kept = nms_sorted(boxes, lvl, cfg.rpn_nms_thresh, max_keep=cfg.post_nms_topk)
    return boxes[kept], logits[kept]


# ------------------------------------------------------------------------------------------------ RoIAlign
def assign_levels(boxes, min_level=2, max_level=5, canonical_box_size=224, canonical_level=4):
    """detectron2 poolers.py assign_boxes_to_levels (fp32)."""
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sizes = torch.sqrt(area)
    lv = torch.floor(canonical_level + torch.log2(sizes / canonical_box_size + 1e-8))
    lv = torch.clamp(lv, min=min_level, max=max_level)
    return lv.to(torch.int64) - min_level


def roi_align(feat, rois, out_size, spatial_scale):
    """torchvision ops roi_align CPU kernel (roi_align_kernel.cpp), aligned=True, sampling_ratio=0, fp32.
    feat [C,H,W] of ONE image; rois [R,4]; returns [R,C,out,out]. Sum order: iy outer, ix inner,
    each sample ((w1*v1 + w2*v2) + w3*v3) + w4*v4, then / count."""
    C, H, W = feat.shape
    R = rois.shape[0]
    f32 = np.float32
    fe = feat.permute(1, 2, 0).contiguous().numpy()  # [H,W,C]
    out = np.zeros((R, out_size, out_size, C), dtype=f32)
    r = rois.numpy().astype(f32)
    sc = f32(spatial_scale)
    off = f32(0.5)
    for i in range(R):
        sw = r[i, 0] * sc - off
        sh = r[i, 1] * sc - off
        ew = r[i, 2] * sc - off
        eh = r[i, 3] * sc - off
        rw = f32(ew - sw)
        rh = f32(eh - sh)
        bh = f32(rh / f32(out_size))
        bw = f32(rw / f32(out_size))
        gh = int(np.ceil(f32(rh / f32(out_size))))   # fp32 division, as the kernel's T = float
        gw = int(np.ceil(f32(rw / f32(out_size))))
        count = f32(max(gh * gw, 1))
        if gh <= 0 or gw <= 0:
            continue
        ph = np.arange(out_size, dtype=f32)
        # y[ph, iy] = sh + ph*bh + (iy + .5) * bh / gh
        ys = sh + ph[:, None] * bh + (np.arange(gh, dtype=f32)[None, :] + f32(0.5)) * bh / f32(gh)
        xs = sw + ph[:, None] * bw + (np.arange(gw, dtype=f32)[None, :] + f32(0.5)) * bw / f32(gw)

        def prep(v, size):
            bad = (v < -1.0) | (v > size)
            v = np.where(v <= 0, f32(0), v)
            lo = v.astype(np.int32)
            hi_clip = lo >= size - 1
            lo = np.where(hi_clip, size - 1, lo)
            hi = np.where(hi_clip, size - 1, lo + 1)
            v = np.where(hi_clip, lo.astype(f32), v)
            l = (v - lo.astype(f32)).astype(f32)
            h = (f32(1) - l).astype(f32)
            return bad, lo, hi, l, h

        ybad, ylo, yhi, ly, hy = prep(ys.astype(f32), H)
        xbad, xlo, xhi, lx, hx = prep(xs.astype(f32), W)
        acc = np.zeros((out_size, out_size, C), dtype=f32)
        for iy in range(gh):
            for ix in range(gw):
                w1 = (hy[:, iy, None] * hx[None, :, ix]).astype(f32)[..., None]
                w2 = (hy[:, iy, None] * lx[None, :, ix]).astype(f32)[..., None]
                w3 = (ly[:, iy, None] * hx[None, :, ix]).astype(f32)[..., None]
                w4 = (ly[:, iy, None] * lx[None, :, ix]).astype(f32)[..., None]
                v1 = fe[ylo[:, iy][:, None], xlo[:, ix][None, :]]
                v2 = fe[ylo[:, iy][:, None], xhi[:, ix][None, :]]
                v3 = fe[yhi[:, iy][:, None], xlo[:, ix][None, :]]
                v4 = fe[yhi[:, iy][:, None], xhi[:, ix][None, :]]
                val = ((w1 * v1 + w2 * v2) + w3 * v3) + w4 * v4
                bad = (ybad[:, iy][:, None] | xbad[:, ix][None, :])[..., None]
                acc = acc + np.where(bad, f32(0), val).astype(f32)
        out[i] = acc / count
    return torch.from_numpy(out).permute(0, 3, 1, 2).contiguous()


def roi_pool(features, boxes_per_img, out_size):
    """detectron2 ROIPooler.forward over p2..p5 (aligned ROIAlign, sampling_ratio 0). Output rows follow the
    order of boxes (image-major). Returns ([R,C,out,out], level [R], batch_idx [R])."""
    C = features[0].shape[1]
    boxes = torch.cat(boxes_per_img) if boxes_per_img else torch.zeros(0, 4)
    bidx = torch.cat([torch.full((len(b),), i, dtype=torch.int64) for i, b in enumerate(boxes_per_img)])
    lv = assign_levels(boxes)
    out = torch.zeros(len(boxes), C, out_size, out_size)
    for l in range(4):
        for b in range(len(boxes_per_img)):
            sel = torch.nonzero((lv == l) & (bidx == b)).squeeze(1)
            if len(sel):
                out[sel] = roi_align(features[l][b], boxes[sel], out_size, 1.0 / STRIDES[l])
    return out, lv, bidx


# ----------------------------------------------------------------------------------------------- ROI heads
def box_head(x, p):
    """FastRCNNConvFCHead (2 FC) + FastRCNNOutputLayers.forward. x [R,C,7,7] -> (scores [R,K+1], deltas [R,4K])."""
    x = torch.flatten(x, start_dim=1)
    x = F.relu(F.linear(x, p["roi_heads.box_head.fc1.weight"], p["roi_heads.box_head.fc1.bias"]))
    x = F.relu(F.linear(x, p["roi_heads.box_head.fc2.weight"], p["roi_heads.box_head.fc2.bias"]))
    scores = F.linear(x, p["roi_heads.box_predictor.cls_score.weight"], p["roi_heads.box_predictor.cls_score.bias"])
    deltas = F.linear(x, p["roi_heads.box_predictor.bbox_pred.weight"], p["roi_heads.box_predictor.bbox_pred.bias"])
    return scores, deltas


def box_inference_single(scores, deltas, proposals, image_size, cfg):
    """FastRCNNOutputLayers.predict_boxes/predict_probs + fast_rcnn_inference_single_image (fast_rcnn.py).
    Returns (boxes [N,4], scores [N], classes [N]) sorted by descending score."""
    K = scores.shape[1] - 1
    boxes = apply_deltas(deltas, proposals, cfg.bbox_reg_weights)  # [R, 4K]
    probs = F.softmax(scores, dim=-1)
    valid = torch.isfinite(boxes).all(dim=1) & torch.isfinite(probs).all(dim=1)
    boxes, probs = boxes[valid], probs[valid]
    probs = probs[:, :-1]
    boxes = clip_boxes(boxes.reshape(-1, 4), image_size[0], image_size[1]).view(-1, K, 4)
    mask = probs > cfg.score_thresh
    inds = mask.nonzero()
    b = boxes[mask]
    s = probs[mask]
    order = sort_desc_stable(s)
    b, s, cls = b[order], s[order], inds[order, 1]
    kept = nms_sorted(b, cls, cfg.nms_thresh)
    if cfg.detections_per_image >= 0:
        kept = kept[: cfg.detections_per_image]
    return b[kept], s[kept], cls[kept]


def mask_head(x, classes, p):
    """MaskRCNNConvUpsampleHead.layers + mask_rcnn_inference: x [N,C,14,14] -> prob of the predicted class [N,28,28]."""
    pre = "roi_heads.mask_head."
    for i in range(1, 5):
        x = F.relu(F.conv2d(x, p[f"{pre}mask_fcn{i}.weight"], p[f"{pre}mask_fcn{i}.bias"], padding=1))
    x = F.relu(F.conv_transpose2d(x, p[pre + "deconv.weight"], p[pre + "deconv.bias"], stride=2))
    logits = F.conv2d(x, p[pre + "predictor.weight"], p[pre + "predictor.bias"])
    idx = torch.arange(x.shape[0])
    return logits[idx, classes].sigmoid()


def paste_prob(prob, box, img_h, img_w):
    """The sampled probability of detectron2's _do_paste_mask (skip_empty=True) BEFORE the threshold: (m [h,w] float32, y0, x0) for
    the region [floor(min)-1, ceil(max)+1) clamped to the image, or (None, 0, 0) when it is empty."""
    b = box.view(1, 4)
    x0_int, y0_int = torch.clamp(b.min(dim=0).values.floor()[:2] - 1, min=0).to(dtype=torch.int32)
    x1_int = torch.clamp(b[:, 2].max().ceil() + 1, max=img_w).to(dtype=torch.int32)
    y1_int = torch.clamp(b[:, 3].max().ceil() + 1, max=img_h).to(dtype=torch.int32)
    x0, y0, x1, y1 = torch.split(b, 1, dim=1)
    img_y = torch.arange(int(y0_int), int(y1_int), dtype=torch.float32) + 0.5
    img_x = torch.arange(int(x0_int), int(x1_int), dtype=torch.float32) + 0.5
    img_y = (img_y - y0) / (y1 - y0) * 2 - 1
    img_x = (img_x - x0) / (x1 - x0) * 2 - 1
    gx = img_x[:, None, :].expand(1, img_y.size(1), img_x.size(1))
    gy = img_y[:, :, None].expand(1, img_y.size(1), img_x.size(1))
    grid = torch.stack([gx, gy], dim=3)
    if grid.shape[1] == 0 or grid.shape[2] == 0:
        return None, 0, 0
    m = F.grid_sample(prob.view(1, 1, *prob.shape).float(), grid, align_corners=False)
    return m[0, 0], int(y0_int), int(x0_int)


def paste_mask(prob, box, img_h, img_w, threshold=0.5):
    """detectron2 layers/mask_ops.py paste_masks_in_image, CPU path: one mask per chunk, _do_paste_mask with
    skip_empty=True (region [floor(min)-1, ceil(max)+1) clamped to the image), bilinear grid_sample
    (align_corners=False, zero padding), then >= threshold. prob [28,28], box [4] -> bool [img_h, img_w]."""
    out = torch.zeros((img_h, img_w), dtype=torch.bool)
    m, y0, x0 = paste_prob(prob, box, img_h, img_w)
    if m is not None:
        out[y0:y0 + m.shape[0], x0:x0 + m.shape[1]] = m >= threshold
    return out


def detector_postprocess(boxes, scores, classes, mask_probs, image_size, out_h, out_w, cfg):
    """detectron2 modeling/postprocessing.py detector_postprocess: rescale + clip boxes, drop empty, paste masks."""
    sx, sy = out_w / image_size[1], out_h / image_size[0]
    b = boxes.clone()
    b[:, 0::2] *= sx
    b[:, 1::2] *= sy
    b = clip_boxes(b, out_h, out_w)
    keep = ((b[:, 2] - b[:, 0]) > 0) & ((b[:, 3] - b[:, 1]) > 0)
    b, scores, classes, mask_probs = b[keep], scores[keep], classes[keep], mask_probs[keep]
    masks = torch.zeros((len(b), out_h, out_w), dtype=torch.bool)
    for i in range(len(b)):
        masks[i] = paste_mask(mask_probs[i], b[i], out_h, out_w, cfg.mask_threshold)
    return b, scores, classes, masks, mask_probs


# ---------------------------------------------------------------------------------------------- end to end
def infer(images_u8, params, cfg, out_sizes=None, stages=None, image_sizes=None):
    """GeneralizedRCNN.inference on a batch of BGR uint8 images [B,H,W,3].
    Returns per image dict(boxes [N,4] f32, scores [N] f32, classes [N] i64, masks [N,H,W] bool).
    `stages`, if a dict, receives intermediate tensors for stage-wise parity tests.
    image_sizes: per image (h, w) of the valid top-left part of the common frame (ImageList.from_tensors: the normalised image is
    padded with 0; proposals and detections are clipped to, and results rescaled from, each image's OWN size); None = the frame."""
    B, H, W, _ = images_u8.shape
    sizes = [(H, W)] * B if image_sizes is None else [tuple(int(v) for v in s) for s in image_sizes]
    with torch.no_grad():
        x = preprocess(images_u8, cfg)
        if image_sizes is not None:
            x = x.clone()
            for b, (h_b, w_b) in enumerate(sizes):
                x[b, :, h_b:, :] = 0.0
                x[b, :, :, w_b:] = 0.0
        res = resnet50(x, params, cfg)
        feats = fpn(res, params)
        rpn_outs = rpn_head(feats, params)
        shapes = [(f.shape[2], f.shape[3]) for f in feats]
        cands = rpn_select_candidates(rpn_outs, shapes, cfg)
        props = [rpn_proposals_from_candidates(c, sizes[n], cfg) for n, c in enumerate(cands)]
        prop_boxes = [pb for pb, _ in props]
        pooled, _, _ = roi_pool(feats[:4], prop_boxes, 7)
        scores, deltas = box_head(pooled, params)
        if stages is not None:
            stages.update(x=x, res=res, feats=feats, rpn_outs=rpn_outs, cands=cands, props=props, pooled=pooled,
                          box_scores=scores, box_deltas=deltas)
        dets = []
        o = 0
        for n in range(B):
            r = len(prop_boxes[n])
            dets.append(box_inference_single(scores[o:o + r], deltas[o:o + r], prop_boxes[n], sizes[n], cfg))
            o += r
        mpooled, _, _ = roi_pool(feats[:4], [d[0] for d in dets], 14)
        mprob = mask_head(mpooled, torch.cat([d[2] for d in dets]), params)
        if stages is not None:
            stages.update(dets=dets, mask_pooled=mpooled, mask_prob=mprob)
        results = []
        o = 0
        for n in range(B):
            nd = len(dets[n][0])
            oh, ow = sizes[n] if out_sizes is None else out_sizes[n]
            b, s, c, m, mp = detector_postprocess(dets[n][0], dets[n][1], dets[n][2], mprob[o:o + nd], sizes[n], oh, ow, cfg)
            o += nd
            # mask_prob: the 28x28 probabilities the masks were pasted from (oracle/gate.py re-samples them where a pixel differs)
            results.append(dict(boxes=b, scores=s, classes=c, masks=m, mask_prob=mp))
    return results
