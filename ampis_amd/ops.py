"""Per-stage entry points of libampis_hip.so on torch CUDA tensors (torch = device memory + stream only).

Used by the parity tests and by tools; the end-to-end path (ampis_amd.engine) calls amp_infer instead.
Every function launches on the stream of the Context it is given and never falls back to torch math.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import ConvDesc, check, lib, ptr


def torch_context(device=0):
    """Context bound to torch's current HIP stream on `device` (so torch ops and ours are ordered)."""
    if not torch.cuda.is_available():
        raise _lib.AmpError("no HIP device visible: the ampis_amd hot path has no CPU fallback")
    with torch.cuda.device(device):
        return _lib.Context(device, borrow_stream=torch.cuda.current_stream().cuda_stream)


def _f32c(t):
    assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()), "need contiguous fp32 CUDA tensor"
    return t


FMT_X_SPLIT, FMT_Y_SPLIT, FMT_RES_SPLIT = 1, 2, 4


def split_rows(ctx, t):
    """fp32 tensor [..., C] (C % 32 == 0) -> the same bytes in the split hi|lo' row format (amp_split_weights)."""
    _f32c(t)
    out = torch.empty_like(t)
    check(lib().amp_split_weights(ctx.handle, ptr(t), t.numel() // t.shape[-1], t.shape[-1], ptr(out)), "amp_split_weights")
    return out


def unsplit_rows(ctx, t):
    """the inverse: rows of hi|lo' halves -> fp32 (exact)."""
    _f32c(t)
    out = torch.empty_like(t)
    check(lib().amp_unsplit_rows(ctx.handle, ptr(t), t.numel() // t.shape[-1], t.shape[-1], ptr(out)), "amp_unsplit_rows")
    return out


def conv2d_nhwc(ctx, x, w, scale=None, shift=None, res=None, stride=1, pad=0, relu=False, res_mode=None,
                deconv2x2=False, mask=None, scatter2=False, out=None, fmt=0):
    """x [B,H,W,Cin], w [Cout,KH,KW,Cin] -> y [B,Ho,Wo,Cout] (or [B,2Ho,2Wo,Cout/4] when deconv2x2).
    fmt: FMT_* bits -- x / res arrive in, y leaves in the split row format (AMP_CONV_F16X3 only)."""
    _f32c(x), _f32c(w), _f32c(scale), _f32c(shift), _f32c(res)
    B, H, W, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    if res_mode is None:
        res_mode = 0 if res is None else 1
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), int(res_mode), 2 if scatter2 else int(deconv2x2))
    if out is not None:
        y = out
    elif deconv2x2:
        y = torch.empty((B, 2 * Ho, 2 * Wo, Cout // 4), device=x.device, dtype=torch.float32)
    elif scatter2:
        y = torch.zeros((B, 2 * Ho, 2 * Wo, Cout), device=x.device, dtype=torch.float32)
    else:
        y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    if fmt:
        assert mask is None
        check(lib().amp_conv2d_nhwc_fmt(ctx.handle, C.byref(d), ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(res), ptr(y), int(fmt)),
              "amp_conv2d_nhwc_fmt")
        return y
    check(lib().amp_conv2d_nhwc_ex(ctx.handle, C.byref(d), ptr(x), ptr(w), ptr(scale), ptr(shift), ptr(res), ptr(_f32c(mask)), ptr(y)),
          "amp_conv2d_nhwc_ex")
    return y


def bottleneck64_tail(ctx, x_split, w2, scale2, shift2, w3, scale3, shift3, res_split):
    """conv2 (3x3, 64 -> 64) + FrozenBN + ReLU + conv3 (1x1, 64 -> C3) + FrozenBN + shortcut + ReLU of a res2 block in one launch
    (amp_bottleneck64_tail); x_split [B,H,W,64] and res_split [B,H,W,C3] split rows, w2 [64,3,3,64] / w3 [C3,1,1,64] fp32 -> y split rows."""
    _f32c(x_split), _f32c(w2), _f32c(w3), _f32c(res_split)
    B, H, W, _ = x_split.shape
    C3 = w3.shape[0]
    w2s, w3s = split_rows(ctx, w2.reshape(64, 576)), split_rows(ctx, w3.reshape(C3, 64))
    y = torch.empty((B, H, W, C3), device=x_split.device, dtype=torch.float32)
    check(lib().amp_bottleneck64_tail(ctx.handle, B, H, W, ptr(x_split), ptr(w2s), ptr(_f32c(scale2)), ptr(_f32c(shift2)), ptr(w3s), ptr(_f32c(scale3)),
                                      ptr(_f32c(shift3)), C3, ptr(res_split), ptr(y)), "amp_bottleneck64_tail")
    return y


def conv2d_grouped_nhwc(ctx, x, w, groups, scale=None, shift=None, res=None, stride=1, pad=0, relu=False, fmt=0):
    """Grouped conv (ResNeXt conv2): x [B,H,W,C], w [C,KH,KW,C/groups] (grouped OHWI) -> y [B,Ho,Wo,C].
    fmt: FMT_* bits (split-format x / y / res, AMP_CONV_F16X3 only)."""
    _f32c(x), _f32c(w), _f32c(scale), _f32c(shift), _f32c(res)
    B, H, W, Cin = x.shape
    Cout, KH, KW, cpg = w.shape
    assert Cin == Cout and cpg * groups == Cin
    w_win = torch.empty((Cout, KH, KW, 64), device=x.device, dtype=torch.float32)
    check(lib().amp_group_expand_weights(ctx.handle, ptr(w), Cout, KH, KW, cpg, ptr(w_win)), "amp_group_expand_weights")
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W + 2 * pad - KW) // stride + 1
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), 0 if res is None else 1, 0)
    y = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    if fmt:
        check(lib().amp_conv2d_grouped_nhwc_fmt(ctx.handle, C.byref(d), int(groups), ptr(x), ptr(w_win), ptr(scale), ptr(shift), ptr(res), ptr(y), int(fmt)),
              "amp_conv2d_grouped_nhwc_fmt")
        return y
    check(lib().amp_conv2d_grouped_nhwc(ctx.handle, C.byref(d), int(groups), ptr(x), ptr(w_win), ptr(scale), ptr(shift), ptr(res), ptr(y)),
          "amp_conv2d_grouped_nhwc")
    return y


def resize_bilinear_u8(ctx, img, h, w):
    """img uint8 ndarray [H,W,3] (host) -> resized uint8 ndarray [h,w,3]; PIL-exact bilinear on the device (amp_resize_bilinear_u8)."""
    import numpy as np
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W, _ = img.shape
    src = torch.from_numpy(img).to("cuda:%d" % ctx.device)
    dst = torch.empty((h, w, 3), dtype=torch.uint8, device=src.device)
    tmp = torch.empty(int(lib().amp_resize_scratch_bytes(H, W, h, w)), dtype=torch.uint8, device=src.device)
    torch.cuda.synchronize()
    check(lib().amp_resize_bilinear_u8(ctx.handle, ptr(src), H, W, ptr(dst), h, w, ptr(tmp)), "amp_resize_bilinear_u8")
    check(lib().amp_sync(ctx.handle), "amp_sync")
    return dst.cpu().numpy()


def conv2d_wgrad(ctx, x, dy, w_shape, stride=1, pad=0, scale=None, grad=None, dy_shift=0, x_shift=0, x_split=False, bias_grad=None,
                 bias_accumulate=False):
    """x [B,H,W,Cin], dy [B,Ho,Wo,Cout] -> dW [Cout,KH,KW,Cin] (accumulated into `grad` when given).  x_split: x is in the split
    hi|lo' row format (split_rows / a conv with FMT_Y_SPLIT); x_split = 3: dy as well, ALREADY multiplied by 2**dy_shift.  bias_grad [Cout]: also receives (or accumulates) the column sums of dy."""
    _f32c(x), _f32c(dy)
    B, H, W, Cin = x.shape
    Cout, KH, KW, _ = w_shape
    d = ConvDesc(B, H, W, Cin, Cout, KH, KW, stride, pad, 0, 0, 0)
    n = lib().amp_conv_wgrad_scratch_floats(C.byref(d))
    scratch = torch.empty(n, device=x.device)
    acc = grad is not None
    if grad is None:
        grad = torch.empty(w_shape, device=x.device)
    check(lib().amp_conv2d_wgrad_fmt(ctx.handle, C.byref(d), ptr(x), ptr(dy), ptr(scale), ptr(scratch), ptr(grad), int(acc),
                                     int(dy_shift), int(x_shift), int(x_split), ptr(bias_grad), int(bool(bias_accumulate))), "amp_conv2d_wgrad")
    return grad


def dgrad_weights(ctx, w, scale=None):
    Cout, KH, KW, Cin = w.shape
    wt = torch.empty((Cin, KH, KW, Cout), device=w.device)
    check(lib().amp_dgrad_weights(ctx.handle, ptr(_f32c(w)), ptr(scale), Cout, KH, KW, Cin, ptr(wt)), "amp_dgrad_weights")
    return wt


def dgrad_weights_split(ctx, w, scale=None):
    """the data-gradient form of w in the split row format, in one pass (= split_rows(dgrad_weights(w, scale)) bit for bit)."""
    Cout, KH, KW, Cin = w.shape
    wt = torch.empty((Cin, KH, KW, Cout), device=w.device)
    check(lib().amp_dgrad_weights_split(ctx.handle, ptr(_f32c(w)), ptr(scale), Cout, KH, KW, Cin, ptr(wt)), "amp_dgrad_weights_split")
    return wt


def colsum(ctx, dy):
    M, N = dy.shape
    scratch = torch.empty(((M + 511) // 512 + 1) * N, device=dy.device)
    out = torch.empty(N, device=dy.device)
    check(lib().amp_colsum(ctx.handle, ptr(_f32c(dy)), M, N, ptr(scratch), ptr(out), 0), "amp_colsum")
    return out


# ------------------------------------------------------------------------------------------------------------------
# Selection / pooling / mask stages (device tensors in, device tensors out). Integer tensors are int32.
# ------------------------------------------------------------------------------------------------------------------
def _i32(*shape, device="cuda:0"):
    return torch.empty(shape, dtype=torch.int32, device=device)


def _u64(*shape, device="cuda:0"):
    return torch.empty(shape, dtype=torch.int64, device=device)   # same bits; viewed as u64 by the library


def make_rpn_levels(preds, shapes, strides=(4, 8, 16, 32, 64), sizes=(32, 64, 128, 256, 512)):
    lv = _lib.RpnLevels()
    lv.nlevels, lv.A, lv.ld = len(preds), 3, preds[0].shape[-1]
    for i, (p, (h, w)) in enumerate(zip(preds, shapes)):
        _f32c(p)
        lv.pred[i] = p.data_ptr()
        lv.h[i], lv.w[i], lv.stride[i], lv.anchor_size[i] = h, w, strides[i], sizes[i]
    return lv


def rpn_topk(ctx, preds, shapes, B, k):
    """preds: per level [B, h*w, 15]. Returns sel_idx [B,L,k] i32, sel_logit [B,L,k] f32, sel_count [B,L] i32."""
    lv = make_rpn_levels(preds, shapes)
    L = len(preds)
    dev = preds[0].device
    max_n = max(h * w * 3 for h, w in shapes)
    scratch = torch.empty((B * L * max_n,), dtype=torch.int32, device=dev)
    sel_idx, sel_logit, sel_count = _i32(B, L, k, device=dev), torch.zeros((B, L, k), device=dev), _i32(B, L, device=dev)
    check(lib().amp_rpn_topk(ctx.handle, C.byref(lv), B, k, ptr(scratch), max_n, ptr(sel_idx), ptr(sel_logit), ptr(sel_count)),
          "amp_rpn_topk")
    return sel_idx, sel_logit, sel_count


def rpn_decode(ctx, preds, shapes, B, k, sel_idx, sel_logit, sel_count, img_h, img_w):
    lv = make_rpn_levels(preds, shapes)
    cap = len(preds) * k
    dev = preds[0].device
    boxes, keys = torch.empty((B, cap, 4), device=dev), _u64(B, cap, device=dev)
    check(lib().amp_rpn_decode(ctx.handle, C.byref(lv), B, k, ptr(sel_idx), ptr(sel_logit), ptr(sel_count), img_h, img_w, cap,
                               ptr(boxes), ptr(keys), None), "amp_rpn_decode")
    return boxes, keys


def sort_gather(ctx, keys, boxes_in, box_stride=None, n_used=None):
    B, cap = keys.shape
    dev = keys.device
    sb, ss, sc, cnt, pos = torch.empty((B, cap, 4), device=dev), torch.empty((B, cap), device=dev), _i32(B, cap, device=dev), \
        _i32(B, device=dev), _i32(B, cap, device=dev)
    check(lib().amp_sort_gather_n(ctx.handle, B, cap, box_stride or boxes_in.shape[1], ptr(keys), ptr(boxes_in), ptr(sb), ptr(ss),
                                  ptr(sc), ptr(cnt), ptr(pos), None, None, ptr(n_used) if n_used is not None else None), "amp_sort_gather")
    return sb, ss, sc, cnt, pos


def nms(ctx, boxes, cats, counts, thresh, max_keep):
    """boxes [B,cap,4] sorted by descending score, cats [B,cap] i32, counts [B] i32 -> keep_idx [B,max_keep], keep_count [B]."""
    B, cap, _ = boxes.shape
    dev = boxes.device
    W = (cap + 63) // 64
    mask = _u64(B * cap * W, device=dev)
    keep, kc = _i32(B, max_keep, device=dev), _i32(B, device=dev)
    check(lib().amp_nms(ctx.handle, B, cap, ptr(boxes), ptr(cats), ptr(counts), float(thresh), max_keep, ptr(mask), ptr(keep),
                        ptr(kc)), "amp_nms")
    return keep, kc


def rpn_nms_levels(ctx, cand_boxes, cand_keys, sel_count, k, thresh, max_keep, payload=None):
    """Per-level NMS of amp_rpn_decode's candidates + merge: (boxes [B,max_keep,4], logits, levels, count [B], payload or None)."""
    B, cap, _ = cand_boxes.shape
    L = sel_count.shape[1]
    dev = cand_boxes.device
    scratch = _u64(lib().amp_rpn_nms_scratch_words(B, L, k), device=dev)
    pb, ps = torch.empty((B, max_keep, 4), device=dev), torch.empty((B, max_keep), device=dev)
    pl, pc = _i32(B, max_keep, device=dev), _i32(B, device=dev)
    po = _i32(B, max_keep, device=dev) if payload is not None else None
    check(lib().amp_rpn_nms_levels(ctx.handle, B, L, k, cap, ptr(cand_boxes), ptr(cand_keys), ptr(sel_count), float(thresh), max_keep,
                                   ptr(scratch), ptr(pb), ptr(ps), ptr(pl), ptr(pc), ptr(payload) if payload is not None else None,
                                   ptr(po) if po is not None else None), "amp_rpn_nms_levels")
    return pb, ps, pl, pc, po


def make_fpn_feats(feats, strides=(4, 8, 16, 32)):
    f = _lib.FpnFeats()
    for i, t in enumerate(feats):
        _f32c(t)
        f.feat[i] = t.data_ptr()
        f.h[i], f.w[i], f.stride[i] = t.shape[1], t.shape[2], strides[i]
    f.C = feats[0].shape[3]
    return f


def roi_align(ctx, feats, rois, batch_idx, P, fmt=0):
    """feats: [p2..p5] NHWC; rois [R,4]; batch_idx [R] i32 -> ([R,P,P,C], level [R] i32).
    fmt: FMT_X_SPLIT = the feature maps are split rows, FMT_Y_SPLIT = write the pooled tensor as split rows."""
    f = make_fpn_feats(feats)
    R = rois.shape[0]
    out = torch.empty((R, P, P, f.C), device=rois.device)
    lvl = _i32(R, device=rois.device)
    if fmt:
        check(lib().amp_roi_align_fmt(ctx.handle, C.byref(f), ptr(_f32c(rois)), ptr(batch_idx), None, R, P, ptr(out), ptr(lvl), int(fmt)),
              "amp_roi_align_fmt")
    else:
        check(lib().amp_roi_align(ctx.handle, C.byref(f), ptr(_f32c(rois)), ptr(batch_idx), None, R, P, ptr(out), ptr(lvl)),
              "amp_roi_align")
    return out, lvl


def roi_align_bwd(ctx, dfeats, strides, rois, batch_idx, P, dout, B=0):
    """dfeats[l] [B,h,w,256] += RoIAlign-backward(dout [R,P,P,256]) (amp_roi_align_bwd_batched; B = 0: derived from batch_idx)."""
    import ctypes as C_
    ptrs = (C.c_void_p * 4)(*[f.data_ptr() for f in dfeats])
    fh = (C_.c_int * 4)(*[f.shape[1] for f in dfeats])
    fw = (C_.c_int * 4)(*[f.shape[2] for f in dfeats])
    st = (C_.c_int * 4)(*strides)
    check(lib().amp_roi_align_bwd_batched(ctx.handle, ptrs, fh, fw, st, dfeats[0].shape[3], ptr(_f32c(rois)), ptr(batch_idx), rois.shape[0], P,
                                          ptr(_f32c(dout)), int(B)), "amp_roi_align_bwd_batched")


def box_candidates(ctx, pred, proposals, prop_count, K, score_thresh, img_h, img_w, weights=(10., 10., 5., 5.), ccap=8192):
    B, Rcap, _ = proposals.shape
    dev = pred.device
    dense = torch.empty((B, Rcap * K, 4), device=dev)
    keys, cnt, ovf = _u64(B, ccap, device=dev), _i32(B, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
    w = (C.c_float * 4)(*weights)
    check(lib().amp_box_candidates(ctx.handle, ptr(_f32c(pred)), pred.shape[-1], ptr(_f32c(proposals)), ptr(prop_count), B, Rcap,
                                   K, w, float(score_thresh), img_h, img_w, ptr(dense), ptr(keys), ccap, ptr(cnt), ptr(ovf)),
          "amp_box_candidates")
    return dense, keys, cnt, ovf


def paste_rle(ctx, prob, det_boxes, det_batch, out_h, out_w, in_h, in_w, threshold=0.5, pool_counts=1 << 22):
    """prob [N,28,28], det_boxes [N,4], det_batch [N] i32, out_h/out_w [B] i32 (device).
    Returns (out_boxes [N,4], valid [N], list of uint32 run-length arrays)."""
    N = prob.shape[0]
    dev = prob.device
    ob, valid = torch.empty((N, 4), device=dev), _i32(N, device=dev)
    pool = torch.empty((pool_counts,), dtype=torch.int32, device=dev)
    used, off, ln = torch.zeros(1, dtype=torch.int64, device=dev), _u64(N, device=dev), _i32(N, device=dev)
    ovf = torch.zeros(1, dtype=torch.int32, device=dev)
    max_hw = int(max(out_h.max().item(), out_w.max().item()))
    check(lib().amp_paste_rle(ctx.handle, ptr(_f32c(prob)), ptr(_f32c(det_boxes)), ptr(det_batch), N, ptr(out_h), ptr(out_w), max_hw,
                              in_h, in_w, float(threshold), ptr(ob), ptr(valid), ptr(pool), pool_counts, ptr(used), ptr(off),
                              ptr(ln), ptr(ovf)), "amp_paste_rle")
    torch.cuda.synchronize()
    assert int(ovf.item()) == 0, "RLE pool overflow"
    pool_h = pool.cpu().numpy().view("uint32")
    off_h, ln_h = off.cpu().numpy(), ln.cpu().numpy()
    runs = [pool_h[int(o): int(o) + int(l)].copy() for o, l in zip(off_h, ln_h)]
    return ob, valid, runs


def rle_strings_device(ctx, pool, off, ln):
    """COCO counts strings of many masks, encoded on the device (amp_rle_strings_device): pool uint32 run lengths (int32 / uint32 tensor on
    the device), mask i = pool[off[i] : off[i] + ln[i]].  Returns the list of bytes objects."""
    import torch
    n = int(off.numel())
    dev = pool.device
    off = off.to(dev).to(torch.int64).contiguous()
    ln = ln.to(dev).to(torch.int32).contiguous()
    pool = pool.contiguous()
    cap = 7 * int(pool.numel()) + 8
    buf = torch.empty(cap, dtype=torch.uint8, device=dev)
    soff = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    slen = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    check(lib().amp_rle_strings_device(ctx.handle, C.c_void_p(pool.data_ptr()), C.c_void_p(off.data_ptr()), C.c_void_p(ln.data_ptr()), n,
                                       C.c_void_p(buf.data_ptr()), C.c_ulonglong(cap), C.c_void_p(soff.data_ptr()),
                                       C.c_void_p(slen.data_ptr()), C.c_void_p(total.data_ptr())), "amp_rle_strings_device")
    torch.cuda.synchronize()
    t = int(total.item())
    assert t <= cap
    raw = bytes(buf[:t].cpu().numpy().tobytes())
    so, sl = soff.cpu().tolist(), slen.cpu().tolist()
    return [raw[so[i]: so[i] + sl[i]] for i in range(n)]
