"""Stage-level parity on IDENTICAL inputs (through the C ABI per-kernel entry points) against the oracle.

Selection stages are integer / ordering work: bit-exact (same indices, same kept sets, same run lengths).  Stages with
fp32 arithmetic that the oracle evaluates in the same op order (IoU, RoIAlign, paste) are compared exactly or to 1 ulp-level
tolerances; stages that call expf (decode, softmax, sigmoid) to 1e-4 px / 1e-6.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _levels(B, rng, shapes, ties=False):
    preds = []
    for (h, w) in shapes:
        p = rng.normal(0, 2, (B, h * w, 16)).astype(np.float32)
        if ties:   # heavy exact ties: quantise logits so the index tie-break is exercised everywhere
            p[:, :, :3] = np.round(p[:, :, :3] * 2) / 2
        p[:, :, 3:] *= 0.3
        p[:, :, 15] = 0
        preds.append(torch.from_numpy(p))
    return preds


@pytest.mark.parametrize("ties", [False, True])
def test_rpn_topk_exact(gpu_ctx, ties):
    from ampis_amd import ops
    from oracle import maskrcnn as O
    rng = np.random.default_rng(1 + ties)
    B, k = 2, 1000
    shapes = [(64, 48), (32, 24), (16, 12), (8, 6), (4, 3)]
    preds = _levels(B, rng, shapes, ties)
    si, sl, sc = ops.rpn_topk(gpu_ctx, [p.to(DEV) for p in preds], shapes, B, k)
    torch.cuda.synchronize()
    si, sl, sc = si.cpu().numpy(), sl.cpu().numpy(), sc.cpu().numpy()
    for b in range(B):
        for l, p in enumerate(preds):
            logits = p[b, :, :3].reshape(-1)
            kk = min(k, logits.numel())
            assert sc[b, l] == kk
            order = O.sort_desc_stable(logits)[:kk].numpy()
            assert np.array_equal(si[b, l, :kk], order)
            assert np.array_equal(sl[b, l, :kk], logits.numpy()[order])


@pytest.mark.parametrize("ties", [False, True])
def test_rpn_topk_exact_on_chunked_levels(gpu_ctx, ties):
    """Levels with more than 49 152 anchors are selected by several workgroups and merged: same exact order (logit descending,
    ties by ascending anchor index), also when a chunk holds fewer candidates than k and when ties straddle chunk borders."""
    from ampis_amd import ops
    from oracle import maskrcnn as O
    rng = np.random.default_rng(7 + ties)
    B, k = 2, 1000
    shapes = [(256, 200), (130, 127), (64, 50), (32, 25), (16, 13)]      # 153 600 / 49 530 / 9 600 ... anchors: 4 and 2 chunks, then 1
    preds = _levels(B, rng, shapes, ties)
    preds[0][1, 40000:, :3] = -50.0 - torch.arange(preds[0].shape[1] - 40000)[:, None] * 1e-3    # image 1: three of four chunks nearly empty of good logits
    si, sl, sc = ops.rpn_topk(gpu_ctx, [p.to(DEV) for p in preds], shapes, B, k)
    torch.cuda.synchronize()
    si, sl, sc = si.cpu().numpy(), sl.cpu().numpy(), sc.cpu().numpy()
    for b in range(B):
        for l, p in enumerate(preds):
            logits = p[b, :, :3].reshape(-1)
            kk = min(k, logits.numel())
            assert sc[b, l] == kk
            order = O.sort_desc_stable(logits)[:kk].numpy()
            assert np.array_equal(si[b, l, :kk], order), (b, l)
            assert np.array_equal(sl[b, l, :kk], logits.numpy()[order])


def test_rpn_decode_and_sort(gpu_ctx):
    from ampis_amd import ops
    from oracle import maskrcnn as O
    rng = np.random.default_rng(3)
    B, k, H, W = 2, 300, 200, 176
    shapes = [(50, 44), (25, 22), (13, 11), (7, 6), (4, 3)]
    preds = _levels(B, rng, shapes)
    dp = [p.to(DEV) for p in preds]
    si, sl, sc = ops.rpn_topk(gpu_ctx, dp, shapes, B, k)
    boxes, keys = ops.rpn_decode(gpu_ctx, dp, shapes, B, k, si, sl, sc, H, W)
    sb, ss, scat, cnt, pos = ops.sort_gather(gpu_ctx, keys, boxes)
    torch.cuda.synchronize()
    cfg = O.Cfg(num_classes=2, pre_nms_topk=k)
    outs = [(p[:, :, :3].reshape(B, -1), p[:, :, 3:15].reshape(B, -1, 4)) for p in preds]
    cands = O.rpn_select_candidates(outs, shapes, cfg)
    for b in range(B):
        cb, cl, clv = cands[b][0], cands[b][1], cands[b][2]
        cb = O.clip_boxes(cb, H, W)
        keep = ((cb[:, 2] - cb[:, 0]) > 0) & ((cb[:, 3] - cb[:, 1]) > 0) & torch.isfinite(cb).all(1)
        order = O.sort_desc_stable(cl[keep])
        rb, rl, rv = cb[keep][order].numpy(), cl[keep][order].numpy(), clv[keep][order].numpy()
        n = int(cnt[b].item())
        assert n == len(rb)
        assert np.array_equal(ss[b, :n].cpu().numpy(), rl)           # logits exact, order exact
        assert np.array_equal(scat[b, :n].cpu().numpy(), rv)
        assert np.abs(sb[b, :n].cpu().numpy() - rb).max() < 1e-4     # expf: <= a few ulp of the box size


@pytest.mark.parametrize("n,ncat,thresh", [(700, 3, 0.5), (3000, 5, 0.7), (64, 1, 0.3), (1, 1, 0.5), (0, 1, 0.5), (10000, 5, 0.6)])
def test_nms_exact(gpu_ctx, n, ncat, thresh):
    from ampis_amd import ops
    from oracle import maskrcnn as O
    rng = np.random.default_rng(n + 7)
    B, cap = 2, (3072 if n <= 3072 else 10240)
    boxes = np.zeros((B, cap, 4), np.float32)
    cats = np.full((B, cap), -1, np.int32)
    ns = [n, max(n - 5, 0)]
    for b in range(B):
        c = rng.uniform(0, 300 if n <= 3072 else 1000, (ns[b], 2))
        s = rng.uniform(4, 80, (ns[b], 2))
        boxes[b, :ns[b]] = np.concatenate([c - s / 2, c + s / 2], 1)
        cats[b, :ns[b]] = rng.integers(0, ncat, ns[b])
    keep, kc = ops.nms(gpu_ctx, torch.from_numpy(boxes).to(DEV), torch.from_numpy(cats).to(DEV),
                       torch.tensor(ns, dtype=torch.int32, device=DEV), thresh, max_keep=1000)
    torch.cuda.synchronize()
    for b in range(B):
        ref = O.nms_sorted(torch.from_numpy(boxes[b, :ns[b]]), torch.from_numpy(cats[b, :ns[b]].astype(np.int64)), thresh,
                           max_keep=1000).numpy()
        assert int(kc[b].item()) == len(ref)
        assert np.array_equal(keep[b, :len(ref)].cpu().numpy(), ref)


def test_roi_align_matches_oracle(gpu_ctx):
    from ampis_amd import ops
    from oracle import maskrcnn as O
    rng = np.random.default_rng(11)
    B, C = 2, 256
    shapes = [(64, 80), (32, 40), (16, 20), (8, 10)]
    feats = [torch.from_numpy(rng.normal(0, 1, (B, h, w, C)).astype(np.float32)) for h, w in shapes]
    R = 120
    ctr = rng.uniform(0, 1, (R, 2)) * np.array([320, 256])
    size = np.exp(rng.uniform(np.log(4), np.log(500), (R, 2)))
    rois = np.concatenate([ctr - size / 2, ctr + size / 2], 1).astype(np.float32)
    rois = np.clip(rois, 0, [320, 256, 320, 256]).astype(np.float32)
    rois[0] = [10, 10, 10, 30]            # zero width -> empty sampling grid -> zeros
    rois[1] = [0, 0, 320, 256]            # whole image, highest level
    bidx = rng.integers(0, B, R).astype(np.int32)
    for P in (7, 14):
        out, lvl = ops.roi_align(gpu_ctx, [f.to(DEV) for f in feats], torch.from_numpy(rois).to(DEV),
                                 torch.from_numpy(bidx).to(DEV), P)
        torch.cuda.synchronize()
        out, lvl = out.cpu(), lvl.cpu().numpy()
        ref_lvl = O.assign_levels(torch.from_numpy(rois)).numpy()
        assert np.array_equal(lvl, ref_lvl)
        for r in range(R):
            f = feats[ref_lvl[r]][bidx[r]].permute(2, 0, 1)
            ref = O.roi_align(f, torch.from_numpy(rois[r:r + 1]), P, 1.0 / (4 * 2 ** ref_lvl[r]))[0].permute(1, 2, 0)
            assert torch.equal(out[r], ref), (P, r, float((out[r] - ref).abs().max()))


def test_box_inference_chain(gpu_ctx):
    """box_candidates -> sort_gather -> nms -> same detections as fast_rcnn_inference_single_image."""
    from ampis_amd import ops
    from oracle import maskrcnn as O
    rng = np.random.default_rng(5)
    B, Rcap, K, H, W, D = 2, 400, 3, 300, 400, 50
    cfg = O.Cfg(num_classes=K, detections_per_image=D)
    pred = rng.normal(0, 1.5, (B * Rcap, 5 * K + 1)).astype(np.float32)
    pred[:, K + 1:] *= 0.5
    ctr = rng.uniform(0, 1, (B, Rcap, 2)) * np.array([W, H])
    size = rng.uniform(8, 120, (B, Rcap, 2))
    props = np.clip(np.concatenate([ctr - size / 2, ctr + size / 2], 2), 0, [W, H, W, H]).astype(np.float32)
    counts = np.array([Rcap, Rcap - 37], np.int32)
    dense, keys, cnt, ovf = ops.box_candidates(gpu_ctx, torch.from_numpy(pred).to(DEV), torch.from_numpy(props).to(DEV),
                                               torch.from_numpy(counts).to(DEV), K, cfg.score_thresh, H, W)
    sb, ss, sc, scount, _ = ops.sort_gather(gpu_ctx, keys, dense, box_stride=Rcap * K)
    keep, kc = ops.nms(gpu_ctx, sb, sc, scount, cfg.nms_thresh, D)
    # the model's form: only the power of two that holds the compacted candidates is sorted (amp_sort_gather_n) -- same outputs
    sb2, ss2, sc2, scount2, pos2 = ops.sort_gather(gpu_ctx, keys, dense, box_stride=Rcap * K, n_used=cnt)
    torch.cuda.synchronize()
    assert int(ovf.item()) == 0
    assert int(cnt.max().item()) < keys.shape[1] // 2            # the hint does shrink the sort
    assert torch.equal(sb, sb2) and torch.equal(ss, ss2) and torch.equal(sc, sc2) and torch.equal(scount, scount2)
    for b in range(B):
        n = counts[b]
        rows = torch.from_numpy(pred[b * Rcap: b * Rcap + n])
        rb, rs, rc = O.box_inference_single(rows[:, :K + 1], rows[:, K + 1:], torch.from_numpy(props[b, :n]), (H, W), cfg)
        k = int(kc[b].item())
        assert k == len(rb)
        idx = keep[b, :k].long()
        assert np.array_equal(sc[b][idx].cpu().numpy(), rc.numpy())
        assert np.abs(ss[b][idx].cpu().numpy() - rs.numpy()).max() < 2e-6
        assert np.abs(sb[b][idx].cpu().numpy() - rb.numpy()).max() < 2e-4


def test_paste_rle_matches_oracle(gpu_ctx):
    """paste + threshold + RLE on device == oracle paste_mask (F.grid_sample) + oracle RLE encode, run for run."""
    from ampis_amd import ops
    from oracle import maskrcnn as O, rle as orle
    rng = np.random.default_rng(9)
    H, W = 160, 208
    N = 40
    # smooth blobs + a few noisy ones (many runs), probabilities in [0,1]
    yy, xx = np.mgrid[0:28, 0:28]
    prob = np.zeros((N, 28, 28), np.float32)
    for i in range(N):
        cy, cx, r = rng.uniform(8, 20), rng.uniform(8, 20), rng.uniform(4, 12)
        prob[i] = 1 / (1 + np.exp(((yy - cy) ** 2 + (xx - cx) ** 2 - r * r) / 8))
        if i % 5 == 0:
            prob[i] = rng.uniform(0, 1, (28, 28))
    ctr = rng.uniform(0, 1, (N, 2)) * np.array([W, H])
    size = np.exp(rng.uniform(np.log(3), np.log(150), (N, 2)))
    boxes = np.concatenate([ctr - size / 2, ctr + size / 2], 1).astype(np.float32)
    boxes[0] = [-20, -10, W + 30, H + 20]     # covers everything (wrap-around columns)
    boxes[1] = [5.5, 0, 40.25, H]             # full-height columns
    boxes[2] = [50, 60, 50, 90]               # empty after clip -> dropped
    boxes[3] = [W - 10.3, 20.7, W + 5, 70.2]  # touches the right border
    ob, valid, runs = ops.paste_rle(gpu_ctx, torch.from_numpy(prob).to(DEV), torch.from_numpy(boxes).to(DEV),
                                    torch.zeros(N, dtype=torch.int32, device=DEV),
                                    torch.tensor([H], dtype=torch.int32, device=DEV), torch.tensor([W], dtype=torch.int32, device=DEV),
                                    H, W)
    ob, valid = ob.cpu().numpy(), valid.cpu().numpy()
    nflip = 0
    for i in range(N):
        b = O.clip_boxes(torch.from_numpy(boxes[i:i + 1]).clone(), H, W)[0]
        nonempty = bool((b[2] - b[0] > 0) and (b[3] - b[1] > 0))
        assert bool(valid[i]) == nonempty
        if not nonempty:
            assert len(runs[i]) == 0
            continue
        assert np.array_equal(ob[i], b.numpy())
        ref = O.paste_mask(torch.from_numpy(prob[i]), b, H, W, 0.5).numpy()
        assert int(runs[i].sum()) == H * W
        got = orle.decode_counts(runs[i], H, W)
        if not np.array_equal(runs[i], orle.encode_counts(ref)):
            # identical inputs (same probabilities, same box): a pixel may differ only where the interpolated value sits ON the
            # threshold -- the two paths sum the four bilinear products in a different order (last-ulp differences)
            pm, y0, x0 = O.paste_prob(torch.from_numpy(prob[i]), b, H, W)
            ys, xs = np.nonzero(got != ref)
            margin = np.abs(pm.numpy()[ys - y0, xs - x0] - 0.5)
            assert margin.max() < 1e-6, (i, float(margin.max()))
            nflip += len(ys)
    assert nflip <= 4, f"{nflip} pixels differ from the oracle paste"


def test_roi_align_lane_parallel_kernel_equals_the_reference_kernel(gpu_ctx):
    """roi_align_lanes_kernel (sample parameters computed once per bin, one lane per sample row / column, broadcast by v_readlane)
    is the same arithmetic as roi_align_kernel (every lane computes every sample): bit-identical, including sampling grids beyond
    64 samples per side (chunked), boxes outside the map and empty boxes."""
    from ampis_amd import ops, _lib
    g = torch.Generator().manual_seed(5)
    B, H, W, C = 2, 128, 160, 256
    feats = [torch.randn(B, H // s, W // s, C, generator=g).to(DEV) for s in (1, 2, 4, 8)]
    R = 400
    ctr = torch.rand(R, 2, generator=g) * torch.tensor([4.0 * W, 4.0 * H])
    size = torch.exp(torch.rand(R, 2, generator=g) * 6.5 + 0.5)
    rois = torch.cat([ctr - size / 2, ctr + size / 2], 1).float()
    rois[0] = torch.tensor([-300.0, -200.0, -100.0, -50.0])          # entirely outside
    rois[1] = torch.tensor([10.0, 10.0, 10.0, 40.0])                 # zero width
    rois[2] = torch.tensor([0.0, 0.0, 4.0 * W, 4.0 * H])             # the whole image: level p5 ... 
    rois[3] = torch.tensor([0.0, 0.0, 20000.0, 30.0])                # > 64 sample columns per bin on p5 (clamped level)
    rois = rois.to(DEV)
    bidx = (torch.arange(R) % B).int().to(DEV)
    outs = []
    try:
        for v in (0, 1, 3):
            _lib.lib().amp_debug_set_roi_lanes(v)
            outs.append([ops.roi_align(gpu_ctx, feats, rois, bidx, P)[0] for P in (7, 14)])
    finally:
        _lib.lib().amp_debug_set_roi_lanes(1)
    torch.cuda.synchronize()
    for k in (1, 2):          # outs[1]: lane-parallel parameters, outs[2]: one workgroup per bin row, cells staged once in LDS
        for a, b in zip(outs[0], outs[k]):
            assert torch.equal(a, b), k


def test_fused_mask_tail_matches_the_three_stage_chain(gpu_ctx):
    """amp_mask_deconv_predict (ConvTranspose 2x2 + ReLU + predictor row of the class + sigmoid in the deconv's epilogue) against
    (a) the three kernels it replaces and (b) torch in fp64."""
    import ctypes as C
    from ampis_amd import ops, _lib
    g = torch.Generator().manual_seed(12)
    N, K = 37, 3
    x = (torch.randn(N, 14, 14, 256, generator=g).clamp_(min=0) * 2).to(DEV)
    wd = (torch.randn(256, 256, 2, 2, generator=g) * 0.05)                       # ConvTranspose2d weight [Cin][Cout][2][2]
    bd = torch.randn(256, generator=g) * 0.1
    wp = (torch.randn(K, 256, generator=g) * 0.05)
    bp = torch.randn(K, generator=g) * 0.1
    cls = torch.randint(0, K, (N,), generator=g).int()
    # library layout of the deconv: [(ky,kx,co)][ci], bias repeated per tap
    w_lib = wd.permute(2, 3, 1, 0).reshape(1024, 1, 1, 256).contiguous().to(DEV)
    b_lib = bd.repeat(4).contiguous().to(DEV)
    xs = ops.split_rows(gpu_ctx, x)
    prob = torch.empty(N, 28, 28, device=DEV)
    wp_d, bp_d, cls_d = wp.to(DEV), bp.to(DEV), cls.to(DEV)          # (kept alive: the call only takes their addresses)
    _lib.check(_lib.lib().amp_mask_deconv_predict(gpu_ctx.handle, _lib.ptr(xs), N, _lib.ptr(w_lib), _lib.ptr(b_lib), _lib.ptr(wp_d),
                                                  _lib.ptr(bp_d), _lib.ptr(cls_d), K, _lib.ptr(prob)), "amp_mask_deconv_predict")
    # (a) the unfused chain of the library
    y = ops.conv2d_nhwc(gpu_ctx, x, w_lib, None, b_lib, relu=True, deconv2x2=True)                       # [N,28,28,256]
    Kp = 4
    wpp = torch.zeros(Kp, 1, 1, 256); wpp[:K, 0, 0] = wp
    bpp = torch.zeros(Kp); bpp[:K] = bp
    logits = ops.conv2d_nhwc(gpu_ctx, y, wpp.to(DEV), None, bpp.to(DEV))
    torch.cuda.synchronize()
    chain = torch.sigmoid(logits.cpu()[torch.arange(N), :, :, cls.long()])
    # (b) fp64 reference
    ref = torch.nn.functional.conv_transpose2d(x.cpu().double().permute(0, 3, 1, 2), wd.double(), bd.double(), stride=2).relu()
    ref = torch.einsum("nchw,nc->nhw", ref, wp.double()[cls.long()]) + bp.double()[cls.long()].view(-1, 1, 1)
    ref = torch.sigmoid(ref)
    e_chain, e_fused = (chain.double() - ref).abs().max().item(), (prob.cpu().double() - ref).abs().max().item()
    print("max |prob - fp64|: three-stage chain", e_chain, "fused", e_fused)
    assert e_chain < 2e-6 and e_fused < 2e-6
    assert not gpu_ctx.conv_range_flag()


def test_fused_rpn_head_matches_the_two_convolutions(gpu_ctx):
    """amp_rpn_head_fused (3x3 conv + ReLU, then the objectness / anchor-delta rows as a second f16x3 product in the conv's epilogue)
    against (a) the two convolutions it replaces and (b) torch in fp64."""
    from ampis_amd import ops, _lib
    g = torch.Generator().manual_seed(21)
    B, H, W = 2, 112, 120                                        # 26 880 pixels, ragged against the 128-row tiles
    x = (torch.randn(B, H, W, 256, generator=g).clamp_(min=0)).to(DEV)
    wc = (torch.randn(256, 3, 3, 256, generator=g) * 0.02).to(DEV)
    bc = (torch.randn(256, generator=g) * 0.1).to(DEV)
    wp = torch.zeros(16, 1, 1, 256); wp[:15] = torch.randn(15, 1, 1, 256, generator=g) * 0.05
    bp = torch.zeros(16); bp[:15] = torch.randn(15, generator=g) * 0.1
    wp_d, bp_d = wp.to(DEV), bp.to(DEV)
    xs = ops.split_rows(gpu_ctx, x)
    pred = torch.empty(B * H * W, 16, device=DEV)
    _lib.check(_lib.lib().amp_rpn_head_fused(gpu_ctx.handle, _lib.ptr(xs), B, H, W, _lib.ptr(wc), _lib.ptr(bc), _lib.ptr(wp_d), _lib.ptr(bp_d),
                                             _lib.ptr(pred)), "amp_rpn_head_fused")
    t = ops.conv2d_nhwc(gpu_ctx, x, wc, None, bc, pad=1, relu=True)
    chain = ops.conv2d_nhwc(gpu_ctx, t, wp_d, None, bp_d).reshape(-1, 16)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(x.cpu().double().permute(0, 3, 1, 2), wc.cpu().double().permute(0, 3, 1, 2), bc.cpu().double(), padding=1).relu()
    ref = torch.nn.functional.conv2d(ref, wp.double().permute(0, 3, 1, 2), bp.double()).permute(0, 2, 3, 1).reshape(-1, 16)
    m = ref.abs().max().item()
    e_chain, e_fused = (chain.cpu().double() - ref).abs().max().item() / m, (pred.cpu().double() - ref).abs().max().item() / m
    print("max |pred - fp64| / max: two convolutions", e_chain, "fused", e_fused)
    assert e_fused <= max(1.5 * e_chain, 5e-7), (e_fused, e_chain)
    assert float(pred[:, 15].abs().max()) == 0.0                 # the pad row stays a pad row
    assert not gpu_ctx.conv_range_flag()


@pytest.mark.parametrize("P,R", [(7, 3000), (14, 700)])
def test_roi_align_xcd_major_order_is_bit_identical(gpu_ctx, P, R):
    """RoIAlign in XCD-major order (roi_order_kernel: every image's RoIs sorted by the 32-px Morton tile of their centre and dealt to the
    eight XCDs in contiguous eighths -- the default for >= 2048 RoIs) writes every output row exactly as the index order does: split maps in,
    split rows out, several images with unequal RoI counts, an image without RoIs, degenerate boxes."""
    from ampis_amd import ops
    from ampis_amd._lib import lib
    rng = np.random.default_rng(P + R)
    B, C, S = 4, 256, 256
    feats = [torch.from_numpy(rng.normal(0, 1, (B, S // s, S // s, C)).astype(np.float32)).to(DEV) for s in (4, 8, 16, 32)]
    fs = [ops.split_rows(gpu_ctx, f) for f in feats]
    ctr = rng.uniform(0, S, (R, 2))
    size = np.exp(rng.normal(3.6, 0.7, (R, 2))).clip(2, 400)
    rois = np.clip(np.concatenate([ctr - size / 2, ctr + size / 2], 1), 0, S).astype(np.float32)
    rois[5] = [10, 10, 10, 40]                               # empty
    bidx = rng.choice([0, 1, 3], R, p=[0.6, 0.1, 0.3]).astype(np.int32)      # image 2 has none
    d_rois, d_b = torch.from_numpy(rois).to(DEV), torch.from_numpy(bidx).to(DEV)
    outs = []
    try:
        for mode in (0, 2):
            lib().amp_debug_set_roi_xcd(mode)
            out, lvl = ops.roi_align(gpu_ctx, fs, d_rois, d_b, P, fmt=3)
            torch.cuda.synchronize()
            outs.append((out.clone(), lvl.clone()))
    finally:
        lib().amp_debug_set_roi_xcd(1)
    assert torch.equal(outs[0][0].view(torch.int32), outs[1][0].view(torch.int32)) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].abs().max()) > 0
