"""amp_rpn_nms_levels (NMS per (image, level) segment + merge of the survivors' lists) against
 (a) the one-list chain it replaces in the model: amp_sort_gather -> amp_nms(cats = level) -> first max_keep, bit for bit, and
 (b) the oracle's find_top_rpn_proposals (detectron2 proposal_utils.py) on the same RPN outputs.
Cases: clustered top-k lists (suppression across chunks and across the 1024-box super-blocks), k = 2000 (training), levels with fewer
anchors than k, invalid candidates (non-finite deltas, boxes clipped to nothing), exact score ties between levels, a small max_keep
(early stop)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _preds(B, rng, shapes, smooth, ties, bad):
    preds = []
    for (h, w) in shapes:
        p = rng.normal(0, 2, (B, h * w, 16)).astype(np.float32)
        if smooth:      # spatially smooth objectness: the top-k of a level are neighbours -> long suppression chains
            yy, xx = np.mgrid[0:h, 0:w]
            bump = 6 * np.exp(-(((yy - h * 0.4) / (0.2 * h + 1)) ** 2 + ((xx - w * 0.6) / (0.2 * w + 1)) ** 2))
            p[:, :, :3] = p[:, :, :3] * 0.3 + bump.reshape(1, -1, 1).astype(np.float32)
        if ties:
            p[:, :, :3] = np.round(p[:, :, :3] * 2) / 2
        p[:, :, 3:] *= 0.05 if smooth else 0.3
        p[:, :, 15] = 0
        if bad:
            idx = rng.integers(0, h * w, max(1, h * w // 50))
            p[0, idx, 3] = np.nan                     # non-finite box
            idx = rng.integers(0, h * w, max(1, h * w // 50))
            p[B - 1, idx, 3] = 1e4                    # far outside: clipped to an empty box
        preds.append(torch.from_numpy(p))
    return preds


@pytest.mark.parametrize("k,max_keep,smooth,ties,bad", [
    (1000, 1000, False, False, False),
    (1000, 1000, True, False, True),
    (1000, 300, True, True, False),
    (2000, 1000, True, False, True),
    (2000, 2000, False, True, False),
    (300, 1000, True, False, False),
    (70, 50, False, False, True),
])
def test_levels_equal_one_list_chain_and_oracle(gpu_ctx, k, max_keep, smooth, ties, bad):
    from ampis_amd import ops
    from oracle import maskrcnn as O
    rng = np.random.default_rng(k + max_keep + 3 * smooth + 5 * ties + 7 * bad)
    B, H, W = 3, 512, 384
    shapes = [(128, 96), (64, 48), (32, 24), (16, 12), (8, 6)]        # 36 864 ... 144 anchors: the last levels hold fewer than k
    preds = _preds(B, rng, shapes, smooth, ties, bad)
    dp = [p.to(DEV) for p in preds]
    thresh = 0.7
    si, sl, sc = ops.rpn_topk(gpu_ctx, dp, shapes, B, k)
    boxes, keys = ops.rpn_decode(gpu_ctx, dp, shapes, B, k, si, sl, sc, H, W)
    payload = torch.arange(B * boxes.shape[1], dtype=torch.int32, device=DEV).reshape(B, -1) * 3 + 1
    pb, ps, pl, pc, po = ops.rpn_nms_levels(gpu_ctx, boxes, keys, sc, k, thresh, max_keep, payload=payload)
    # (a) the one-list chain
    sb, ss, scat, cnt, pos = ops.sort_gather(gpu_ctx, keys, boxes)
    keep, kc = ops.nms(gpu_ctx, sb, scat, cnt, thresh, max_keep)
    torch.cuda.synchronize()
    total_suppressed = 0
    for b in range(B):
        n = int(kc[b].item())
        assert int(pc[b].item()) == n, (b, int(pc[b].item()), n)
        idx = keep[b, :n].long()
        assert torch.equal(pb[b, :n], sb[b][idx]), b
        assert torch.equal(ps[b, :n], ss[b][idx])
        assert torch.equal(pl[b, :n], scat[b][idx])
        assert torch.equal(po[b, :n], payload[b][pos[b][idx].long()])
        assert torch.all(pb[b, n:] == 0) and torch.all(ps[b, n:] == 0) and torch.all(pl[b, n:] == -1) and torch.all(po[b, n:] == -1)
        total_suppressed += int(cnt[b].item()) - n
    if smooth:
        assert total_suppressed > 200        # the case does exercise suppression
    # (b) the oracle on the same RPN outputs
    cfg = O.Cfg(num_classes=2, pre_nms_topk=k, post_nms_topk=max_keep, rpn_nms_thresh=thresh)
    outs = [(p[:, :, :3].reshape(B, -1), p[:, :, 3:15].reshape(B, -1, 4)) for p in preds]
    cands = O.rpn_select_candidates(outs, shapes, cfg)
    for b in range(B):
        rb, rl = O.rpn_proposals_from_candidates(cands[b], (H, W), cfg)
        n = int(pc[b].item())
        assert n == len(rb), (b, n, len(rb))
        assert np.array_equal(ps[b, :n].cpu().numpy(), rl.numpy())          # same survivors in the same order (logits exact)
        assert np.abs(pb[b, :n].cpu().numpy() - rb.numpy()).max() < 1e-4


def test_single_level_and_empty_levels(gpu_ctx):
    """One level only; a level whose candidates are all invalid; an image without any valid candidate."""
    from ampis_amd import ops
    rng = np.random.default_rng(5)
    B, L, k = 2, 3, 128
    cap = L * k
    boxes = torch.zeros((B, cap, 4))
    keys = torch.zeros((B, cap), dtype=torch.int64)
    sel = torch.tensor([[128, 40, 128], [128, 128, 0]], dtype=torch.int32)

    def make_sortkeys(score, pos, cat):          # csrc/common.h make_sortkey
        u = score.astype(np.float32).view(np.uint32).astype(np.uint64)
        o = np.where(u & 0x80000000, ~u & 0xffffffff, u | 0x80000000).astype(np.uint64)
        return ((o << np.uint64(32)) | ((np.uint64(0xffffff) - pos.astype(np.uint64)) << np.uint64(8)) | cat.astype(np.uint64)).view(np.int64)
    for b in range(B):
        off = 0
        for l in range(L):
            n = int(sel[b, l])
            c = rng.uniform(0, 200, (n, 2)); s = rng.uniform(10, 60, (n, 2))
            boxes[b, off:off + n] = torch.from_numpy(np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32))
            sc_ = np.sort(rng.normal(0, 1, n).astype(np.float32))[::-1].copy()
            kk = make_sortkeys(sc_, np.arange(off, off + n), np.full(n, l))
            if (b, l) == (0, 1) or b == 1:
                kk[:] = 0                      # all invalid
            keys[b, off:off + n] = torch.from_numpy(kk)
            off += n
    db, dk, ds = boxes.to(DEV), keys.to(DEV), sel.to(DEV)
    pb, ps, pl, pc, _ = ops.rpn_nms_levels(gpu_ctx, db, dk, ds, k, 0.5, 200)
    sb, ss, scat, cnt, pos = ops.sort_gather(gpu_ctx, dk, db)
    keep, kc = ops.nms(gpu_ctx, sb, scat, cnt, 0.5, 200)
    torch.cuda.synchronize()
    assert int(pc[1].item()) == 0 and int(kc[1].item()) == 0
    n = int(kc[0].item())
    assert int(pc[0].item()) == n and n > 10
    assert torch.equal(pb[0, :n], sb[0][keep[0, :n].long()])
    assert not bool((pl[0, :n] == 1).any())
