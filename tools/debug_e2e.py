"""Prints per-stage differences between the HIP path and the oracle on a small batch (debug aid, GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ampis_amd import params as P, ops
from ampis_amd.model import MaskRCNN
from ampis_amd import _lib, rle
from oracle import maskrcnn as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_e2e_gpu import synth_image, _nhwc, _relerr, _match_boxes

K, B, H, W, D = 2, 2, 224, 288, 60
rng = np.random.default_rng(5)
imgs = np.stack([synth_image(rng, H, W) for _ in range(B)])
npp = P.init_params(K, seed=3, style="spread")
cfg = O.Cfg(num_classes=K, detections_per_image=D)
st = {}
t = time.time(); ref = O.infer(imgs, O.to_torch_params(npp), cfg, stages=st); print("oracle s", time.time() - t, flush=True)
ctx = _lib.Context(0)
m = MaskRCNN(ctx, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D)
print("workspace MB", m.workspace_bytes / 1e6, flush=True)
m.load_params(npp)
t = time.time(); out = m.infer(imgs); print("hip infer s", time.time() - t, flush=True)
for i, n in enumerate(["res2", "res3", "res4", "res5"]):
    print(n, _relerr(m.tap(n), _nhwc(st["res"][n])))
for i, n in enumerate(["p2", "p3", "p4", "p5", "p6"]):
    print(n, _relerr(m.tap(n), _nhwc(st["feats"][i])))
for i, n in enumerate(["rpn_pred2", "rpn_pred3", "rpn_pred4", "rpn_pred5", "rpn_pred6"]):
    lg, dl = st["rpn_outs"][i]; got = m.tap(n)[:, :, :15]; Bq, HW, _ = got.shape
    r = np.concatenate([lg.numpy().reshape(Bq, HW, 3), dl.numpy().reshape(Bq, HW, 12)], 2)
    print(n, _relerr(got, r))
si, sl = m.tap("rpn_sel_idx"), m.tap("rpn_sel_logit")
for b in range(B):
    c = st["cands"][b]
    # oracle per-level order: first k of level 0 etc
    o = 0
    for l in range(5):
        kk = int((c[2] == l).sum())
        ref_idx = c[3][o:o + kk].numpy(); got_idx = si[b, l, :kk]
        print(f"img{b} lvl{l} k={kk} topk idx equal frac", float((ref_idx == got_idx).mean()), "logit maxdiff", float(np.abs(sl[b, l, :kk] - c[1][o:o + kk].numpy()).max()))
        o += kk
pb, pc = m.tap("prop_boxes"), m.tap("prop_count")
for b in range(B):
    r = st["props"][b][0].numpy(); g = pb[b, :pc[b]]
    print(f"img{b} proposals ref {len(r)} got {len(g)} match", _match_boxes(r, g, 5e-3), "exact-order maxdiff", float(np.abs(r[:min(len(r),len(g))] - g[:min(len(r),len(g))]).max()))
dc = m.tap("det_count"); db = m.tap("det_boxes"); ds = m.tap("det_scores"); dcl = m.tap("det_classes")
for b in range(B):
    r = st["dets"][b]
    n = min(len(r[0]), dc[b])
    print(f"img{b} dets ref {len(r[0])} got {dc[b]} box maxdiff(in order) {float(np.abs(r[0].numpy()[:n]-db[b,:n]).max()):.3e} score maxdiff {float(np.abs(r[1].numpy()[:n]-ds[b,:n]).max()):.3e} cls eq {float((r[2].numpy()[:n]==dcl[b,:n]).mean())}")
mp = m.tap("mask_prob"); rp = st["mask_prob"].numpy(); n = min(len(mp), len(rp))
print("mask_prob maxdiff (in order)", float(np.abs(mp[:n] - rp[:n]).max()))
tot = good = 0
for o, r in zip(out, ref):
    rb, rm = r["boxes"].numpy(), r["masks"].numpy()
    print("final n", len(rb), len(o["boxes"]))
    for i in range(min(len(rb), len(o["boxes"]))):
        tot += 1
        gm = rle.decode(o["masks"][i]).astype(bool)
        u = (gm | rm[i]).sum(); iou = 1.0 if u == 0 else (gm & rm[i]).sum() / u
        ok = np.abs(o["boxes"][i] - rb[i]).max() < 1e-3 and iou >= 0.999
        good += ok
        if not ok and tot < 400 and (tot - good) < 6:
            print("  mismatch", i, o["boxes"][i], rb[i], "iou", iou, "areas", gm.sum(), rm[i].sum())
print("in-order good", good, "/", tot)
