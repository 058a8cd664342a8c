import sys, os, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ampis_amd import _lib, params as P, synth
if os.environ.get('AMP_LIB'):
    _lib.LIB_PATH = os.environ['AMP_LIB']
from ampis_amd.model import MaskRCNN
K, B, H, W, D = 2, 2, 256, 320, 40
params = P.init_params(K, seed=3, style="spread")
batches = [synth.batch(B, H, W, first_index=10 * i)[0] for i in range(7)]
NT = int(os.environ.get("NT", "3"))
models = []
for t in range(NT):
    c = _lib.Context(0)
    m = MaskRCNN(c, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D)
    m.load_params(params)
    models.append((c, m))
want = [models[0][1].infer(b) for b in batches]
bad = []
def work(t):
    m = models[t][1]
    for rep in range(30):
        for bi, b in enumerate(batches):
            got = m.infer(b)
            for ii, (go, wo) in enumerate(zip(got, want[bi])):
                if len(go["boxes"]) != len(wo["boxes"]) or not np.array_equal(go["boxes"], wo["boxes"]) or not np.array_equal(go["scores"], wo["scores"]):
                    d = np.abs(go["boxes"] - wo["boxes"]).max() if len(go["boxes"]) == len(wo["boxes"]) else -1
                    bad.append((t, rep, bi, ii, float(d)))
                    if len(bad) <= 2 and d > 0:
                        i = int(np.nonzero(np.abs(go["boxes"] - wo["boxes"]).max(axis=1))[0][0])
                        mr = m.tap("mask_rois").reshape(-1, 4); db = m.tap("det_boxes")
                        off = sum(len(g["boxes"]) for g in got[:ii])
                        pb = m.tap("prop_boxes")
                        dense = m.tap("box_dense"); pred = m.tap("box_pred"); props = m.tap("prop_boxes"); srt = m.tap("box_sorted")
                        # host decode of every (proposal, class) of this image from the tapped inputs
                        pr = props[ii]; w_ = pr[:, 2] - pr[:, 0]; h_ = pr[:, 3] - pr[:, 1]; cx = pr[:, 0] + 0.5 * w_; cy = pr[:, 1] + 0.5 * h_
                        R1 = pr.shape[0]; pd = pred[ii * R1:(ii + 1) * R1]
                        exp = np.zeros((R1, K, 4), np.float32)
                        for k in range(K):
                            dlt = pd[:, K + 1 + 4 * k: K + 5 + 4 * k]
                            dx, dy = dlt[:, 0] / 10, dlt[:, 1] / 10; dw = np.minimum(dlt[:, 2] / 5, np.log(1000 / 16)); dh = np.minimum(dlt[:, 3] / 5, np.log(1000 / 16))
                            pcx, pcy = dx * w_ + cx, dy * h_ + cy; pw, ph = np.exp(dw) * w_, np.exp(dh) * h_
                            exp[:, k] = np.stack([np.clip(pcx - 0.5 * pw, 0, W), np.clip(pcy - 0.5 * ph, 0, H), np.clip(pcx + 0.5 * pw, 0, W), np.clip(pcy + 0.5 * ph, 0, H)], 1)
                        dd = np.abs(dense[ii].reshape(R1, K, 4) - exp).max(axis=2)
                        nbad = int((dd > 0.01).sum())
                        rows = np.argwhere(dd > 0.01)
                        for (r_, k_) in rows[:20]:
                            print("      prop", r_, "cls", k_, "dense", dense[ii].reshape(R1, K, 4)[r_, k_], "host", exp[r_, k_], "prop box", pr[r_])
                        hit_dense = np.nonzero((np.abs(dense[ii] - go["boxes"][i]).max(axis=1) < 1e-4))[0]
                        hit_sorted = np.nonzero((np.abs(srt[ii] - go["boxes"][i]).max(axis=1) < 1e-4))[0]
                        print(f"   dense rows differing from a host decode of the tapped inputs: {nbad}; returned box found in dense rows {hit_dense[:3]}, in sorted rows {hit_sorted[:3]}", flush=True)
                        print(f"thread {t} batch {bi} img {ii} det {i}: returned {go['boxes'][i]} want {wo['boxes'][i]} | device mask_rois {mr[off + i]} det_boxes {db[ii, i]}", flush=True)
th = [threading.Thread(target=work, args=(t,)) for t in range(NT)]
for t in th: t.start()
for t in th: t.join()
print("threads", NT, "mismatches", len(bad), bad[:10])
