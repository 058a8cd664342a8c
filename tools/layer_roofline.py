"""Per-launch roofline of the convolutions of ONE inference step of the bench workload (BASELINE configs[1]).

Every conv launch of a step (HIP events attached to the dispatch, amp_prof_launches) with its GEMM shape, duration, algorithmic flops and
algorithmic bytes (every operand once), the bound that applies to it -- the longer of flops / matrix peak and bytes / 8 TB/s -- and the
fraction of that bound it reaches.  The MFMA-busy counter of a kernel that spends its launches on short-K 1x1 layers says little: those
layers are bounded by their bytes.  Usage: python tools/layer_roofline.py [--mode f16x3|f32] [--json out.json]
"""
import sys, os, json, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ampis_amd import ops, synth, params as P
from ampis_amd.model import MaskRCNN


def table(L, steps, mode, title):
    per = len(L) // steps
    assert per * steps == len(L), (len(L), steps)
    peak = (bench.PEAK_F16X3_TFLOPS if mode == "f16x3" else bench.PEAK_F32_MFMA_TFLOPS) * 1e12
    rows = []
    for i in range(per):
        rs = [L[s * per + i] for s in range(steps)]
        r = dict(rs[0])
        r["ms"] = sorted(x["ms"] for x in rs)[len(rs) // 2]          # median over the steps
        t_m, t_h = r["flops"] / peak * 1e3, r["bytes"] / (bench.PEAK_HBM_TBS * 1e12) * 1e3
        r["bound"] = "mfma" if t_m >= t_h else "hbm"
        r["frac"] = max(t_m, t_h) / r["ms"]
        r["tflops"] = r["flops"] / r["ms"] / 1e9
        r["tbs"] = r["bytes"] / r["ms"] / 1e9
        rows.append(r)
    print(f"# {title}, mode {mode}: {per} launches, median of {steps} steps; matrix peak {peak / 1e12:.1f} TFLOP/s, HBM {bench.PEAK_HBM_TBS} TB/s")
    print(f"{'#':>3} {'kind':>5} {'M':>8} {'N':>5} {'K':>8} {'us':>8} {'TFLOP/s':>8} {'TB/s':>6} {'bound':>5} {'frac':>5}")
    for i, r in enumerate(rows):
        kind = "wgrad" if r["slot"] == 2 else "conv"
        print(f"{i:3d} {kind:>5} {r['M']:8d} {r['N']:5d} {r['K']:8d} {r['ms'] * 1e3:8.1f} {r['tflops']:8.1f} {r['tbs']:6.2f} {r['bound']:>5} {r['frac']:5.2f}")
    for k in ("mfma", "hbm"):
        sel = [r for r in rows if r["bound"] == k]
        if sel:
            ms = sum(r["ms"] for r in sel)
            print(f"# {k}-bound: {len(sel)} launches, {ms:.3f} ms per step, time-weighted frac of the bound {sum(r['frac'] * r['ms'] for r in sel) / ms:.3f}")
    ms = sum(r["ms"] for r in rows)
    print(f"# all: {ms:.3f} ms per step, frac of the per-launch roofline {sum(r['frac'] * r['ms'] for r in rows) / ms:.3f}")
    return rows


def train(ctx, a):
    from ampis_amd.model import PackedGt
    TB, S = bench.TRAIN_BATCH, bench.SIZE
    model = MaskRCNN(ctx, bench.K, max_batch=TB, max_h=S, max_w=S, max_out_hw=S, train=True, max_gt=TB * 800, max_poly_doubles=TB * 800 * 64)
    model.load_params(P.init_params(bench.K, seed=0, style="spread"))
    imgs, gts = synth.batch(TB, S, S, first_index=1000)
    d_imgs = ctx.malloc(imgs.nbytes)
    ctx.h2d(d_imgs, imgs)
    packed = PackedGt(gts)
    ctx.conv_mode = a.mode

    def step(i):
        model.forward_losses(None, packed, seed=i, backward=True, device_ptr=d_imgs, shape=(TB, S, S))
        model.sgd_step(1e-3, 0.9, 1e-4, grad_scale=1.0)
    for i in range(3):
        step(i)
    ctx.prof_begin(max_launches=a.steps * 512)
    for i in range(a.steps):
        step(100 + i)
    torch.cuda.synchronize()
    ctx.prof_end()
    L = ctx.prof_launches()
    rows = table(L, a.steps, a.mode, f"one training step, B={TB} {S}x{S}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"mode": a.mode, "rows": rows}, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="f16x3")
    ap.add_argument("--json", default=None)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--train", action="store_true", help="one TRAINING step at local batch 16 (forward, data- and weight-gradient launches)")
    a = ap.parse_args()
    ctx = ops.torch_context(0)
    B, S = bench.BATCH, bench.SIZE
    if a.train:
        return train(ctx, a)
    model = MaskRCNN(ctx, bench.K, max_batch=B, max_h=S, max_w=S, max_out_hw=S, detections_per_image=bench.DETS)
    model.load_params(P.init_params(bench.K, seed=0, style="spread"))
    model.set_rle_output(1)
    imgs, _ = synth.batch(B, S, S)
    d_imgs = ctx.malloc(imgs.nbytes)
    ctx.h2d(d_imgs, imgs)
    ctx.conv_mode = a.mode
    for _ in range(3):
        model.infer_raw(None, device_ptr=d_imgs, shape=(B, S, S))
    ctx.prof_begin(max_launches=a.steps * 96)
    for _ in range(a.steps):
        model.infer_raw(None, device_ptr=d_imgs, shape=(B, S, S))
    torch.cuda.synchronize()
    ctx.prof_end()
    L = ctx.prof_launches()
    rows = table(L, a.steps, a.mode, f"one inference step, B={B} {S}x{S}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"mode": a.mode, "rows": rows, "by_bound": bench.by_bound(L, a.mode, a.steps)}, f, indent=1)


if __name__ == "__main__":
    main()
