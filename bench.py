"""bench.py — images/s of Mask R-CNN R50-FPN inference on synthetic 1024x1024 micrographs (BASELINE.json configs[1]).

python bench.py --gpus N --steps K --warmup W       (N > 1: one rank per GPU -- started by torch.distributed.run, or by bench.py itself when RANK is not set)

A step = one pass of the hot path (amp_model_infer: preprocess -> backbone -> FPN -> RPN -> proposals -> box head ->
detections -> mask head -> paste -> RLE counts strings on the host) over one batch of 8 images that is already resident in HBM.
Image-parallel replicas: every rank runs the same per-GPU batch, no data-path collective ("weak" scaling); RCCL (through the C ABI) is only
used for the barrier and the max-over-ranks of the elapsed time.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-power-probe", action="store_true", help="skip roofline.power_probe (the dominant layer on random and on all-zero operands)")
    ap.add_argument("--cpu-images", type=int, default=10)
    ap.add_argument("--train-steps", type=int, default=50, help="timed training steps of the secondary `train` object (0 = skip)")
    ap.add_argument("--train-warmup", type=int, default=10)
    ap.add_argument("--x101-steps", type=int, default=5, help="timed steps of the secondary `x101_2048` object (0 = skip)")
    ap.add_argument("--no-strict", action="store_true", help="skip the fp32-MFMA reference run of the headline workload")
    ap.add_argument("--no-two-pipelines", action="store_true", help="skip the secondary `two_pipelines` object")
    ap.add_argument("--no-host-inclusive", action="store_true", help="skip the secondary `host_inclusive` object")
    ap.add_argument("--train-timeout", type=float, default=420.0, help="watchdog for the secondary legs, seconds")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` from a plain shell (no RANK in the environment): start the N ranks ourselves, one process per
    GPU, through torch.distributed.run -- BEFORE this process has made any GPU call (a process that has initialised HIP must not
    be replaced or forked) -- pass their output through and exit with their code."""
    import socket
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print(f"[bench] --gpus {args.gpus} without RANK in the environment: launching {' '.join(cmd[1:8])} ...", file=sys.stderr, flush=True)
    sys.exit(subprocess.call(cmd, env=env))


ARGS = None
if __name__ == "__main__":
    ARGS = parse_args()
    if ARGS.gpus > 1 and "RANK" not in os.environ:
        self_launch(ARGS)
    if os.environ.get("AMP_BENCH_DRY_LAUNCH"):     # tests/test_comm_cpu.py: prove the self-launch on a host without GPUs
        print(f"dry-launch rank {os.environ.get('RANK', '0')} of {os.environ.get('WORLD_SIZE', '1')} local {os.environ.get('LOCAL_RANK', '0')}", flush=True)
        sys.exit(0)

import numpy as np
import torch

from ampis_amd import _lib, params as P, synth
from ampis_amd.model import MaskRCNN, RLE_STRINGS

BATCH, SIZE, K, DETS = 8, 1024, 2, 200
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 matrix peak (v_mfma_f32_32x32x16_f16)
# AMP_CONV_F16X3 (the default conv arithmetic, include/ampis_hip.h): fp32 in / fp32 out, every product a*b evaluated as three
# exact-product f16 MFMAs with fp32 accumulation -> its ceiling in ALGORITHMIC flops (2*M*N*K) is a third of the f16 matrix peak.
PEAK_F16X3_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0
# What the f16 matrix pipe SUSTAINS on this chip when its operands toggle like a GEMM's (tools/mfma_peak.py, profiles/r01/mfma_peak.json:
# register-only back-to-back v_mfma_f32_32x32x16_f16 on random operands 1655 TFLOP/s, on constant operands 2475; the fp32 MFMA holds
# 155 either way).  Reported beside `peak`, never instead of it.
SUSTAINED_F16_MFMA_RANDOM_TFLOPS = 1655.0
PROF_EVERY = 10                # timed launches (events attached to the dispatch) on every 10th timed step: an attached event still costs ~8 us of idle GPU per
                               # launch (0.8 ms on a sampled inference step, rocprofv3 trace), i.e. 0.08 ms per step on average at this cadence; 5 of the
                               # default 50 steps = 270 launches of the dominant kernel


PEAK_HBM_TBS = 8.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def by_bound(launches, mode, steps):
    """Per-LAUNCH roofline of the conv launches of the sampled steps (amp_prof_launches: HIP-event duration, algorithmic flops and algorithmic
    bytes of each): a launch is bounded by whichever of flops / matrix peak and bytes / HBM peak is the longer time (the short-K 1x1 layers of
    res2-res4 and the FPN laterals move 2-4 x more bytes than the matrix pipe needs time for).  Returns, per bound, the launches, their measured
    time, the achieved rate and frac = sum of the bound's ideal times / sum of the measured times; `all` = the same over every launch."""
    peak = (PEAK_F16X3_TFLOPS if mode == "f16x3" else PEAK_F32_MFMA_TFLOPS) * 1e12
    g = {"mfma": [0, 0.0, 0.0, 0.0, 0.0], "hbm": [0, 0.0, 0.0, 0.0, 0.0]}      # launches, ms, ideal ms, flops, bytes
    for r in launches:
        if r["slot"] == 2 or r["ms"] <= 0:
            continue
        t_m, t_h = r["flops"] / peak * 1e3, r["bytes"] / (PEAK_HBM_TBS * 1e12) * 1e3
        k = "mfma" if t_m >= t_h else "hbm"
        g[k][0] += 1; g[k][1] += r["ms"]; g[k][2] += max(t_m, t_h); g[k][3] += r["flops"]; g[k][4] += r["bytes"]
    out = {}
    for k, (n, ms, ideal, fl, by) in g.items():
        if n == 0:
            continue
        out[k] = {"launches_per_step": round(n / max(steps, 1), 1), "kernel_ms_per_step": round(ms / max(steps, 1), 3),
                  "achieved": round(fl / (ms * 1e-3) / 1e12, 2) if k == "mfma" else round(by / (ms * 1e-3) / 1e12, 3),
                  "peak": round(peak / 1e12, 1) if k == "mfma" else PEAK_HBM_TBS, "unit": "TFLOP/s" if k == "mfma" else "TB/s (algorithmic bytes)",
                  "frac": round(ideal / ms, 4)}
    tot_ms = sum(v[1] for v in g.values())
    if tot_ms > 0:
        out["all"] = {"frac": round(sum(v[2] for v in g.values()) / tot_ms, 4),
                      "what": "sum over launches of max(flops / matrix peak, algorithmic bytes / HBM peak) / sum of measured durations"}
    return out


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cores():
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU box's 16-per-GPU share."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(n_images):
    """The torch-CPU oracle ("port": the reference's detectron2 path cannot run here) on a bounded sample of
    the same workload: n_images single 1024x1024 micrographs, same weights and cfg."""
    from oracle import maskrcnn as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline: oracle on {n_images} image(s) with {cores} threads")
    imgs, _ = synth.batch(n_images, SIZE, SIZE)
    p = O.to_torch_params(P.init_params(K, seed=0, style="spread"))
    cfg = O.Cfg(num_classes=K, detections_per_image=DETS)
    O.infer(imgs[:1, :256, :256], p, cfg)   # warm the thread pool
    t = time.perf_counter()
    for i in range(n_images):
        O.infer(imgs[i:i + 1], p, cfg)
        log(f"cpu_baseline: image {i + 1}/{n_images} done at {time.perf_counter() - t:.1f} s")
    dt = time.perf_counter() - t
    return {"value": n_images / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n_images} x 1024x1024 synthetic micrographs, batch 1, torch-CPU oracle (oracle/maskrcnn.py), "
                      f"{DETS} detections/image, {dt:.1f} s"}


TRAIN_BATCH = 16
PROFILE_DIRS = ("r03", "r02", "r01")      # committed rocprofv3 --pmc summaries (tools/profile_round.sh): newest first


def pmc_traffic(kernel_keys, leg="infer"):
    """HBM bytes per launch of the first of `kernel_keys` found in the committed PMC summaries of this leg's own command
    (profiles/rNN/pmc_summary_vK.json: the inference bench; pmc_summary_train_vK.json: the training leg), newest version first."""
    import glob
    import re

    def version(f):
        m = re.search(r"_v(\d+)\.json$", f)
        return int(m.group(1)) if m else -1

    for d in PROFILE_DIRS:
        files = [f for f in glob.glob(os.path.join(ROOT, "profiles", d, "pmc_summary*.json")) if ("_train_" in os.path.basename(f)) == (leg == "train")]
        for f in sorted(files, key=version, reverse=True):
            try:
                k = json.load(open(f))["kernels"]
            except Exception:
                continue
            for key in kernel_keys:
                if key in k and k[key].get("hbm_bytes_per_launch"):
                    return k[key]["hbm_bytes_per_launch"], os.path.relpath(f, ROOT), key
    return None, None, None


def rocprof_reported(leg="infer"):
    """north_star's "rocprof-reported HBM GB/s and MFMA utilisation for the backbone conv stack and RoIAlign": the committed PMC
    summary of this leg's own command (tools/profile_round.sh + tools/summarize_profiles.py), quoted -- not measured in this run."""
    import glob
    import re
    for d in PROFILE_DIRS:
        files = [f for f in glob.glob(os.path.join(ROOT, "profiles", d, "pmc_summary*.json")) if ("_train_" in os.path.basename(f)) == (leg == "train")]
        files.sort(key=lambda f: int((re.search(r"_v(\d+)\.json$", f) or [0, -1])[1]), reverse=True)
        for f in files:
            try:
                k = json.load(open(f))["kernels"]
            except Exception:
                continue
            pick = {}
            for name, v in k.items():
                if name.startswith(("conv_split_kernel", "conv_glds_kernel", "conv_f16x3_kernel", "roi_align", "wgrad_split_kernel", "wgrad_f16x3_kernel")):
                    pick[name] = {"avg_us": v.get("avg_duration_us"), "launches": v.get("launches_in_stats_run"), "hbm_GBps": v.get("hbm_GBps"),
                                  "hbm_bytes_per_launch": v.get("hbm_bytes_per_launch"), "mfma_util": v.get("mfma_util")}
            if pick:
                return {"source": os.path.relpath(f, ROOT), "what": "rocprofv3 --kernel-trace --stats + separate --pmc passes (FETCH_SIZE, WRITE_SIZE, "
                        "SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) of this command on one MI355X; hbm = L2 <-> fabric bytes", "kernels": pick}
    return None


class Ranks:
    """Barrier and max-over-ranks for the timed regions.  N > 1: the device collectives are the library's own RCCL calls
    (amp_barrier / amp_allreduce on the context's communicator); torch.distributed (gloo) only carried the RCCL id.
    AMP_BENCH_REHEARSAL=1 (two ranks sharing the card of a one-GPU box, where RCCL refuses to run): gloo on host values."""

    def __init__(self, ctx, dev, world, staged):
        self.ctx, self.dev, self.world, self.staged = ctx, dev, world, staged
        self._d = ctx.malloc(8) if (world > 1 and not staged) else None

    def barrier(self):
        if self.world > 1:
            if self.staged:
                torch.distributed.barrier()
            else:
                self.ctx.barrier()
        torch.cuda.synchronize(self.dev)

    def max(self, value):
        if self.world == 1:
            return value
        if self.staged:
            t = torch.tensor([value], dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())
        a = np.array([value], dtype=np.float64)
        self.ctx.h2d(self._d, a)
        self.ctx.allreduce(self._d, 1, self.ctx.F64, self.ctx.MAX)
        self.ctx.sync()
        self.ctx.d2h(a, self._d)
        return float(a[0])


def train_leg(ctx, infer_model, dev, rank, world, warmup, steps, ranks):
    """images/s of a full training step; whole-job aggregate over the ranks (weak scaling, global batch 16 x N).  The gradient
    exchange is the library's bucketed RCCL all-reduce, issued from inside amp_model_forward_backward as the backward pass
    completes each bucket."""
    from ampis_amd.utils import comm
    infer_model.close()                       # give its workspace back before the 40 GiB training workspace
    log(f"rank {rank}: creating the training model (local batch {TRAIN_BATCH})")
    model = MaskRCNN(ctx, K, max_batch=TRAIN_BATCH, max_h=SIZE, max_w=SIZE, max_out_hw=SIZE, train=True,
                     max_gt=TRAIN_BATCH * 800, max_poly_doubles=TRAIN_BATCH * 800 * 64)
    params = P.init_params(K, seed=0, style="spread")
    model.load_params(params)
    imgs, gts = synth.batch(TRAIN_BATCH, SIZE, SIZE, first_index=1000 + rank * TRAIN_BATCH)
    # inputs resident in HBM before the timed region (as for the inference legs); annotations flattened once, as a loader worker would
    from ampis_amd.model import PackedGt
    d_imgs = ctx.malloc(imgs.nbytes)
    ctx.h2d(d_imgs, imgs)
    packed = PackedGt(gts)
    if world == 1 and ctx.comm_info()[1] == 0 and not ranks.staged:
        # one rank: still bring RCCL up (a communicator of size 1), so that the N=1 line measures the step WITH the exchange
        # machinery the N>1 runs use (bucket events, communication stream, the wait in front of SGD)
        try:
            comm.attach_rccl(ctx)
        except Exception as e:   # noqa: BLE001
            log(f"rank {rank}: RCCL communicator of size 1 not available ({e}); training step timed without the exchange machinery")
    has_comm = ctx.comm_info()[1] > 0
    if has_comm:
        model.broadcast_params(0)             # rank 0's parameters and momentum to every rank (DDP's constructor broadcast; a no-op exchange at 1 rank)

    def step(i):
        losses = model.forward_losses(None, packed, seed=i, backward=True, device_ptr=d_imgs, shape=(TRAIN_BATCH, SIZE, SIZE))
        scale = comm.all_reduce_gradients(model, ctx)
        model.sgd_step(1e-3, 0.9, 1e-4, grad_scale=scale)
        return losses

    for i in range(warmup):
        step(i)
    ranks.barrier()
    ctx.prof_begin(max_launches=(steps // PROF_EVERY + 1) * 512)
    exposed, span, nstat = 0.0, 0.0, 0
    bucket_us = None
    t0 = time.perf_counter()
    prof_steps = 0
    for i in range(steps):
        sampled = i % PROF_EVERY == 0
        ctx.prof_pause(not sampled)
        prof_steps += int(sampled)
        losses = step(1000 + i)
        if has_comm and sampled:      # event times of this step's exchange (waits for the step: only on the sampled steps)
            st = ctx.comm_stats()
            exposed += st["exposed_ms"]; span += st["span_ms"]; nstat += 1
            bu = ctx.comm_bucket_stats()
            bucket_us = bu if bucket_us is None else [a + b for a, b in zip(bucket_us, bu)]
    ctx.sync()
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    prof = ctx.prof_end()
    ranks.barrier()
    el = ranks.max(el)
    buckets = model.grad_buckets()
    model.close()
    if world > 1 or has_comm:     # per rank: what its gradient exchange cost (the JSON line carries rank 0's)
        log(f"rank {rank}: grad exchange over {ctx.comm_info()[1] if has_comm else 0} RCCL ranks (version {ctx.comm_info()[2] if has_comm else None}): "
            f"exposed_comm_ms {exposed / nstat if nstat else None}, comm_span_ms {span / nstat if nstat else None}, bucket_us "
            f"{[round(v / nstat, 1) for v in bucket_us] if (nstat and bucket_us) else None} (mask head, box head, RPN, FPN, res5, res4, res3); "
            f"step {el / steps * 1e3:.2f} ms (max over ranks)")
    f16 = ctx.conv_mode == ctx.CONV_F16X3
    peak = PEAK_F16X3_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
    wg_ms, wg_fl, wg_n = prof["ms"][2], prof["flops"][2], prof["launches"][2]
    ach = wg_fl / (wg_ms * 1e-3) / 1e12 if wg_ms > 0 else 0.0
    kname = "wgrad_split_kernel + wgrad_f16x3_kernel" if f16 else "wgrad_mfma_kernel"
    traffic, traffic_src, traffic_key = pmc_traffic(["wgrad_split_kernel", "wgrad_f16x3_kernel"] if f16 else [kname], leg="train")
    conv_ms = prof["ms"][0] + prof["ms"][1]
    conv_tf = (prof["flops"][0] + prof["flops"][1]) / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    conv_traffic, conv_traffic_src, conv_traffic_key = pmc_traffic(["conv_split_kernel<128x256>"] if f16 else ["conv_glds_kernel<128>"], leg="train")
    out = {"metric": "images/sec Mask R-CNN R50-FPN @1024x1024 training (fwd + losses + bwd + all-reduce + SGD)",
           "value": round(world * TRAIN_BATCH * steps / el, 3), "unit": "images/s", "ms_per_step": round(el / steps * 1e3, 2),
           "steps": steps, "warmup": warmup, "batch_per_gpu": TRAIN_BATCH, "global_batch": TRAIN_BATCH * world,
           "dtype": "f32 (forward, data-gradient and weight-gradient convs: f16x3 split-operand MFMA, fp32 accumulate)" if f16 else "f32",
           "workload": "BASELINE configs[2] (N=1) / configs[3] (N=8): K=2, ~480 GT instances/image (polygons), 256 anchors + 512 RoIs "
                       "sampled per image, seeded random-init weights, uint8 images resident in HBM, annotations (boxes, classes, polygons) passed from the host each step; "
                       "every gradient of every trainable tensor is produced -- the RPN head's four gradient GEMMs run over the <= 256 sampled anchors' pixels per image, "
                       "the only rows of its loss gradient that are not exact zeros (csrc/rpn_sparse.hip; AMP_NO_RPN_SPARSE=1: over every pixel, same gradients to 6e-7)",
           # the dominant kernel of a training step is the forward / data-gradient convolution (conv_split_kernel<128x256>: ~46 of 81 ms);
           # the weight-gradient kernels (~23 ms) are reported beside it
           "roofline": {"bound": "mfma",
                        "kernel": ("forward + data-gradient convolutions: conv_split_kernel<128x256> (dominant) with conv_split_kernel<256x128>, conv_glds_kernel<*,F16> "
                                   "and conv_f16x3_kernel on the layers those do not take" if f16 else "forward + data-gradient convolutions: conv_glds_kernel<128>"),
                        "achieved": round(conv_tf, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(conv_tf / peak, 4),
                        **({"frac_of_sustained": round(conv_tf / (SUSTAINED_F16_MFMA_RANDOM_TFLOPS / 3.0), 4)} if f16 else {}),
                        "traffic": conv_traffic, "traffic_from": (f"{conv_traffic_src}: {conv_traffic_key}" if conv_traffic_src else None),
                        "launches_per_step": round((prof["launches"][0] + prof["launches"][1]) / max(prof_steps, 1), 1),
                        "kernel_ms_per_step": round(conv_ms / max(prof_steps, 1), 3),
                        "wgrad": {"kernel": (f"{kname} (dW = dY^T X; both operands in the split row format: 128x256 tiles, LDS-DMA ring + transposing LDS reads; fp32 dY: "
                                             "128x128 tiles split in registers; split-K slabs reduced in fixed order)" if f16 else
                                             f"{kname} (dW = dY^T X, 128x128 tiles, split-K slabs reduced in fixed order)"),
                                  "achieved": round(ach, 2), "frac": round(ach / peak, 4),
                                  **({"frac_of_sustained": round(ach / (SUSTAINED_F16_MFMA_RANDOM_TFLOPS / 3.0), 4)} if f16 else {}),
                                  "traffic": traffic, "traffic_from": (f"{traffic_src}: {traffic_key}" if traffic_src else None),
                                  "launches_per_step": round(wg_n / max(prof_steps, 1), 1), "kernel_ms_per_step": round(wg_ms / max(prof_steps, 1), 3)},
                        "events_on": f"every {PROF_EVERY}th timed step ({prof_steps} of {steps})", "truncated": prof["truncated"]},
           "grad_exchange": {"backend": "rccl (amp_comm_*, issued inside amp_model_forward_backward)" if has_comm else
                                        ("staged (rehearsal: gloo on host buffers)" if world > 1 else "none (1 rank, no communicator)"),
                             "rccl_ranks": ctx.comm_info()[1] if has_comm else 0,
                             "rccl_version": ctx.comm_info()[2] if has_comm else None,
                             "buckets": len({b for b, _, _ in buckets}), "ranges": len(buckets),
                             "MB_per_step": round(sum(n for _, _, n in buckets) * 4 / 1e6, 1),
                             "exposed_comm_ms": round(exposed / nstat, 3) if nstat else None,
                             "comm_span_ms": round(span / nstat, 3) if nstat else None,
                             # mean microseconds of each bucket's grouped all-reduce on the communication stream, in issue order
                             # (mask head, box head, RPN, FPN, res5, res4, res3): with N > 1 ranks this is the per-bucket xGMI time
                             "bucket_us": [round(v / nstat, 1) for v in bucket_us] if (nstat and bucket_us) else None},
           "last_losses": {k: round(v, 4) for k, v in losses.items()}}
    return out


def two_pipelines_leg(local_rank, dev, rank, world, steps, ranks, params, imgs):
    """The headline workload with TWO batches in flight per GPU through the library's own pipeline object (include/ampis_hip.h
    amp_pipeline: one handle, ONE calling thread, a worker thread and a context per lane inside the library).  A single synchronous
    call leaves the chip idle while the host reads the detection counts and the results, in the latency-bound selection / NMS / paste
    kernels and in the ramp and tail of every launch; the next batch's convolutions fill those.  Same kernels, same batch of 8 per
    step, bit-identical results (tests/test_pipeline_gpu.py) -- only consecutive batches overlap, which is how a serving process runs.
    (Splitting ONE call into micro-batches on several streams does not help: tools/exp_halfbatch.py)"""
    from ampis_amd.model import InferPipeline
    depth = int(os.environ.get("AMP_BENCH_PIPE_DEPTH", "2"))      # (3 in flight: measured, no better than 2 -- DESIGN §6)
    pipe = InferPipeline(local_rank, K, depth=depth, max_batch=BATCH, max_h=SIZE, max_w=SIZE, max_out_hw=SIZE, detections_per_image=DETS)
    pipe.load_params(params)
    c = _lib.Context(local_rank)
    d = c.malloc(imgs.nbytes)
    c.h2d(d, imgs)
    c.sync()

    def run(n):
        pend, nd = [], 0
        for _ in range(n):
            if len(pend) == depth:
                r = pipe.wait_raw(pend.pop(0))
                nd += sum(r.n[b] for b in range(BATCH))
            pend.append(pipe.submit(device_ptr=d, shape=(BATCH, SIZE, SIZE)))
        while pend:
            r = pipe.wait_raw(pend.pop(0))
            nd += sum(r.n[b] for b in range(BATCH))
        return nd

    try:
        run(4)
        n = max(2, steps)
        ranks.barrier()
        t0 = time.perf_counter()
        nd = run(n)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        ranks.barrier()
        el = ranks.max(el)
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"[:200]}
    finally:
        pipe.close()
        c.free(d)
        c.close()
    return {"what": f"same workload, {depth} batches of 8 in flight per GPU: amp_pipeline depth {depth} (one handle, one calling thread; every batch's results on the host inside the timed region)",
            "value": round(world * BATCH * n / el, 3), "unit": "images/s", "steps": n, "pipeline_depth": depth,
            "ms_per_step": round(el / n * 1e3, 3), "detections_per_image_mean": round(nd / (n * BATCH), 2)}


def host_inclusive_leg(model, imgs, dev, world, steps, ranks):
    """The headline workload as the reference's `predictor(img)` sees it: the uint8 batch starts on the HOST (24 MB over PCIe per
    step) and the masks come back as COCO compressed-RLE byte strings (what compress_pred stores, ampis/data_utils.py:275).  Never
    `value`: the contract times inputs resident in HBM."""
    model.infer(imgs)
    ranks.barrier()
    t0 = time.perf_counter()
    nm = 0
    for _ in range(steps):
        out = model.infer(imgs)
        nm += sum(len(o["masks"]) for o in out)
    torch.cuda.synchronize(dev)
    el = ranks.max(time.perf_counter() - t0)
    ranks.barrier()
    return {"what": "same workload, timer also around H2D of the uint8 batch (pageable host memory) and the encoding of every mask's run lengths "
                    "to COCO RLE strings + the per-image result dicts",
            "value": round(world * BATCH * steps / el, 3), "unit": "images/s", "steps": steps, "ms_per_step": round(el / steps * 1e3, 3),
            "rle_strings_per_step": round(nm / steps, 1)}


def power_probe_leg(ctx, dev):
    """The dominant kernel on ONE of its layers (the FPN output conv at p2: B = 8, 256 x 256 pixels, 3x3 256 -> 256) with random operands and with
    all-zero operands: the same instruction stream, the same bytes, without the data-dependent switching energy.  The gap between the two is what
    the chip's power management takes (it holds ~1.8 of 2.4 GHz under this loop on random data, tools/stamp_conv.py); what is left to 833 TFLOP/s in the
    zero-operand run is the loop itself.  A diagnostic beside `roofline`, never part of `value`."""
    from ampis_amd import ops
    Bp, Hp, Wp, Cp = BATCH, SIZE // 4, SIZE // 4, 256
    flops = 2.0 * Bp * Hp * Wp * Cp * 9 * Cp
    out = {"layer": f"3x3 {Cp}->{Cp}, stride 1, B={Bp} {Hp}x{Wp} (FPN output conv at p2), split operands in and out", "launches": 20}
    sc, sh = torch.ones(Cp, device=dev), torch.zeros(Cp, device=dev)
    for name in ("random", "zero"):
        x = torch.randn(Bp, Hp, Wp, Cp, device=dev) if name == "random" else torch.zeros(Bp, Hp, Wp, Cp, device=dev)
        w = torch.randn(Cp, 3, 3, Cp, device=dev) * 0.05 if name == "random" else torch.zeros(Cp, 3, 3, Cp, device=dev)
        xs = ops.split_rows(ctx, x)
        del x
        run = lambda n: [ops.conv2d_nhwc(ctx, xs, w, sc, sh, stride=1, pad=1, relu=True, fmt=ops.FMT_X_SPLIT | ops.FMT_Y_SPLIT) for _ in range(n)]
        run(10)
        torch.cuda.synchronize(dev)
        ctx.timer_start()
        run(20)
        ms = ctx.timer_stop() / 20
        out[f"{name}_operands_us"] = round(ms * 1e3, 1)
        out[f"{name}_operands_tflops"] = round(flops / ms / 1e9, 1)
        del xs, w
    out["zero_over_random"] = round(out["zero_operands_tflops"] / out["random_operands_tflops"], 3)
    out["frac_of_peak_zero_operands"] = round(out["zero_operands_tflops"] / PEAK_F16X3_TFLOPS, 3)
    torch.cuda.empty_cache()
    return out


def x101_leg(ctx, dev, rank, world, steps, ranks):
    """BASELINE configs[4] per GPU: X-101-32x8d-FPN inference on native 2048x2048 synthetic micrographs, 500 detections/image."""
    S, D, XB = 2048, 500, 2
    log(f"rank {rank}: creating the X-101-32x8d model (batch {XB} x {S}x{S})")
    model = MaskRCNN(ctx, K, max_batch=XB, max_h=S, max_w=S, max_out_hw=S, detections_per_image=D,
                     pixel_std=(57.375, 57.120, 58.395), arch="X101")
    p = P.init_params(K, seed=0, style="spread", arch="X101")
    p["backbone.bottom_up.stem.conv1.weight"] = p["backbone.bottom_up.stem.conv1.weight"] * np.float32(57.0)   # seeded stem assumes unit std
    model.load_params(p)
    model.set_rle_output(RLE_STRINGS)
    del p
    imgs, _ = synth.batch(XB, S, S, first_index=2000 + rank * XB)
    d_imgs = ctx.malloc(imgs.nbytes)
    ctx.h2d(d_imgs, imgs)
    for _ in range(2):
        model.infer_raw(None, device_ptr=d_imgs, shape=(XB, S, S))
    ranks.barrier()
    ctx.prof_begin(max_launches=steps * 256)
    t0 = time.perf_counter()
    ndet = 0
    for _ in range(steps):
        d = model.infer_raw(None, device_ptr=d_imgs, shape=(XB, S, S))
        ndet += sum(d.n[b] for b in range(XB))
    torch.cuda.synchronize(dev)
    el = time.perf_counter() - t0
    prof = ctx.prof_end()
    ranks.barrier()
    el = ranks.max(el)
    model.close()
    conv_ms = prof["ms"][0] + prof["ms"][1]
    f16 = ctx.conv_mode == ctx.CONV_F16X3
    return {"metric": "images/sec Mask R-CNN X-101-32x8d-FPN @2048x2048 inference", "value": round(world * XB * steps / el, 3),
            "unit": "images/s", "ms_per_step": round(el / steps * 1e3, 2), "steps": steps, "batch_per_gpu": XB,
            "dtype": "f32 (convs: f16x3 split-operand MFMA, fp32 accumulate)" if f16 else "f32",
            "workload": "BASELINE configs[4]: native 2048x2048 (no tiling), K=2, 1000 proposals/img, TEST.DETECTIONS_PER_IMAGE=500, "
                        "seeded random-init weights, grouped 3x3 convs as 64-wide block-diagonal MFMA tiles",
            "detections_per_image_mean": round(ndet / (steps * XB), 1),
            "conv_useful_tflops": round((prof["flops"][0] + prof["flops"][1]) / (conv_ms * 1e-3) / 1e12, 2) if conv_ms > 0 else None}


def main(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device: there is no CPU fallback for the hot path"
    staged = bool(os.environ.get("AMP_BENCH_REHEARSAL"))   # rehearsal of the N>1 code path on a one-GPU box: every rank on device 0, gloo on host buffers
    if staged:
        local_rank = 0
        os.environ["AMP_COMM_BACKEND"] = "staged"
    elif os.environ.get("AMP_BENCH_ONE_CARD"):
        # second rehearsal mode: every rank on device 0 with the library's OWN exchange path (amp_comm_*: barrier, max of the elapsed times,
        # parameter broadcast, the seven overlapped gradient buckets) -- possible with AMP_RCCL_LIB pointing at the shared-memory stand-in
        # for librccl (tests/fake_rccl: RCCL itself refuses two ranks on one device).  Timings of such a run mean nothing; that it runs does.
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # torch.distributed on gloo is the side channel only: it carries the RCCL id to the ranks; every device collective below
        # (barrier, max of the elapsed times, the gradient all-reduce) is an RCCL call made by libampis_hip.so
        torch.distributed.init_process_group("gloo")

    log(f"rank {rank}/{world}: creating model")
    ctx = _lib.Context(local_rank)
    rccl = None
    rccl_error = None
    if world > 1 and not staged:
        from ampis_amd.utils import comm
        try:
            rccl = comm.attach_rccl(ctx)
        except Exception as e:   # noqa: BLE001
            rccl_error = f"{type(e).__name__}: {e}"[:300]
        # the ranks agree over the side channel: one rank without a communicator puts every rank's timing collectives (barrier, max of the
        # elapsed times -- the inference path has no data-path collective) on gloo, loudly; the training leg, whose gradient exchange IS RCCL,
        # then reports an error object instead of a number and the line carries "rccl_error"
        flag = torch.tensor([1 if rccl_error else 0])
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if int(flag.item()):
            if rccl is not None:
                comm.detach_rccl()
                rccl = None
            rccl_error = rccl_error or "another rank could not create its RCCL communicator"
            log(f"rank {rank}: RCCL communicator NOT created ({rccl_error}); barrier / max-over-ranks fall back to gloo, the training leg is skipped")
    if rccl is not None:
        # every rank says what it got, so that the first multi-GPU run explains itself from its log alone
        log(f"rank {rank}: RCCL communicator up (rank {rccl[0]} of {rccl[1]}, version {rccl[2]}, device {local_rank}"
            + (f", AMP_RCCL_LIB={os.environ['AMP_RCCL_LIB']} OVERRIDES librccl" if os.environ.get("AMP_RCCL_LIB") else "") + ")")
        if rccl[1] != world or rccl[0] != rank:
            log(f"rank {rank}: FATAL: the communicator has {rccl[1]} ranks (this is rank {rccl[0]}) but the job has {world} (this is rank {rank})")
            sys.exit(4)
    ranks = Ranks(ctx, dev, world, staged or rccl_error is not None)
    model = MaskRCNN(ctx, K, max_batch=BATCH, max_h=SIZE, max_w=SIZE, max_out_hw=SIZE, detections_per_image=DETS)
    params = P.init_params(K, seed=0, style="spread")
    model.load_params(params)
    model.set_rle_output(RLE_STRINGS)      # masks come back as COCO counts strings, encoded on the device (what compress_pred stores)
    log(f"workspace {model.workspace_bytes / 2**30:.2f} GiB; weights loaded; generating {BATCH} micrographs")
    imgs, _ = synth.batch(BATCH, SIZE, SIZE, first_index=rank * BATCH)
    d_imgs = ctx.malloc(imgs.nbytes)
    ctx.h2d(d_imgs, imgs)

    def step():
        return model.infer_raw(None, device_ptr=d_imgs, shape=(BATCH, SIZE, SIZE))

    def timed(mode, warmup, steps):
        """W untimed + exactly K timed steps in one conv arithmetic, bracketed by barrier + synchronize; max over ranks."""
        ctx.conv_mode = mode
        for i in range(warmup):
            t = time.perf_counter()
            step()
            log(f"[{mode}] warmup step {i}: {(time.perf_counter() - t) * 1e3:.1f} ms")
        ranks.barrier()
        # per-launch HIP events on every PROF_EVERY-th timed step: they cost ~8 us of idle GPU per launch (0.8 ms
        # = 4.7 % of a step when every step carries them)
        ctx.prof_begin(max_launches=(steps // PROF_EVERY + 1) * 96)
        t0 = time.perf_counter()
        ndet = 0
        prof_steps = 0
        for i in range(steps):
            sampled = i % PROF_EVERY == 0
            ctx.prof_pause(not sampled)
            prof_steps += int(sampled)
            d = step()
            ndet += sum(d.n[b] for b in range(BATCH))
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        prof = ctx.prof_end()
        prof["steps"] = prof_steps
        prof["by_bound"] = by_bound(ctx.prof_launches(), mode, prof_steps)
        log(f"[{mode}] timed {steps} steps in {el:.3f} s")
        ranks.barrier()
        return ranks.max(el), prof, ndet

    mode = "f32" if os.environ.get("AMP_CONV_MODE") == "f32" else "f16x3"
    elapsed, prof, ndet = timed(mode, args.warmup, args.steps)
    strict = None
    if mode == "f16x3" and not args.no_strict:      # the same workload on the fp32-MFMA kernels, for reference
        s_el, s_prof, _ = timed("f32", 2, args.steps)
        ctx.conv_mode = "f16x3"
        s_ach = s_prof["flops"][0] / (s_prof["ms"][0] * 1e-3) / 1e12 if s_prof["ms"][0] > 0 else 0.0
        strict = {"what": "same workload with AMP_CONV_MODE=f32: every conv on v_mfma_f32_32x32x2_f32",
                  "value": round(world * BATCH * args.steps / s_el, 3), "unit": "images/s", "ms_per_step": round(s_el / args.steps * 1e3, 3),
                  "roofline": {"bound": "mfma", "kernel": "conv_glds_kernel<128>", "achieved": round(s_ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(s_ach / PEAK_F32_MFMA_TFLOPS, 4),
                               "kernel_ms_per_step": round(s_prof["ms"][0] / s_prof["steps"], 3)}}

    out = None
    if rank == 0:
        keys = ["conv_split_kernel<128x256>", "conv_glds_kernel<256>[f16x3, both operands pre-split]", "conv_f16x3_kernel<256>"] if mode == "f16x3" else ["conv_glds_kernel<128>"]
        traffic, traffic_src, traffic_key = pmc_traffic(keys)   # HBM bytes per launch of the dominant kernel, from the committed PMC passes (profiles/)
        ms_per_step = elapsed / args.steps * 1e3
        value = world * BATCH * args.steps / elapsed
        ach = prof["flops"][0] / (prof["ms"][0] * 1e-3) / 1e12 if prof["ms"][0] > 0 else 0.0
        peak = PEAK_F16X3_TFLOPS if mode == "f16x3" else PEAK_F32_MFMA_TFLOPS
        conv_all = (prof["flops"][0] + prof["flops"][1]) / ((prof["ms"][0] + prof["ms"][1]) * 1e-3) / 1e12
        all_conv_ms = (prof["ms"][0] + prof["ms"][1]) / prof["steps"]
        out = {
            "metric": "images/sec Mask R-CNN R50-FPN @1024x1024 inference", "value": round(value, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if mode == "f32" else "f32 (convs: f16x3 split-operand MFMA -- 22-bit operands, exact products, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: R50-FPN inference, batch=8 synthetic 1024x1024 micrographs per GPU, "
                                   "K=2, 1000 proposals/img, TEST.DETECTIONS_PER_IMAGE=200, seeded random-init weights, "
                                   "outputs on the host: boxes, scores, classes and every mask as its COCO compressed-RLE counts string (encoded on the device)",
                       "batch_per_gpu": BATCH, "global_batch": BATCH * world, "image_size": [SIZE, SIZE],
                       "detections_per_image_mean": round(ndet / (args.steps * BATCH), 2),
                       "parallelism": f"image-parallel replicas x{world}, no data-path collective"
                                      + (f"; barrier / max-over-ranks on RCCL {rccl[2]} through the C ABI" if rccl else "")
                                      + (f"; RCCL COMMUNICATOR NOT CREATED ({rccl_error}): barrier / max-over-ranks on gloo" if rccl_error else "")},
            "roofline": {"bound": "mfma",
                         "kernel": ("conv_split_kernel<128x256> -- every launch of it and nothing else (tagged where it is launched: the 3x3 256->256 layers "
                                    "of FPN / RPN / mask head / res4, fc1, the fused RPN tail; the fused mask-head tail is mask_tail_kernel since round 4 and not counted here): implicit-GEMM conv on "
                                    "v_mfma_f32_16x16x32_f16, 3 MFMAs per product, both operands pre-split and staged by LDS-DMA through a ring of "
                                    "three 48-KB tiles, the two waves of a SIMD ping-pong between loading and multiplying") if mode == "f16x3" else
                                   "conv_glds_kernel<128> (fp32 MFMA implicit-GEMM conv, 128x128x32 tiles, LDS-DMA staging)",
                         "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "peak_is": ("f16 dense MFMA peak 2500 / 3 MFMAs per algorithmic product" if mode == "f16x3" else "fp32 dense MFMA peak"),
                         "achieved_over_fp32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 3),
                         **({"sustained_peak_random_operands": round(SUSTAINED_F16_MFMA_RANDOM_TFLOPS / 3.0, 1),
                             "frac_of_sustained": round(ach / (SUSTAINED_F16_MFMA_RANDOM_TFLOPS / 3.0), 4),
                             "sustained_is": "measured on this chip: MFMA-only loop on random f16 operands 1655 TFLOP/s (clock drops under "
                                             "toggling inputs; 2475 on constants), / 3 (profiles/r01/mfma_peak.json)"} if mode == "f16x3" else {}),
                         "traffic": traffic,
                         "traffic_unit": f"HBM bytes per launch of {traffic_key} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, {traffic_src})",
                         "launches_per_step": prof["launches"][0] / prof["steps"],
                         "kernel_ms_per_step": round(prof["ms"][0] / prof["steps"], 3),
                         "all_conv_ms_per_step": round(all_conv_ms, 3),
                         "non_conv_ms_per_step": round(ms_per_step - all_conv_ms, 3),
                         "events_on": f"every {PROF_EVERY}th timed step ({prof['steps']} of {args.steps} steps, {prof['launches'][0]} launches of the dominant kernel)",
                         "all_conv_tflops": round(conv_all, 2), "by_bound": prof["by_bound"], "truncated": prof["truncated"]},
        }
        rp = rocprof_reported("infer") if mode == "f16x3" else None
        if rp:
            out["rocprof_reported"] = rp

    emitted = threading.Lock()

    extra = {}

    def emit(train_obj):
        """Rank 0 prints the ONE JSON line, exactly once (also from the watchdog below)."""
        if not emitted.acquire(blocking=False) or out is None:
            return
        out.update(extra)
        if rccl_error:
            out["rccl_error"] = rccl_error
        if strict is not None:
            out["f32_mfma_reference"] = strict
        if train_obj is not None:
            out["train"] = train_obj
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_images)
        print(json.dumps(out), flush=True)

    # ---- secondary measurements.  Never allowed to break the headline line above: an exception is reported in the object, and a
    # collective that does not return (a rank died) is cut off by a watchdog that prints the headline and exits NON-ZERO, naming the leg.
    train_obj = None
    leg = {"name": None}
    if args.train_steps > 0 or args.x101_steps > 0 or not args.no_two_pipelines or not args.no_host_inclusive or not args.no_power_probe:
        def on_stall():
            msg = f"secondary leg '{leg['name']}' did not finish within {args.train_timeout} s"
            log(f"rank {rank}: {msg}; emitting the inference line without it and exiting with code 3")
            extra["stalled_leg"] = leg["name"]
            emit(train_obj if train_obj is not None else {"error": msg})
            os._exit(3)
        dog = threading.Timer(args.train_timeout, on_stall)
        dog.daemon = True
        dog.start()
        if mode == "f16x3" and rank == 0 and out is not None and not args.no_power_probe:
            leg["name"] = "power_probe"
            try:
                out["roofline"]["power_probe"] = power_probe_leg(ctx, dev)
            except Exception as e:   # noqa: BLE001
                out["roofline"]["power_probe"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_host_inclusive:
            leg["name"] = "host_inclusive"
            try:
                extra["host_inclusive"] = host_inclusive_leg(model, imgs, dev, world, max(4, args.steps // 4), ranks)
            except Exception as e:   # noqa: BLE001
                extra["host_inclusive"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if not args.no_two_pipelines and mode == "f16x3":
            leg["name"] = "two_pipelines"
            try:
                extra["two_pipelines"] = two_pipelines_leg(local_rank, dev, rank, world, args.steps, ranks, params, imgs)
            except Exception as e:   # noqa: BLE001
                extra["two_pipelines"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if args.train_steps > 0 and rccl_error:
            train_obj = {"error": f"not run: the gradient exchange needs the RCCL communicator ({rccl_error})"}
        elif args.train_steps > 0:
            leg["name"] = "train"
            try:
                train_obj = train_leg(ctx, model, dev, rank, world, args.train_warmup, args.train_steps, ranks)
            except Exception as e:   # noqa: BLE001
                train_obj = {"error": f"{type(e).__name__}: {e}"[:300]}
        if args.x101_steps > 0:
            leg["name"] = "x101_2048"
            try:
                model.close()
                extra["x101_2048"] = x101_leg(ctx, dev, rank, world, args.x101_steps, ranks)
            except Exception as e:   # noqa: BLE001
                extra["x101_2048"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog.cancel()
    emit(train_obj)
    # a multi-GPU line whose gradient exchange did not run over all N ranks is not a scaling measurement: say so with the exit code
    bad_ranks = (world > 1 and not staged and isinstance(train_obj, dict) and "grad_exchange" in train_obj
                 and train_obj["grad_exchange"]["rccl_ranks"] != world)
    if bad_ranks:
        log(f"rank {rank}: FATAL: grad_exchange.rccl_ranks = {train_obj['grad_exchange']['rccl_ranks']} but --gpus {world}")
    if world > 1:
        ranks.barrier()
        if rccl:
            from ampis_amd.utils import comm
            comm.detach_rccl()
        torch.distributed.destroy_process_group()
    if bad_ranks:
        sys.exit(4)


if __name__ == "__main__":
    main(ARGS)
