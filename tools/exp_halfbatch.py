"""Experiment (not product): what could ONE amp_model_infer gain by running its batch as micro-batches on several streams?
Emulated with the existing machinery -- P contexts (own stream, own workspace), each given B/P of the 8 images, one host thread each,
all P started together and JOINED per step (a step ends when every part has its results on the host: the contract of one call).
Compared with the plain call (P = 1) and with the free-running two-pipeline mode of bench.py (no join per step)."""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ampis_amd import _lib, params as P, synth
from ampis_amd.model import MaskRCNN, RLE_STRINGS

BATCH, SIZE, K, DETS = 8, 1024, 2, 200
imgs, _ = synth.batch(BATCH, SIZE, SIZE, first_index=0)
params = P.init_params(K, seed=0, style="spread")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def build(parts, stagger_us=0):
    per = BATCH // parts
    out = []
    for i in range(parts):
        c = _lib.Context(0)
        m = MaskRCNN(c, K, max_batch=per, max_h=SIZE, max_w=SIZE, max_out_hw=SIZE, detections_per_image=DETS)
        m.load_params(params)
        m.set_rle_output(RLE_STRINGS)
        sub = np.ascontiguousarray(imgs[i * per:(i + 1) * per])
        d = c.malloc(sub.nbytes); c.h2d(d, sub)
        out.append((c, m, d, per))
    return out


def run_joined(parts, steps, stagger_us=0):
    pipes = build(parts)
    bar = threading.Barrier(parts + 1)
    dets = [0] * parts
    def worker(i):
        c, m, d, per = pipes[i]
        for s in range(steps + 3):
            bar.wait()
            if stagger_us and i:
                time.sleep(i * stagger_us * 1e-6)
            r = m.infer_raw(None, device_ptr=d, shape=(per, SIZE, SIZE))
            dets[i] = sum(r.n[b] for b in range(per))
            bar.wait()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(parts)]
    for t in th: t.start()
    for s in range(3):
        bar.wait(); bar.wait()
    t0 = time.perf_counter()
    for s in range(steps):
        bar.wait(); bar.wait()
    el = time.perf_counter() - t0
    for t in th: t.join()
    for c, m, d, per in pipes:
        m.close(); c.free(d); c.close()
    return el / steps * 1e3, sum(dets)


for parts, stag in ((1, 0), (2, 0), (2, 1500), (2, 3000), (4, 0), (4, 1000), (1, 0)):
    ms, nd = run_joined(parts, STEPS, stag)
    print(f"parts {parts} stagger {stag:5d} us: {ms:7.3f} ms per batch of {BATCH} = {BATCH / ms * 1e3:6.1f} images/s   ({nd} detections)", flush=True)


def run_pipeline(depth, steps):
    """amp_pipeline: `depth` full batches in flight from this one thread (consecutive batches overlap, nothing is joined per step)."""
    from ampis_amd.model import InferPipeline
    pipe = InferPipeline(0, K, depth=depth, max_batch=BATCH, max_h=SIZE, max_w=SIZE, max_out_hw=SIZE, detections_per_image=DETS)
    pipe.load_params(params)
    c = _lib.Context(0)
    d = c.malloc(imgs.nbytes); c.h2d(d, imgs); c.sync()
    def loop(n):
        pend, nd = [], 0
        for _ in range(n):
            if len(pend) == depth:
                r = pipe.wait_raw(pend.pop(0)); nd += sum(r.n[b] for b in range(BATCH))
            pend.append(pipe.submit(device_ptr=d, shape=(BATCH, SIZE, SIZE)))
        while pend:
            r = pipe.wait_raw(pend.pop(0)); nd += sum(r.n[b] for b in range(BATCH))
        return nd
    loop(4)
    t0 = time.perf_counter()
    nd = loop(steps)
    el = time.perf_counter() - t0
    pipe.close(); c.free(d); c.close()
    return el / steps * 1e3, nd


for depth in (1, 2, 3, 2, 1):
    ms, nd = run_pipeline(depth, STEPS)
    print(f"amp_pipeline depth {depth}: {ms:7.3f} ms per batch of {BATCH} = {BATCH / ms * 1e3:6.1f} images/s   ({nd} detections over {STEPS} batches)", flush=True)
