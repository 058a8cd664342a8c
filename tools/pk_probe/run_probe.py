"""Driver of tools/pk_probe/pk_opsel_probe.hip: the probe kernel alone, then beside a partner load (two threads running the whole
inference path through libampis_hip.so, i.e. what the failing three-context runs of round 3 had in flight)."""
import ctypes as C, os, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
NPAT = 18
NAMES = ["pk_add plain", "pk_add op_sel:[0,1] op_sel_hi:[1,0]", "pk_add op_sel_hi:[1,0]", "pk_add op_sel:[0,1]", "pk_add op_sel:[1,0] op_sel_hi:[0,1]",
         "pk_mul op_sel:[0,1] op_sel_hi:[1,0]", "pk_mul plain", "pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1]", "pk_fma plain", "pk_add op_sel:[0,1] op_sel_hi:[1,0] neg",
         "pk_add op_sel:[1,0] (src0.hi broadcast)", "pk_fma op_sel:[0,0,1] op_sel_hi:[1,1,0] (src2 swapped)", "pk_fma op_sel:[1,0,0] op_sel_hi:[0,1,1] (src0 swapped)",
         "fma_mix_f32 op_sel:[1,0,1] op_sel_hi:[1,0,1]", "fma_mix_f32 op_sel:[0,0,0] op_sel_hi:[1,0,1]", "pk_mov_b32 op_sel:[1,0]", "pk_mov_b32 op_sel:[0,1]",
         "pk_mul op_sel:[0,1] (src1.hi broadcast)"]
P = C.CDLL(os.path.join(HERE, "libpk_probe.so"))
BLOCKS = int(os.environ.get("PK_BLOCKS", "8"))
assert P.pk_probe_init(BLOCKS) == 0


def probe(launches, iters=2048):
    counts = (C.c_ulonglong * (NPAT + 2))()
    log = (C.c_uint * (256 * 8))()
    assert P.pk_probe_run(BLOCKS, iters, launches, counts, log) == 0
    return list(counts), list(log)


def report(tag, before, after, log, n_ops):
    d = [a - b for a, b in zip(after, before)]
    print(f"[{tag}] {n_ops:.2e} results checked per form; mismatches per form:")
    for i in range(NPAT):
        print(f"    {NAMES[i]:45s} {d[i]}")
    return d


c0, _ = probe(1)
t0 = time.time()
c1, lg = probe(400)
n_ops = 400 * BLOCKS * 256 * 2048
print(f"alone: {time.time() - t0:.1f} s")
report("alone", c0, c1, lg, n_ops)

from ampis_amd import _lib, params as PP, synth      # noqa: E402
from ampis_amd.model import MaskRCNN                  # noqa: E402
K, B, H, W, D = 2, 2, 256, 320, 40
params = PP.init_params(K, seed=3, style="spread")
imgs = synth.batch(B, H, W, first_index=0)[0]
stop = False
models = []
for _ in range(2):
    c = _lib.Context(0)
    m = MaskRCNN(c, K, max_batch=B, max_h=H, max_w=W, max_out_hw=max(H, W), detections_per_image=D)
    m.load_params(params)
    models.append(m)


def partner(m):
    while not stop:
        m.infer(imgs)


th = [threading.Thread(target=partner, args=(m,)) for m in models]
for t in th:
    t.start()
time.sleep(1.0)
t0 = time.time()
c2, lg = probe(400)
el = time.time() - t0
stop = True
for t in th:
    t.join()
print(f"beside two inference threads: {el:.1f} s")
d = report("beside inference", c1, c2, lg, n_ops)
tot = c2[NPAT]
print("first logged mismatches (form, lane, wave, iter, got, want):")
import struct
f = lambda u: struct.unpack("f", struct.pack("I", u))[0]
for i in range(min(tot, 24)):
    e = lg[8 * i: 8 * i + 8]
    print(f"    {NAMES[e[0]]:45s} lane {e[1]:2d} wave {e[2]:3d} iter {e[3]:5d} got ({f(e[4]):.6g}, {f(e[5]):.6g}) want ({f(e[6]):.6g}, {f(e[7]):.6g})")
