#!/bin/bash
# kernel trace of a few training steps -> per-step busy time and top non-conv kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/train_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 6 --train-warmup 2 --x101-steps 0 --no-two-pipelines --no-host-inclusive --no-strict > $OUT/log.txt 2>&1
python3 $ROOT/tools/train_timeline.py $(ls $OUT/*/*kernel_trace.csv | head -1) > $OUT/timeline.txt; cat $OUT/timeline.txt
python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0] for r in rows]
# training steps: delimited by anchor_match_kernel launches
idx = [i for i, n in enumerate(names) if "anchor_match_kernel" in n]
a, b = idx[3], idx[4]
t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b])
print(f"one training step: {b - a} launches, wall {(t1 - t0) / 1e6:.2f} ms, kernels busy {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for r, n in zip(rows[a:b], names[a:b]):
    k = re.sub(r"<.*", "", n)[-48:]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"   {k:50s} {c:5d} {ms:8.3f} ms")
# largest gaps
gaps = []
for i in range(a + 1, b):
    g = int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])
    gaps.append((g, names[i - 1][-40:], names[i][-40:]))
print("largest gaps:")
for g, p, n in sorted(gaps, reverse=True)[:8]:
    print(f"   {g / 1e3:8.1f} us  after {p}  before {n}")
PY
