#!/bin/bash
# Run ON THE GPU BOX: kernel trace of the notebook flow at batch 1 (tools/bench_predictor.py) -> wall / busy / idle of one call, largest gaps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pred_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_predictor.py > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 - <<PY
import csv, glob, re, collections
f = glob.glob("$OUT/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0] for r in rows]
idx = [i for i, n in enumerate(names) if "stem_pool_u8" in n]
print(len(rows), "launches;", len(idx), "calls")
for a, b in ((idx[8], idx[9]), (idx[-3], idx[-2])):
    t0, t1 = int(rows[a]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b])
    print(f"one call: {b - a} launches, wall {(t1 - t0) / 1e6:.3f} ms, kernels busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r, n in zip(rows[a:b], names[a:b]):
        k = re.sub(r"<.*", "", n)[-48:]
        agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"   {k:50s} {c:5d} {ms:8.3f} ms")
    gaps = []
    for i in range(a + 1, b + 1):
        g = int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])
        gaps.append((g, names[i - 1][-40:], names[i][-40:]))
    print("   largest gaps:")
    for g, p, n in sorted(gaps, reverse=True)[:10]:
        print(f"   {g / 1e3:8.1f} us  after {p}  before {n}")
    print("   gaps > 2 us:", sum(1 for g in gaps if g[0] > 2000), "sum", sum(g[0] for g in gaps if g[0] > 2000) / 1e3, "us")
PY
find $OUT -name '*kernel_trace.csv' -delete
