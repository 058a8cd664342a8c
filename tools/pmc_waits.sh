#!/bin/bash
# Run ON THE GPU BOX: where the waves of every kernel of a step spend their cycles (parked in s_waitcnt / barriers, stalled at issue, issuing) and how
# many scalar-memory instructions they execute -- the comparison that found wgrad_split_kernel's two stalls.  usage: pmc_waits.sh infer|train
MODE=${1:-infer}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_waits_$MODE
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$MODE" = train ]; then
  BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --train-steps 3 --train-warmup 1 --x101-steps 0 --no-two-pipelines --no-host-inclusive --no-strict"
else
  BENCH="python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-two-pipelines --no-host-inclusive --no-strict"
fi
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- $BENCH > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
done
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*", "", k)[:64]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))
print(f"{'kernel':64s} {'launches':>8s} {'busy%':>6s} {'parked':>7s} {'stall':>6s} {'issue':>6s} {'smem/vmem':>9s} {'salu/valu':>9s}")
tot = sum(v.get("SQ_BUSY_CYCLES", 0) for _, v in rows)
for k, v in rows[:32]:
    wc = max(v.get("SQ_WAVE_CYCLES", 0), 1)
    print(f"{k:64s} {n[k]:8d} {100 * v.get('SQ_BUSY_CYCLES', 0) / tot:6.1f} {v.get('SQ_WAIT_ANY', 0) / wc:7.2f} {v.get('SQ_WAIT_INST_ANY', 0) / wc:6.2f} {v.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.2f} "
          f"{v.get('SQ_INSTS_SMEM', 0) / max(v.get('SQ_INSTS_VMEM', 0), 1):9.2f} {v.get('SQ_INSTS_SALU', 0) / max(v.get('SQ_INSTS_VALU', 0), 1):9.2f}")
PY
find $OUT -name '*kernel_trace.csv' -delete
