#!/bin/bash
# Run ON THE GPU BOX: SQ counters of one conv layer of tools/bench_conv.py (default res4.3x3 on the trunk-native split data path), one
# counter set per rocprofv3 pass with --kernel-trace only.  Output: gpurun_out/<tag>/pmc_<set>/...counter_collection.csv
set -o pipefail
TAG=${1:-pmc_conv}
LAYER=${2:-res4.3x3}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AMP_SPLIT_IN=1 AMP_ONLY=$LAYER
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VALU" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-60)
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$name -- python3 $ROOT/tools/bench_conv.py > $OUT/$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/$name.log; }
  echo "pmc $name done"
done
find $OUT -name '*kernel_trace.csv' -delete
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k in tot:
    if "conv" not in k: continue
    print(k)
    for c in sorted(tot[k]):
        print(f"   {c:36s} {tot[k][c] / cnt[k][c]:16.0f} per launch ({cnt[k][c]} launches)")
PY
