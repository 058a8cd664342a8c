#!/bin/bash
# Run ON THE GPU BOX: LDS / wait counters of wgrad_split_kernel against conv_split_kernel<128x256> on the same GEMM shape (fpn.out.p2, mask 3x3)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_wv
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AMP_SPLIT_IN=3 AMP_ONLY=fpn.out.p2,mask.3x3
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  for prog in bench_wgrad bench_conv; do
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/${prog}_$i -- python3 $ROOT/tools/$prog.py > $OUT/${prog}_$i.log 2>&1 || { echo "$prog $i failed"; tail -3 $OUT/${prog}_$i.log; }
  done
done
python3 - <<PY
import csv, glob, collections, re
for prog in ("bench_wgrad", "bench_conv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
    for f in glob.glob("$OUT/%s_*/*/*counter_collection.csv" % prog):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "wgrad_split_kernel" in n: k = "wgrad_split_kernel"
            elif "conv_split_kernel<128, 256, 1" in n: k = "conv_split_kernel<128,256,1>"
            else: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"): cnt[(k, r["Counter_Name"])] += 1
    for k, v in agg.items():
        print(prog, k)
        for c, x in sorted(v.items()):
            print(f"   {c:28s} {x:16.0f}")
PY
find $OUT -name '*kernel_trace.csv' -delete
