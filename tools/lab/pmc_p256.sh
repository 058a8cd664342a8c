#!/bin/bash
# LDS / fetch counters of the wide 3x3 layer on the patch kernel (AMP_PATCH256=1) and on the ring kernel (0)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_p256
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  export AMP_PATCH256=$v
  for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS" "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | tr ' ' '_')
    timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p${v}_$tag -- python3 $ROOT/tools/lab/run_conv.py 8 256 256 256 256 6 > $OUT/log_p${v}_$tag.txt 2>&1 || echo "failed: $v $set"
  done
done
python3 - <<PY
import csv, glob, collections
for v in (1, 0):
    print("AMP_PATCH256 =", v)
    for d in sorted(glob.glob("$OUT/p%d_*" % v)):
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            agg = collections.defaultdict(lambda: [0, 0.0])
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "conv3x3_patch" in k or "conv_split_kernel" in k:
                    a = agg[(k[:60], r["Counter_Name"])]; a[0] += 1; a[1] += float(r["Counter_Value"])
            for (k, c), (n, s) in sorted(agg.items()):
                print(f"   {k:60s} {c:28s} launches {n:3d}  per launch {s / n:16.1f}")
PY
