#!/bin/bash
# L2 <- fabric bytes of roi_align_split_kernel per launch, index order against the XCD-major order (tools/bench_roi.py runs both, in turn)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_roi
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_roi.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "roi_align_split_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
# the XCD section of bench_roi.py: per (P, box set) the modes 0,1,0,1 with 2 warm-up + 5 timed + 1 extra launch each = 8 launches per mode
vals = [float(r["Counter_Value"]) * 2 * 1024 / 1e9 for r in rows]      # KiB, doubled (gfx950 correction) -> GB
print(len(vals), "launches of roi_align_split_kernel")
tail = vals[-4 * 4 * 8:]
names = ["P=7 uniform", "P=7 clustered", "P=14 uniform", "P=14 clustered"]
for i, nm in enumerate(names):
    blk = tail[i * 32:(i + 1) * 32]
    m0 = blk[0:8] + blk[16:24]; m1 = blk[8:16] + blk[24:32]
    print(f"{nm:16s} index order {sum(m0) / len(m0):6.3f} GB read per launch   XCD-major {sum(m1) / len(m1):6.3f} GB")
PY
