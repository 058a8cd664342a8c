#!/bin/bash
# Run ON THE GPU BOX: per-kernel time of the headline inference loop alone (f16x3, no secondary legs): gpurun -- 'bash tools/kstat_quick.sh <tag>'
set -o pipefail
TAG=${1:-kq}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-two-pipelines --no-host-inclusive --no-strict > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
find $OUT -name '*kernel_trace.csv' -delete
python3 $ROOT/tools/kstats.py $(find $OUT/stats -name '*kernel_stats.csv' | head -1) 12 45
