#!/bin/bash
# Run ON THE GPU BOX: per-kernel time of the notebook flow at batch 1 (tools/bench_predictor.py): gpurun -- 'bash tools/kstat_predictor.sh <tag>'
set -o pipefail
TAG=${1:-kp}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/bench_predictor.py > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
find $OUT -name '*kernel_trace.csv' -delete
tail -2 $OUT/bench.log
python3 $ROOT/tools/kstats.py $(find $OUT/stats -name '*kernel_stats.csv' | head -1) 1 40
