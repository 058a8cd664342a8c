#!/bin/bash
# ON THE GPU BOX: per-dispatch FETCH_SIZE / WRITE_SIZE of the conv launches of an inference step (one counter per pass, --kernel-trace only).
# usage: bash tools/traffic_table.sh <tag>      (environment, e.g. AMP_KORDER=1, is inherited by the profiled python3)
set -o pipefail
TAG=${1:-traffic}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/layer_roofline.py --steps 2 --json $OUT/L.json > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $ROOT/tools/layer_roofline.py --steps 2 > $OUT/write.log 2>&1 || { echo "write pass failed"; tail -5 $OUT/write.log; exit 1; }
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
echo "traffic passes done: $OUT"
