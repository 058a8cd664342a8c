#!/bin/bash
# kernel trace of X-101 training steps (tools/bench_x101_train.py) -> per-step busy time and kernel totals (tools/train_timeline.py)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/x101_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_x101_train.py > $OUT/log.txt 2>&1
tail -2 $OUT/log.txt
python3 $ROOT/tools/train_timeline.py $(ls $OUT/*/*kernel_trace.csv | head -1) > $OUT/timeline.txt; cat $OUT/timeline.txt
