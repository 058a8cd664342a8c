#!/bin/bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh <tag>'): rocprofv3 kernel statistics of the bench command and the four PMC
# passes of MI355X_MICROARCH.md's HBM/rocprofv3 section, each counter set in its own run with --kernel-trace only.  Raw output
# lands under gpurun_out/<tag>/ (scratch); tools/summarize_profiles.py turns it into the files committed under profiles/.
set -o pipefail
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
MODE=${2:-infer}    # infer: the headline workload; train: the training step (wgrad / dgrad kernels)
if [ "$MODE" = train ]; then
  BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 6 --train-warmup 2 --x101-steps 0 --no-two-pipelines --no-host-inclusive"
else
  BENCH="python3 $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-two-pipelines --no-host-inclusive"
fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats_bench.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/stats_bench.log; exit 1; }
grep '^{' $OUT/stats_bench.log > $OUT/bench_line_under_rocprof.json
echo "stats done"
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_$name -- $BENCH --no-strict > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
  echo "pmc $name done"
done
# keep the merge-back small: drop the per-dispatch traces of the PMC runs (the counter files carry the kernel names)
find $OUT -name '*kernel_trace.csv' -path '*pmc_*' -delete
ls -la $OUT
