set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace1; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-two-pipelines --no-host-inclusive --no-strict > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
ls $OUT/*/ | head -3
