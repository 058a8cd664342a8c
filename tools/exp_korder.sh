#!/bin/bash
# Round-4 experiment: channel-major against tap-major K order of conv_split_kernel in ONE gpurun call -- correctness, three
# alternations of the bench, the per-launch table of both, per-dispatch traffic of the variant, then the full-size gate and smoke.
set -o pipefail
O=gpurun_out/c2; mkdir -p $O
BQ="--steps 30 --warmup 5 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-strict --no-two-pipelines --no-host-inclusive"
timeout -k 10 200 python tools/exp_korder.py > $O/exp_korder.log 2>&1 || { tail -20 $O/exp_korder.log; exit 1; }
tail -3 $O/exp_korder.log
for ko in 0 1 0 1 0 1; do
  AMP_KORDER=$ko timeout -k 10 200 python bench.py $BQ > $O/bench_ko$ko.$RANDOM.log 2>&1 || { echo bench fail; exit 1; }
done
grep -h '"value"' $O/bench_ko*.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['frac'])
"
AMP_KORDER=0 timeout -k 10 200 python tools/layer_roofline.py --json $O/layers_ko0.json > $O/layers_ko0.txt 2>&1 || exit 1
AMP_KORDER=1 timeout -k 10 200 python tools/layer_roofline.py --json $O/layers_ko1.json > $O/layers_ko1.txt 2>&1 || exit 1
tail -3 $O/layers_ko0.txt; tail -3 $O/layers_ko1.txt
true
AMP_KORDER=1 bash tools/traffic_table.sh c2/traffic_ko1 || exit 1
timeout -k 10 600 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "oracle" > $O/fullsize.log 2>&1; echo "fullsize rc $?"
grep -h "full-size gate\|HIP vs\|fp32 oracle vs\|passed\|failed\|Error" $O/fullsize.log | cut -c1-900
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc $?"; grep smoke $O/smoke.log | cut -c1-900
