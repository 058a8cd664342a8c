#!/bin/bash
# end-of-round measurements on one box: X-101 training, training profile + timeline, the driver-sized default bench run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for b in 4 8; do B=$b timeout -k 10 280 python3 tools/bench_x101_train.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-120; done
bash tools/profile_round.sh ${1:-prof_round_end}_train train > gpurun_out/${1:-prof_round_end}_train.log 2>&1; tail -2 gpurun_out/${1:-prof_round_end}_train.log
bash tools/train_trace.sh > gpurun_out/train_trace_round_end.txt 2>&1; head -34 gpurun_out/train_trace/timeline.txt
cd $ROOT && timeout -k 10 500 python3 bench.py > gpurun_out/bench_default_round_end.json 2> gpurun_out/bench_default_round_end.err
python3 - <<PY
import json
d = json.loads(open("gpurun_out/bench_default_round_end.json").read().strip().splitlines()[-1])
print("inference", d["value"], "pipelined", d["two_pipelines"]["value"], "train", d["train"]["value"], d["train"]["ms_per_step"], "x101", d["x101_2048"]["value"],
      "train conv TFLOP/s", d["train"]["roofline"]["achieved"], "wgrad", d["train"]["roofline"]["wgrad"]["achieved"])
PY
