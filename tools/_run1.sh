set -e
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py tests/test_predictor_gpu.py tests/test_pipeline_gpu.py tests/test_concurrency_gpu.py tests/test_fullsize_gpu.py tests/test_x101_gpu.py -x -q > gpurun_out/r3_q_tests.log 2>&1 || { tail -30 gpurun_out/r3_q_tests.log; exit 1; }
tail -2 gpurun_out/r3_q_tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-strict --no-host-inclusive 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['two_pipelines']['value'])"
done
