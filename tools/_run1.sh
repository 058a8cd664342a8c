set -e
timeout -k 10 600 python -m pytest tests/test_e2e_gpu.py tests/test_rle_string_gpu.py tests/test_predictor_gpu.py tests/test_edge_cases_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -2
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-strict --no-host-inclusive 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['events_on'], d['two_pipelines']['value'])"
done
