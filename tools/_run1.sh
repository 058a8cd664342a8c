set -e
timeout -k 10 600 python -m pytest tests/test_rpn_nms_levels_gpu.py tests/test_e2e_gpu.py tests/test_stages_gpu.py tests/test_edge_cases_gpu.py -x -q > gpurun_out/r3_lvl_tests.log 2>&1 || { tail -30 gpurun_out/r3_lvl_tests.log; exit 1; }
tail -3 gpurun_out/r3_lvl_tests.log
for mode in flat new flat new; do
  if [ $mode = flat ]; then export AMP_RPN_NMS=flat; else unset AMP_RPN_NMS; fi
  timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --train-steps 0 --x101-steps 0 --no-strict 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('$mode', d['value'], d['ms_per_step'], d.get('two_pipelines',{}).get('value'))"
done
