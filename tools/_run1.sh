set -e
timeout -k 10 1000 python -m pytest tests/test_train_bwd_gpu.py tests/test_backward_gpu.py tests/test_comm_gpu.py tests/test_comm_world2_gpu.py tests/test_bitmask_gt_gpu.py tests/test_trainer_gpu.py tests/test_x101_gpu.py -x -q > gpurun_out/r3_async_tests.log 2>&1 || { tail -30 gpurun_out/r3_async_tests.log; exit 1; }
tail -2 gpurun_out/r3_async_tests.log
for m in 1 0 1 0; do
export AMP_ASYNC_REDUCE=$m
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --train-steps 30 --train-warmup 5 --x101-steps 0 --no-strict --no-host-inclusive --no-two-pipelines 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); t=d['train']; print('async $m train', t['value'], t['ms_per_step'])"
done
