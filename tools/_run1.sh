set -e
timeout -k 10 900 python -m pytest tests/test_stages_gpu.py tests/test_rpn_nms_levels_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | tail -2
bash tools/kstat_quick.sh kq20 | grep "topk\|all kernels"
