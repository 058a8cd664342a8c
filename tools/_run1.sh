set -e
timeout -k 10 600 python -m pytest tests/test_stages_gpu.py tests/test_e2e_gpu.py tests/test_pipeline_gpu.py tests/test_concurrency_gpu.py -x -q 2>&1 | tail -2
bash tools/kstat_quick.sh kq10 | grep "roi_\|all kernels"
