set -e
timeout -k 10 900 python -m pytest tests/test_stages_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py tests/test_fullsize_gpu.py tests/test_pipeline_gpu.py -x -q > gpurun_out/r3_x_tests.log 2>&1 || { tail -40 gpurun_out/r3_x_tests.log; exit 1; }
tail -2 gpurun_out/r3_x_tests.log
bash tools/kstat_quick.sh kq19 | grep "roi_\|all kernels"
