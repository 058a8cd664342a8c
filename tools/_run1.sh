set -e
timeout -k 10 300 python tools/bench_roi.py > gpurun_out/r3_roi_tab.log 2>&1 || { tail -30 gpurun_out/r3_roi_tab.log; exit 1; }
tail -20 gpurun_out/r3_roi_tab.log
timeout -k 10 600 python -m pytest tests/test_stages_gpu.py tests/test_conv_modes_gpu.py tests/test_e2e_gpu.py -x -q 2>&1 | tail -3
