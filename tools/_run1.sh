set -e
timeout -k 10 900 python -m pytest tests/test_stages_gpu.py tests/test_rpn_nms_levels_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py tests/test_train_fwd_gpu.py -x -q > gpurun_out/r3_sort_tests.log 2>&1 || { tail -30 gpurun_out/r3_sort_tests.log; exit 1; }
tail -3 gpurun_out/r3_sort_tests.log
bash tools/kstat_quick.sh kq2 | grep -v "conv_\|rocclr" 
