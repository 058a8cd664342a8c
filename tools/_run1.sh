set -e
timeout -k 10 900 python -m pytest tests/test_conv_modes_gpu.py tests/test_e2e_gpu.py tests/test_edge_cases_gpu.py tests/test_concurrency_gpu.py tests/test_pipeline_gpu.py -x -q > gpurun_out/r3_stem_tests.log 2>&1 || { tail -30 gpurun_out/r3_stem_tests.log; exit 1; }
tail -2 gpurun_out/r3_stem_tests.log
bash tools/kstat_quick.sh kq15 | grep "stem\|all kernels"
timeout -k 10 200 python tools/_probe_conc.py 2>&1 | tail -2
