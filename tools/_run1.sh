set -e
timeout -k 10 900 python -m pytest tests/test_conv_modes_gpu.py tests/test_e2e_gpu.py -x -q > gpurun_out/r3_dual_tests.log 2>&1 || { tail -40 gpurun_out/r3_dual_tests.log; exit 1; }
tail -2 gpurun_out/r3_dual_tests.log
bash tools/kstat_quick.sh kq17 | grep "conv_split_kernel<128, 256\|conv_glds_kernel<64\|all kernels"
export AMP_NO_DUAL=1
bash tools/kstat_quick.sh kq18 | grep "conv_split_kernel<128, 256\|conv_glds_kernel<64\|all kernels"
