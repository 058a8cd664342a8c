set -e
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r3_full7.log 2>&1 || { tail -40 gpurun_out/r3_full7.log; exit 1; }
tail -3 gpurun_out/r3_full7.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
