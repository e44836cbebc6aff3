cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04m
timeout -k 10 1000 python -m pytest tests/test_switches_gpu.py -q -m gpu > gpurun_out/r04m/switch_tests.log 2>&1
tail -40 gpurun_out/r04m/switch_tests.log
