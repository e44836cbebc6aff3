cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04s
timeout -k 10 800 python -m pytest tests/test_bench_launch_gpu.py -q -m gpu > gpurun_out/r04s/tests.log 2>&1
tail -15 gpurun_out/r04s/tests.log
