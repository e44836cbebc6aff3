cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final_r04
timeout -k 10 1100 python -m pytest tests -q -m gpu > gpurun_out/final_r04/full_gpu_tests.log 2>&1
tail -15 gpurun_out/final_r04/full_gpu_tests.log
