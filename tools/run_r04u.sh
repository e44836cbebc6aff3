cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final_r04
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/final_r04/full_gpu_tests.log 2>&1
tail -6 gpurun_out/final_r04/full_gpu_tests.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/final_r04/smoke.log 2>&1
tail -2 gpurun_out/final_r04/smoke.log
