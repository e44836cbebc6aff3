cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ad
timeout -k 10 600 python -m pytest tests/test_trans_u_net_gpu.py tests/test_conv_bf16_gpu.py -q -m gpu > gpurun_out/r04ad/tests.log 2>&1
tail -4 gpurun_out/r04ad/tests.log
