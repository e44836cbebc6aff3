cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ac
timeout -k 10 600 python -m pytest tests/test_gemm_bf16_gpu.py tests/test_trans_u_net_gpu.py -q -m gpu -k "swap_last2 or trans_u_net" > gpurun_out/r04ac/tests.log 2>&1
tail -6 gpurun_out/r04ac/tests.log
timeout -k 10 400 python bench.py --workload transunet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04ac/bench.err | grep "^{" > gpurun_out/r04ac/bench.json
python -c "import json; d=json.load(open('gpurun_out/r04ac/bench.json')); print('transunet', d['value'], d['ms_per_step'])"
