cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04p
timeout -k 10 600 python -m pytest tests/test_distributed_gpu.py tests/test_bench_launch_gpu.py -q -m gpu > gpurun_out/r04p/dist_tests.log 2>&1
tail -5 gpurun_out/r04p/dist_tests.log
timeout -k 10 400 python bench.py --workload emanet --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r04p/bench_ema.err | grep "^{" > gpurun_out/r04p/bench_ema.json
python -c "import json; d=json.load(open('gpurun_out/r04p/bench_ema.json')); print(d['value'], d.get('data_parallel_rehearsal'))"
