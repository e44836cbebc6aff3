cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04t
timeout -k 10 600 python -m pytest tests/test_generator_gpu.py -q -m gpu > gpurun_out/r04t/tests.log 2>&1
tail -5 gpurun_out/r04t/tests.log
timeout -k 10 300 python tools/bench_upfir.py > gpurun_out/r04t/upfir.txt 2>&1
tail -12 gpurun_out/r04t/upfir.txt
timeout -k 10 300 python bench.py --workload synthesis --steps 20 --warmup 5 --no-cpu-baseline 2> gpurun_out/r04t/bench.err | grep "^{" > gpurun_out/r04t/bench.json
python -c "import json; d=json.load(open('gpurun_out/r04t/bench.json')); print('synth', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()})"
