cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04r
timeout -k 10 500 python -m pytest tests/test_seg_ops_gpu.py tests/test_ema_net_gpu.py -q -m gpu > gpurun_out/r04r/tests.log 2>&1
tail -12 gpurun_out/r04r/tests.log
timeout -k 10 200 python tools/layer_times.py emanet bn_ > gpurun_out/r04r/bn_layers.txt 2>&1
cat gpurun_out/r04r/bn_layers.txt
SIS_BN_SPLIT=0 timeout -k 10 200 python tools/layer_times.py emanet bn_ > gpurun_out/r04r/bn_layers_nosplit.txt 2>&1
cat gpurun_out/r04r/bn_layers_nosplit.txt
timeout -k 10 300 python bench.py --workload emanet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04r/bench_ema.err | grep "^{" > gpurun_out/r04r/bench_ema.json
python -c "import json; d=json.load(open('gpurun_out/r04r/bench_ema.json')); print('emanet', d['value'], d['ms_per_step'])"
SIS_BN_SPLIT=0 timeout -k 10 300 python bench.py --workload emanet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04r/bench_ema0.err | grep "^{" > gpurun_out/r04r/bench_ema0.json
python -c "import json; d=json.load(open('gpurun_out/r04r/bench_ema0.json')); print('emanet nosplit', d['value'], d['ms_per_step'])"
