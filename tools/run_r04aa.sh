cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04aa
timeout -k 10 700 python -m pytest tests/test_ema_net_gpu.py tests/test_step_graph_gpu.py "tests/test_switches_gpu.py" -q -m gpu -k "ema or EMA or step_graph" > gpurun_out/r04aa/tests.log 2>&1
tail -8 gpurun_out/r04aa/tests.log
timeout -k 10 300 python bench.py --workload emanet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04aa/bench.err | grep "^{" > gpurun_out/r04aa/bench.json
python -c "import json; d=json.load(open('gpurun_out/r04aa/bench.json')); print('emanet', d['value'], d['ms_per_step'], d['library_calls_per_step'], d['library_ms_per_step'])"
SIS_SUB_IMAGE_UNITS=0 timeout -k 10 300 python bench.py --workload emanet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04aa/bench0.err | grep "^{" > gpurun_out/r04aa/bench0.json
python -c "import json; d=json.load(open('gpurun_out/r04aa/bench0.json')); print('emanet plain arrangement', d['value'], d['ms_per_step'])"
