cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04ae
timeout -k 10 600 python -m pytest tests/test_switches_gpu.py -q -m gpu -k "WGRAD_SIDE_STREAM or _PW_WGRAD_OWN" > gpurun_out/r04ae/tests.log 2>&1
tail -4 gpurun_out/r04ae/tests.log
for v in 0 1 0 1; do
SIS_CONV_WGRAD_STREAM=$v timeout -k 10 400 python bench.py --workload transunet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04ae/bench$v.err | grep "^{" > gpurun_out/r04ae/bench$v.json
python -c "import json; d=json.load(open('gpurun_out/r04ae/bench$v.json')); print('transunet side=$v', d['value'], d['ms_per_step'], d['config']['hip_graph'])"
done
