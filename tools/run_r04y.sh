cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04y
timeout -k 10 400 python bench.py --workload transunet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04y/bench.err | grep "^{" > gpurun_out/r04y/bench.json
python -c "import json; d=json.load(open('gpurun_out/r04y/bench.json')); print('transunet', d['value'], d['ms_per_step'])"
SIS_UP2_DIRECT=0 timeout -k 10 400 python bench.py --workload transunet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> gpurun_out/r04y/bench0.err | grep "^{" > gpurun_out/r04y/bench0.json
python -c "import json; d=json.load(open('gpurun_out/r04y/bench0.json')); print('transunet tiled upsampling', d['value'], d['ms_per_step'])"
