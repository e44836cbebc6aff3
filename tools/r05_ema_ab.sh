set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05j}; mkdir -p $O
python -m pytest tests/test_conv1x1_f32_gpu.py tests/test_hip_conv_gpu.py tests/test_seg_ops_gpu.py -x -q -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
python bench.py --workload emanet --steps 30 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> $O/ema.err | grep "^{" > $O/ema.json
python -c "import json; d=json.load(open('$O/ema.json')); print('emanet', d['value'], d['ms_per_step'])"
python bench.py --workload transunet --steps 30 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> $O/tu.err | grep "^{" > $O/tu.json
python -c "import json; d=json.load(open('$O/tu.json')); print('transunet', d['value'], d['ms_per_step'])"
python tools/bench_conv1x1_f32.py > $O/conv1x1_f32.txt 2>&1 || true; tail -12 $O/conv1x1_f32.txt
