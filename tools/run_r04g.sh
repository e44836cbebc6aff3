cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/full_gpu_tests.log 2>&1
tail -8 $O/full_gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench.json || (tail -30 $O/bench.err; exit 1)
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04g/bench.json'))
print('synth', d['value'], d['roofline']['frac'])
for k,v in d['seg_train'].items():
    print(k, v['images_per_s'], v['config'].get('hip_graph'), v['library_calls_per_step']['fallback'], v['library_ms_per_step'], v['data_parallel_rehearsal'])
PY
