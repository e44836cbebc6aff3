set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_distributed_gpu.py tests/test_vit_block_gpu.py tests/test_gemm_bf16_gpu.py -m gpu -q -k "rccl or gemm256 or dropout or fused_block or forwards" > $O/tests.log 2>&1 || true
tail -12 $O/tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench.json || (tail -30 $O/bench.err; exit 1)
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04e/bench.json'))
print('synth', d['value'], d['roofline']['frac'])
for k,v in d['seg_train'].items():
    print(k, v['images_per_s'], v['config'].get('hip_graph'), v['library_calls_per_step'], v['library_ms_per_step'], v['data_parallel_rehearsal'])
PY
