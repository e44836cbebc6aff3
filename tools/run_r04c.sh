set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
python -m pytest tests/test_distributed_gpu.py tests/test_ema_net_gpu.py tests/test_hip_conv_gpu.py "tests/test_trans_u_net_gpu.py::test_decoder_layers_bf16_gradients_vs_fp32_on_the_same_inputs" -m gpu -q > $O/new_tests.log 2>&1 || true
tail -30 $O/new_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench.json || (tail -30 $O/bench.err; exit 1)
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04c/bench.json'))
print('synth', d['value'], d['roofline']['frac'])
for k,v in d['seg_train'].items():
    print(k, v['images_per_s'], v['config'].get('hip_graph'), v['library_calls_per_step'], v['library_ms_per_step'], v['data_parallel_rehearsal'])
PY
