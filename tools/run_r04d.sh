set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gemm_bf16_gpu.py -m gpu -q -k "gemm256 or transpose_bank or dropout" > $O/gemm256_tests.log 2>&1 || tail -40 $O/gemm256_tests.log
tail -3 $O/gemm256_tests.log
SIS_GEMM256_TILES=288,192,96 timeout -k 10 300 python tools/bench_gemm256.py > $O/bench_gemm256.txt 2>&1 || (tail -20 $O/bench_gemm256.txt; exit 1)
cat $O/bench_gemm256.txt
timeout -k 10 900 python -m pytest tests/test_vit_block_gpu.py tests/test_ema_net_gpu.py tests/test_distributed_gpu.py tests/test_trans_u_net_gpu.py -m gpu -q > $O/tests2.log 2>&1 || true
tail -15 $O/tests2.log
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench.json || (tail -30 $O/bench.err; exit 1)
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04d/bench.json'))
print('synth', d['value'], d['roofline']['frac'])
for k,v in d['seg_train'].items():
    print(k, v['images_per_s'], v['config'].get('hip_graph'), v['library_calls_per_step'], v['library_ms_per_step'], v['data_parallel_rehearsal'])
PY
