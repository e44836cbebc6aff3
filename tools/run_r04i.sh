cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_generator_gpu.py tests/test_dataset_ops_gpu.py tests/test_determinism_gpu.py tests/test_gan_gpu.py tests/test_swagan_gpu.py -m gpu -q > $O/gen_tests.log 2>&1; tail -6 $O/gen_tests.log
python bench.py --workload synthesis --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench_syn.json; python -c "
import json; d=json.load(open('$O/bench_syn.json')); print('synth', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac']); [print('  ',k,v) for k,v in d['roofline']['kernels'].items()]"
SIS_UP_FIR=0 python bench.py --workload synthesis --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $O/bench_syn_nofir.json; python -c "
import json; d=json.load(open('$O/bench_syn_nofir.json')); print('synth without fir', d['value'], d['ms_per_step'])"
python bench.py --workload dataset --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | grep "^{" > $O/bench_dataset.json; python -c "
import json; print('dataset', json.load(open('$O/bench_dataset.json'))['value'])"
