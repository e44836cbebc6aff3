set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_vit_block_gpu.py tests/test_conv_bf16_gpu.py tests/test_trans_u_net_gpu.py tests/test_step_graph_gpu.py -x -q -m gpu 2>&1 | tail -2
for round in 1 2; do for d in 0 1; do
 SIS_DEFER_REDUCES=$d python bench.py --workload transunet --steps 30 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('defer $d', d['value'], d['ms_per_step'])"
done; done
