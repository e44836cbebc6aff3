# data-parallel wrap: tests, then the rehearsal numbers of both training workloads
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05e}; mkdir -p $O
python -m pytest tests/test_distributed_gpu.py tests/test_step_graph_gpu.py -x -q -m gpu > $O/tests_dp.log 2>&1 || { tail -40 $O/tests_dp.log; exit 1; }
tail -2 $O/tests_dp.log
for w in emanet transunet; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline 2> $O/$w.err | grep "^{" > $O/$w.json || { tail -20 $O/$w.err; exit 1; }
  python -c "import json; d=json.load(open('$O/$w.json')); print('$w', d['value'], d['ms_per_step'], d['data_parallel_rehearsal'])"
done
