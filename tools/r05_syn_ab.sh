# synthesis: parts of the last-resolution tail (SIS_RGB_TAIL_SPLIT), same box, alternating
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05n}; mkdir -p $O
python -m pytest tests/test_generator_gpu.py -x -q -m gpu -k "tail_split or batch32 or golden" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for round in 1 2; do
 for parts in 1 2 4; do
  SIS_RGB_TAIL_SPLIT=$parts python bench.py --workload synthesis --steps 40 --warmup 5 --no-cpu-baseline 2> $O/err.txt | grep "^{" > $O/out.json
  python -c "import json; d=json.load(open('$O/out.json')); print('parts $parts round $round', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
 done
done
