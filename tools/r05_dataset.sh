set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05r}; mkdir -p $O
for round in 1 2; do
for w in synthesis dataset; do
  python bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline 2> $O/err.txt | grep "^{" > $O/out.json
  python -c "import json; d=json.load(open('$O/out.json')); print('$w round $round', d['value'], d['ms_per_step'])" | tee -a $O/ab.txt
done; done
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --workload dataset --steps 8 --warmup 3 --no-cpu-baseline > $O/rocprof.log 2>&1
python tools/step_timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/dataset_timeline.txt
rm -rf $O/prof
grep -n "kmeans\|to_uint8\|make_image\|u8\|stream/queue\|step:" $O/dataset_timeline.txt | head -30
