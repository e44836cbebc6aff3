# Round 5 measurement, part 1: driver-style bench line, rocprofv3 kernel stats of the synthesis command, step breakdowns and
# per-launch listings of both training steps (plain and behind the data-parallel wrap), dataset loop.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r05; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench.json
python -c "import json; s=open('$O/bench.json').read(); d=json.loads(s); print(len(s), 'bytes; synth', d['value'], d['roofline']['frac'], {k: (v['images_per_s'], v['roofline']['frac'], v['dp_rehearsal_graph_ms_per_step']) for k, v in d['seg_train'].items()})"
cp gpurun_out/bench_detail_all.json $O/bench_detail_all.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_syn -- python3 bench.py --workload synthesis --steps 20 --warmup 5 > $O/bench_rocprof.log 2>&1
grep "^{" $O/bench_rocprof.log > $O/bench_under_rocprof.json
cp $(find $O/prof_syn -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/step_timeline.py $(find $O/prof_syn -name "*kernel_trace.csv" | head -1) > $O/step_timeline.txt
rm -rf $O/prof_syn
for w in emanet transunet; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline --no-dp-rehearsal > $O/${w}_rocprof.log 2>&1
python tools/step_breakdown.py $O/prof_$w 120 130 > $O/${w}_step_breakdown.txt
python tools/step_launches.py $O/prof_$w > $O/${w}_step_launches.txt
cp $(find $O/prof_$w -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
rm -rf $O/prof_$w
head -3 $O/${w}_step_breakdown.txt
done
python bench.py --workload dataset --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | grep "^{" > $O/bench_dataset.json
python -c "import json; print('dataset', json.load(open('$O/bench_dataset.json'))['value'])"
