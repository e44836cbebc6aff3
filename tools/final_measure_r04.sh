# Round 4 measurement, part 1: driver-style bench, rocprofv3 kernel stats of the same command, step breakdowns, side workloads.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench.json
python -c "import json; d=json.load(open('$O/bench.json')); print('synth', d['value'], d['roofline']['frac'], 'ema', d['seg_train']['emanet']['images_per_s'], 'tu', d['seg_train']['transunet_bf16']['images_per_s'], 'tu32', d['seg_train']['transunet_f32']['images_per_s'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_syn -- python3 bench.py --workload synthesis --steps 20 --warmup 5 > $O/bench_rocprof.log 2>&1
grep "^{" $O/bench_rocprof.log > $O/bench_under_rocprof.json
cp $(find $O/prof_syn -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python tools/step_timeline.py $(find $O/prof_syn -name "*kernel_trace.csv" | head -1) > $O/step_timeline.txt
rm -rf $O/prof_syn
rocprofv3 --kernel-trace --output-format csv -d $O/prof_ema -- python3 bench.py --workload emanet --steps 6 --warmup 3 --no-cpu-baseline --no-dp-rehearsal > $O/ema_rocprof.log 2>&1
python tools/step_breakdown.py $O/prof_ema 70 130 > $O/emanet_step_breakdown.txt
rm -rf $O/prof_ema
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_tu -- python3 bench.py --workload transunet --steps 6 --warmup 3 --no-cpu-baseline --no-dp-rehearsal > $O/tu_rocprof.log 2>&1
python tools/step_breakdown.py $O/prof_tu 90 130 > $O/transunet_step_breakdown.txt
cp $(find $O/prof_tu -name "*kernel_stats.csv" | head -1) $O/transunet_kernel_stats.csv
rm -rf $O/prof_tu
head -3 $O/emanet_step_breakdown.txt; head -3 $O/transunet_step_breakdown.txt; head -4 $O/step_timeline.txt
python bench.py --workload dataset --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | grep "^{" > $O/bench_dataset.json
python -c "import json; print('dataset', json.load(open('$O/bench_dataset.json'))['value'])"
SIS_GEMM256_TILES=288,192,96 python tools/bench_gemm256.py > $O/gemm256_vs_128.txt 2>&1
tail -9 $O/gemm256_vs_128.txt
