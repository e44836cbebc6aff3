# Round 5 baseline: driver-style bench line, per-launch listings of both training steps.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench.err | grep "^{" > $O/bench.json
python -c "import json; d=json.load(open('$O/bench.json')); print(len(open('$O/bench.json').read()), 'bytes; synth', d['value'], d['roofline']['frac'], {k: (v['images_per_s'], v['dp_rehearsal_graph_ms_per_step']) for k, v in d['seg_train'].items()})"
cp gpurun_out/bench_detail_all.json $O/ || true
python tools/layer_times.py transunet > $O/transunet_per_launch.txt 2>&1
python tools/layer_times.py emanet > $O/emanet_per_launch.txt 2>&1
head -40 $O/transunet_per_launch.txt
