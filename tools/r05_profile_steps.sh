# per-launch listing + per-kernel breakdown of both training steps (rocprofv3 kernel trace): gpurun_out/<tag>/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05_steps}; mkdir -p $O
for w in ${2:-transunet emanet}; do
rocprofv3 --kernel-trace --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline --no-dp-rehearsal > $O/${w}_rocprof.log 2>&1
python tools/step_breakdown.py $O/prof_$w 120 130 > $O/${w}_step_breakdown.txt
python tools/step_launches.py $O/prof_$w > $O/${w}_step_launches.txt
rm -rf $O/prof_$w
head -3 $O/${w}_step_breakdown.txt
done
