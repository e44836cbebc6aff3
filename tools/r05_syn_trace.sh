set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05l}; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --workload synthesis --steps 10 --warmup 3 --no-cpu-baseline > $O/rocprof.log 2>&1
python tools/step_timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) > $O/step_timeline.txt
rm -rf $O/prof
head -70 $O/step_timeline.txt
