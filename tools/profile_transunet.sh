set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-tu_prof}; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof_tu -- python3 bench.py --workload transunet --steps 6 --warmup 3 --no-cpu-baseline > $O/tu_rocprof.log 2>&1
python tools/step_breakdown.py $O/prof_tu 90 130 > $O/transunet_step_breakdown.txt
rm -rf $O/prof_tu
head -60 $O/transunet_step_breakdown.txt
