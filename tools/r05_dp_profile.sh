# kernel trace of the data-parallel rehearsal's eager leg (direct RCCL calls, world size 1): what the wrap adds to a step
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05f}; mkdir -p $O
for w in ${2:-transunet emanet}; do
SIS_BENCH_DP_MODES=eager rocprofv3 --kernel-trace --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --steps 6 --warmup 3 --no-cpu-baseline > $O/${w}_rocprof.log 2>&1
python tools/step_breakdown.py $O/prof_$w 200 150 > $O/${w}_dp_step_breakdown.txt
rm -rf $O/prof_$w
head -2 $O/${w}_dp_step_breakdown.txt; grep -i "nccl\|rccl\|foreach\|multi_tensor\|copy" $O/${w}_dp_step_breakdown.txt | head
done
