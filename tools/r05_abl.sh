set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
SIS_HIP_LIB=libsis_hip_abl.so SIS_ABL_SKIP_REDUCES=1 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --workload transunet --steps 6 --warmup 3 --no-cpu-baseline --no-dp-rehearsal > $O/rocprof.log 2>&1
python tools/step_breakdown.py $O/prof 40 110 > $O/abl_breakdown.txt
rm -rf $O/prof
head -30 $O/abl_breakdown.txt | cut -c1-150
