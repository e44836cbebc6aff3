# same-box A/B of two library builds (lib/libsis_hip_base.so = HEAD, lib/libsis_hip.so = working tree): alternating runs
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r05k}; mkdir -p $O
W=${2:-emanet}
for round in 1 2; do
 for lib in libsis_hip_base.so libsis_hip.so; do
  SIS_HIP_LIB=$lib python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2> $O/err.txt | grep "^{" > $O/out.json
  python -c "import json; d=json.load(open('$O/out.json')); print('$W $lib round $round', d['value'], d['ms_per_step'])" | tee -a $O/ab_$W.txt
 done
done
