#!/bin/bash
# HBM traffic of the bench workloads, per kernel: two separate rocprofv3 --pmc passes per workload (FETCH_SIZE and WRITE_SIZE
# do not fit one pass on gfx950), kernel trace only -- as MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots" prescribe.
# The training steps run eagerly here (SIS_STEP_GRAPH=0): the same kernels, dispatched one by one.
# usage (on the GPU box): bash tools/pmc_traffic.sh <tag> [workloads...]      -> gpurun_out/traffic_<tag>.json
set -e
TAG=${1:-r01}
shift || true
WORKLOADS=${@:-synthesis emanet transunet}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic_$TAG
mkdir -p $OUT
export SIS_STEP_GRAPH=0
for W in $WORKLOADS; do
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$W/fetch --pmc FETCH_SIZE -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-dp-rehearsal > $OUT/$W.fetch.log 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$W/write --pmc WRITE_SIZE -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-dp-rehearsal > $OUT/$W.write.log 2>&1
  echo "pmc passes of $W done"
done
python3 tools/pmc_traffic.py $OUT gpurun_out/traffic_$TAG.json $WORKLOADS
