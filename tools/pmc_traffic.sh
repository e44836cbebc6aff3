#!/bin/bash
# HBM traffic of the bench command, per kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not
# fit one pass on gfx950), kernel trace only -- as MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots" prescribe.
# usage (on the GPU box): bash tools/pmc_traffic.sh <tag>      -> gpurun_out/traffic_<tag>.json
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/fetch --pmc FETCH_SIZE -- python3 bench.py --workload synthesis --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/write --pmc WRITE_SIZE -- python3 bench.py --workload synthesis --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
python3 tools/pmc_traffic.py $OUT gpurun_out/traffic_$TAG.json
