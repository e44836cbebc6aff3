#!/bin/bash
# L2-side traffic of the generator's kernels by request size, with hit / miss counts (separate rocprofv3 --pmc passes, kernel
# trace only, as MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots" prescribe).  usage (GPU box): bash tools/pmc_traffic_detail.sh <tag>
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_traffic_detail_$TAG
mkdir -p $OUT
CMD="python3 bench.py --workload synthesis --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc FETCH_SIZE -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc WRITE_SIZE -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p4 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum -- $CMD > $OUT/p4.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p5 --pmc TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum -- $CMD > $OUT/p5.log 2>&1
python3 tools/pmc_traffic_detail.py $OUT > gpurun_out/traffic_detail_$TAG.txt
cat gpurun_out/traffic_detail_$TAG.txt
