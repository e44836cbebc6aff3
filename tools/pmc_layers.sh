#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over tools/bench_layers.py for one layer.
# usage: tools/pmc_layers.sh <tag> <layer e.g. conv64|up64> [batch]
set -e
TAG=$1; LAYER=$2; B=${3:-32}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_${TAG}_${LAYER}
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/bench_layers.py $B 3 $LAYER > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM FETCH_SIZE -- python3 tools/bench_layers.py $B 3 $LAYER > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- python3 tools/bench_layers.py $B 3 $LAYER > $OUT/p3.log 2>&1
find $OUT -name "*counter_collection.csv" | head
