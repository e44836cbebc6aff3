#!/bin/bash
# PMC passes (separate runs, kernel-trace only) of the fast-FIR up-convolution.  usage: tools/pmc_upfir.sh <tag> cin cout h
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_${TAG}
mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/bench_upfir_one.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM FETCH_SIZE -- python3 tools/bench_upfir_one.py "$@" > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU -- python3 tools/bench_upfir_one.py "$@" > $OUT/p3.log 2>&1
python3 tools/pmc_summary.py $OUT all | grep -v "at::native"
