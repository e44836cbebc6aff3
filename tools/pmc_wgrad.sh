#!/bin/bash
# PMC passes (separate runs, kernel-trace only) over tools/bench_wgrad_one.py.  usage: tools/pmc_wgrad.sh <tag> [shape args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_wgrad_${TAG}
mkdir -p $OUT
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/p1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -- python3 tools/bench_wgrad_one.py "$@" > $OUT/p1.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/p2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM FETCH_SIZE -- python3 tools/bench_wgrad_one.py "$@" > $OUT/p2.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/p3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -- python3 tools/bench_wgrad_one.py "$@" > $OUT/p3.log 2>&1
timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $OUT/p4 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_SALU -- python3 tools/bench_wgrad_one.py "$@" > $OUT/p4.log 2>&1
python3 tools/pmc_summary.py $OUT all > $OUT/summary.txt 2>&1
grep -A12 "conv_wgrad_wino" $OUT/summary.txt
