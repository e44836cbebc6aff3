#!/bin/bash
# rocprofv3 kernel trace of the synthesis workload -> tools/step_timeline.py (one timed step).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof_syn -- python3 bench.py --workload synthesis --steps 20 --warmup 5 > $O/bench_rocprof2.log 2>&1
python3 tools/step_timeline.py $(find $O/prof_syn -name "*kernel_trace.csv" | head -1) > $O/step_timeline.txt
rm -rf $O/prof_syn
head -5 $O/step_timeline.txt
