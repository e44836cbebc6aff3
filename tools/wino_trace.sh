#!/bin/bash
# Development tool: builds lib/libsis_hip_trace.so (the product library with -DSIS_WINO_TRACE in modconv_wino.hip).
# The trace build stamps the cycle counter four times per chunk per wave in 4 workgroups (tools/wino_trace.py reads it).
set -e
cd "$(dirname "$0")/../synthesis-in-style_amd/csrc"
make -j8 >/dev/null
mkdir -p _obj_trace
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize -DSIS_WINO_TRACE ${SIS_TRACE_DEFS} -c modconv_wino.hip -o _obj_trace/modconv_wino.o
objs=$(ls _obj/*.o | grep -v modconv_wino.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libsis_hip_trace${SIS_TRACE_SUFFIX}.so $objs _obj_trace/modconv_wino.o
echo built ../lib/libsis_hip_trace${SIS_TRACE_SUFFIX}.so
