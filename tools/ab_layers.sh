# per-layer same-box A/B of two kernel libraries (isolated layers, back to back): tools/ab_layers.sh <tag> [reps]
for i in 1 2; do for lib in libsis_hip_$1.so libsis_hip.so; do
  echo "== $lib"; SIS_HIP_LIB=$lib python tools/bench_layers.py 32 ${2:-5} 2>/dev/null | grep -E "@ *(32|64|128|256) " 
done; done
