for lib in libsis_hip_old.so libsis_hip.so libsis_hip_b29.so libsis_hip_b31.so libsis_hip_old.so libsis_hip.so; do
  export SIS_HIP_LIB=$lib
  s=$(python bench.py --workload synthesis --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])")
  e=$(python bench.py --workload emanet --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "$lib synth $s ema $e"
done
