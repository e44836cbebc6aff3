# same-box A/B of two kernel libraries (lib/libsis_hip_<tag>.so from tools/build_variant.sh against the working library) on one workload
#   tools/ab_lib.sh <tag> <workload> [rounds]
w=${2:-emanet}
for i in $(seq ${3:-2}); do for lib in libsis_hip_$1.so libsis_hip.so; do
  s=$(SIS_HIP_LIB=$lib python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-dp-rehearsal 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; print(d['value'], d['ms_per_step'], r.get('frac'), r.get('avg_launch_ms'))")
  echo "$lib $w $s"
done; done
