# same-box A/B of the fast-FIR chunk loop forms (SIS_UPFIR_PIPE): parity tests under the candidate, then synthesis rates alternating
set -e
SIS_UPFIR_PIPE=${1:-4} timeout -k 10 600 python -m pytest tests/test_generator_gpu.py -x -q -m gpu -k "upfir or up or generator" > gpurun_out/upfir_pipe_tests.log 2>&1 || { tail -20 gpurun_out/upfir_pipe_tests.log; exit 1; }
tail -2 gpurun_out/upfir_pipe_tests.log
for pipe in 1 ${1:-4} 1 ${1:-4}; do
  s=$(SIS_UPFIR_PIPE=$pipe python bench.py --workload synthesis --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "pipe $pipe synth $s"
done
