p='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])'
echo "synthesis-only: $(python bench.py --workload synthesis --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "$p")"
echo "all:            $(python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep '^{' | python -c "$p")"
echo "all no-dp:      $(python bench.py --gpus 1 --steps 20 --warmup 5 --no-dp-rehearsal 2>/dev/null | grep '^{' | python -c "$p")"
echo "synthesis-only: $(python bench.py --workload synthesis --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "$p")"
