cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v
for w in synthesis emanet transunet; do
  SIS_OCC=1 timeout -k 10 300 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-dp-rehearsal > /dev/null 2> gpurun_out/r04v/occ_$w.err
  echo "== $w" >> gpurun_out/r04v/occupancy.txt
  grep "^\[occ\]" gpurun_out/r04v/occ_$w.err | sort | uniq -c | sort -rn >> gpurun_out/r04v/occupancy.txt
done
cat gpurun_out/r04v/occupancy.txt
