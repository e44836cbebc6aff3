cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/ds; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload dataset --steps 20 --warmup 5 > $O/log.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
head -25 $O/kernel_stats.csv | cut -c1-150
