cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04z
timeout -k 10 300 python tools/bench_norms.py > gpurun_out/r04z/norms.txt 2>&1
cat gpurun_out/r04z/norms.txt
timeout -k 10 300 python tools/bench_ln.py > gpurun_out/r04z/ln.txt 2>&1
tail -12 gpurun_out/r04z/ln.txt
