cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04w
timeout -k 10 200 python tools/bench_bilinear.py > gpurun_out/r04w/bilinear.txt 2>&1
cat gpurun_out/r04w/bilinear.txt
