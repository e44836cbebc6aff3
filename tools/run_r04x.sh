cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04x
timeout -k 10 600 python -m pytest tests/test_upsample_gpu.py -q -m gpu > gpurun_out/r04x/tests.log 2>&1
tail -12 gpurun_out/r04x/tests.log
timeout -k 10 200 python tools/bench_bilinear.py > gpurun_out/r04x/bilinear.txt 2>&1
cat gpurun_out/r04x/bilinear.txt
SIS_UP2_DIRECT=0 timeout -k 10 200 python tools/bench_bilinear.py > gpurun_out/r04x/bilinear_tiled.txt 2>&1
tail -2 gpurun_out/r04x/bilinear_tiled.txt
