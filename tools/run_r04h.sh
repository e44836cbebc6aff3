cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_generator_gpu.py -m gpu -q -x -k "fir" > $O/fir_tests.log 2>&1; tail -25 $O/fir_tests.log
timeout -k 10 300 python tools/bench_upfir.py > $O/bench_upfir.txt 2>&1; cat $O/bench_upfir.txt | tail -6
