cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for P in 1 3 1 3; do echo "== SIS_UPFIR_PIPE=$P"; SIS_UPFIR_PIPE=$P timeout -k 10 120 python tools/bench_upfir.py 2>&1 | grep -v amdgpu.ids | tee -a $O/bench_upfir_pipe$P.txt; done
for P in 3; do SIS_UPFIR_PIPE=$P timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -m gpu -q -x -k "fir" 2>&1 | tail -2; done
