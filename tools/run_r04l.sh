cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_generator_gpu.py -m gpu -q -x -k "fir" 2>&1 | tail -2
for A in 0 8 0; do echo "== SIS_UPFIR_ABL=$A"; SIS_UPFIR_ABL=$A timeout -k 10 120 python tools/bench_upfir.py 2>&1 | grep -v amdgpu.ids | sed 's/max rel diff.*//' | tee -a $O/abl$A.txt; done
