cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
for A in 0 1 2 4 8 15 0; do echo "== SIS_UPFIR_ABL=$A"; SIS_UPFIR_ABL=$A timeout -k 10 120 python tools/bench_upfir.py 2>&1 | grep -v amdgpu.ids | sed 's/max rel diff.*//' | tee -a $O/abl$A.txt; done
