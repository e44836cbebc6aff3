cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
SIS_DP_DEBUG=1 timeout -k 10 300 python -m pytest tests/test_distributed_gpu.py -m gpu -q -s -k "rccl" > $O/tests.log 2>&1
grep -n "grad_exchange\] bucket 0\|passed\|failed\|what()" $O/tests.log | head -40
for i in 1 2; do timeout -k 10 300 python -m pytest tests/test_distributed_gpu.py -m gpu -q -k "rccl" > $O/tests_$i.log 2>&1; tail -2 $O/tests_$i.log; done
