# Round 4 measurement, part 2 (PMC passes of the final library): the new GEMM tile, per-kernel HBM traffic of all three workloads.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O
bash tools/pmc_gemm.sh r04z_qkv256 nt 8192 2304 768 11 > $O/pmc_gemm_qkv_fwd_256x288.txt 2>&1
bash tools/pmc_gemm.sh r04z_qkv128 nt 8192 2304 768 8 > $O/pmc_gemm_qkv_fwd_128x96.txt 2>&1
bash tools/pmc_gemm.sh r04z_fc1 nt 8192 3072 768 10 > $O/pmc_gemm_fc1_fwd_256x192.txt 2>&1
tail -12 $O/pmc_gemm_qkv_fwd_256x288.txt
bash tools/pmc_traffic.sh r04z synthesis emanet transunet > $O/pmc_traffic.log 2>&1 || tail -20 $O/pmc_traffic.log
ls -la gpurun_out/traffic_r04z.json && head -40 $O/pmc_traffic.log
