# Final measurement of round 3, part 2 (PMC passes of the final library): generator layers, traffic, GEMM, fp32 1x1.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r03; mkdir -p $O
bash tools/pmc_layers.sh r03z conv64 > $O/pmc_conv64.log 2>&1
bash tools/pmc_layers.sh r03z up64 > $O/pmc_up64.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_r03z_conv64 all | grep -v "at::native" > $O/pmc_wino2_conv64.txt
python3 tools/pmc_summary.py gpurun_out/pmc_r03z_up64 all | grep -v "at::native" > $O/pmc_up64.txt
bash tools/pmc_conv1x1_f32.sh r03z_c1fwd 512 2048 32 > $O/pmc_conv1x1_f32_fwd.txt 2>&1
bash tools/pmc_gemm.sh r03z_qkv nt 8192 2304 768 8 > $O/pmc_gemm_qkv_fwd.txt 2>&1
bash tools/pmc_traffic.sh r03z > $O/pmc_traffic.log 2>&1 || true
ls gpurun_out | grep -i "traffic_r03z" || true
tail -3 $O/pmc_wino2_conv64.txt
