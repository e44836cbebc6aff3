# Round 5 measurement, part 2 (PMC passes of the final library): per-kernel HBM traffic of all three workloads, the dominant
# synthesis kernel's matrix-pipe counters (refresh of the round-3 file), the fp32 1x1 kernel after its fragment pipelining.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r05; mkdir -p $O
bash tools/pmc_traffic.sh r05z synthesis emanet transunet > $O/pmc_traffic.log 2>&1 || tail -20 $O/pmc_traffic.log
ls -la gpurun_out/traffic_r05z.json && head -30 $O/pmc_traffic.log
bash tools/pmc_layers.sh r05z conv64 > $O/pmc_wino2_conv64.log 2>&1 && python3 tools/pmc_summary.py gpurun_out/pmc_r05z_conv64 > $O/pmc_wino2_conv64.txt 2>&1 || tail -5 $O/pmc_wino2_conv64.log
bash tools/pmc_conv1x1_f32.sh r05z_pw_fwd 512 2048 32 > $O/pmc_conv1x1_f32_fwd_512_2048.txt 2>&1 || true
tail -14 $O/pmc_conv1x1_f32_fwd_512_2048.txt
