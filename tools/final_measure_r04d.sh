# Round 4 measurement, part 4: PMC passes of EMANet's three fp32 matrix kernels: MFMA-busy, LDS, waits.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O
bash tools/pmc_conv1x1_f32.sh r04z_pw_fwd 512 2048 32 > $O/pmc_conv1x1_f32_fwd_512_2048.txt 2>&1
bash tools/pmc_conv1x1_f32.sh r04z_pw_dgrad 512 2048 32 dgrad > $O/pmc_conv1x1_f32_dgrad_512_2048.txt 2>&1
bash tools/pmc_wgrad.sh r04z_fc0 2048 512 32 16 5 > $O/pmc_conv_wgrad_wino_fc0.txt 2>&1
bash tools/pmc_wgrad.sh r04z_l3 256 256 32 16 5 > $O/pmc_conv_wgrad_wino_256.txt 2>&1
tail -14 $O/pmc_conv1x1_f32_fwd_512_2048.txt; tail -14 $O/pmc_conv_wgrad_wino_fc0.txt
