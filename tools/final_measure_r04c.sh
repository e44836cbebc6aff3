# Round 4 measurement, part 3: PMC passes of the up-convolution 64^2 -> 128^2 (512 -> 256 channels, B = 32) on the fast-FIR kernel and on
# the 4-phase gather kernel it replaced: executed MFMA operations (SQ_INSTS_VALU_MFMA_MOPS_F32), MFMA-busy, LDS, HBM-side bytes.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O
bash tools/pmc_upfir.sh r04z_up64_fir 512 256 64 32 1 > $O/pmc_up64_fir.txt 2>&1
bash tools/pmc_upfir.sh r04z_up64_gather 512 256 64 32 0 > $O/pmc_up64_gather.txt 2>&1
python3 tools/pmc_up64_summary.py $O/pmc_up64_fir.txt $O/pmc_up64_gather.txt > $O/pmc_up64.txt
cat $O/pmc_up64.txt
