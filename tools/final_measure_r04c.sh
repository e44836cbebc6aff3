# Round 4 measurement, part 3: PMC passes of the up-convolution 64^2 -> 128^2 (512 -> 256 channels, B = 32) on the fast-FIR kernel and on
# the 4-phase gather kernel it replaced: executed MFMA operations (SQ_INSTS_VALU_MFMA_MOPS_F32), MFMA-busy, LDS, HBM-side bytes.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final_r04; mkdir -p $O
bash tools/pmc_upfir.sh r04z_up64_fir 512 256 64 32 1 > $O/pmc_up64_fir.txt 2>&1
bash tools/pmc_upfir.sh r04z_up64_gather 512 256 64 32 0 > $O/pmc_up64_gather.txt 2>&1
python3 - <<'PY' > gpurun_out/final_r04/pmc_up64.txt
import re
def blocks(path):
    out, cur = {}, None
    for line in open(path):
        if line.startswith("gpurun_out/"):
            cur = line.strip(); out[cur] = []
        elif cur is not None and line.startswith("    "):
            out[cur].append(line.rstrip())
    return out
for title, path, key in (("fast-FIR kernel (shipped)", "gpurun_out/final_r04/pmc_up64_fir.txt", "modconv_upfir_kernel"),
                         ("4-phase gather kernel (SIS_UP_FIR=0)", "gpurun_out/final_r04/pmc_up64_gather.txt", "modconv_v2_kernel")):
    print("==", title, "-- up-convolution 512 -> 256 channels, 64^2 -> 128^2, B = 32; counters are sums over one launch")
    for head, rows in blocks(path).items():
        if key in head:
            print(head); print("\n".join(rows))
PY
cat $O/pmc_up64.txt
