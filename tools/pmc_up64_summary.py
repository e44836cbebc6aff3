"""Side-by-side PMC summary of the up-convolution 64^2 -> 128^2 on the fast-FIR kernel and on the 4-phase gather kernel
(inputs: the two outputs of tools/pmc_upfir.sh).  usage: pmc_up64_summary.py <fir.txt> <gather.txt>"""
import sys


def blocks(path):
    out, cur, seen = [], None, set()
    for line in open(path):
        if line.startswith("gpurun_out/"):
            cur = [line.strip()]; out.append(cur); seen = set()
        elif cur is not None and line.startswith("    "):
            name = line.split()[0]
            if name in seen:      # rows of a kernel whose header line was filtered out
                cur = None
            else:
                seen.add(name); cur.append(line.rstrip())
    return out


vals = {}
f, g = "modconv_upfir_kernel", "modconv_v2_kernel"
for title, path, key in (("fast-FIR kernel (shipped)", sys.argv[1], f), ("4-phase gather kernel (SIS_UP_FIR=0)", sys.argv[2], g)):
    print("==", title, "-- up-convolution 512 -> 256 channels, 64^2 -> 128^2, B = 32; counters are sums over one launch (rocprofv3 --pmc, three passes)")
    for b in blocks(path):
        if key in b[0]:
            print("\n".join(b))
            for r in b[1:]:
                n, v = r.split(); vals[(key, n)] = float(v)
            vals[(key, "us")] = float(b[0].split("avg_us=")[1])
mops = "SQ_INSTS_VALU_MFMA_MOPS_F32"
print("\n== summary")
print(f"executed fp32 MFMA operations ({mops}): {vals[(f, mops)]:.4g} vs {vals[(g, mops)]:.4g} = x{vals[(f, mops)] / vals[(g, mops)]:.3f} (25 / 36 = 0.694 plus the tiles' padding)")
print(f"launch time under the counters: {vals[(f, 'us')]:.0f} us vs {vals[(g, 'us')]:.0f} us = x{vals[(g, 'us')] / vals[(f, 'us')]:.3f} faster")
for k in (f, g):
    print(f"{k}: MFMA-busy {vals[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')]:.4g} cycles over {vals[(k, 'GRBM_GUI_ACTIVE')]:.4g} GRBM cycles; LDS bank-conflict cycles / LDS active cycles = "
          f"{vals[(k, 'SQ_LDS_BANK_CONFLICT')] / vals[(k, 'SQ_LDS_IDX_ACTIVE')]:.3f}; HBM-side bytes = (FETCH_SIZE x 2 + WRITE_SIZE) x 1 KB = "
          f"{(vals[(k, 'FETCH_SIZE')] * 2 + vals[(k, 'WRITE_SIZE')]) * 1024 / 1e9:.2f} GB (gfx950: FETCH_SIZE tallies 128-byte requests as 64; algorithmic: 0.27 GB input x 4 output-channel blocks + 0.56 GB output)")
