"""One GEMM shape / tile of csrc/gemm_bf16.hip, a few launches (for rocprofv3 --pmc passes).
usage: bench_gemm_one.py layout(nt|nn|tn) m n k tile [splits]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip as S  # noqa: E402

layout, m, n, k, tile = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
splits = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=gen).bfloat16().to(dev)  # noqa: E731
if layout == "nt":
    a, b, lay, epi = r(m, k), r(n, k), S.GEMM_NT, S.EPI_NONE
elif layout == "nn":
    a, b, lay, epi = r(m, k), r(k, n), S.GEMM_NN, S.EPI_NONE
else:
    a, b, lay, epi = r(k, m), r(k, n), S.GEMM_TN, S.EPI_F32
for _ in range(10):
    S.gemm_bf16(a, b, lay, epi, splits=splits, tile=tile)
torch.cuda.synchronize()
