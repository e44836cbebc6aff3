"""Cycle anatomy of the 256-row GEMM tile (development build with -DG256_TRACE: tools/build_variant.sh WORK trace -DG256_TRACE
-fno-slp-vectorize; run with SIS_HIP_LIB=libsis_hip_trace.so).  Stamps of wave 0 per workgroup: kernel entry, first operands
landed (prologue), main loop done, stores retired.  usage: trace_gemm256.py n k [epilogue: bias|gelu|none]"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip as S  # noqa: E402

n, k = int(sys.argv[1]), int(sys.argv[2])
epi = sys.argv[3] if len(sys.argv) > 3 else "gelu"
M = 8192
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
a = (torch.randn(M, k, generator=gen)).bfloat16().to(dev)
w = (torch.randn(n, k, generator=gen) * k ** -0.5).bfloat16().to(dev)
bias = torch.randn(n, device=dev)
seed = S.dropout_seed(dev)
kw = {"gelu": dict(epilogue=S.EPI_BIAS_GELU_DROP, bias=bias, seed=seed, site=2, drop_p=0.1), "bias": dict(epilogue=S.EPI_BIAS, bias=bias),
      "none": dict(epilogue=S.EPI_NONE)}[epi]
tile = S.gemm_tile_256(M, n, k)
assert tile is not None
lib = S.lib()
lib.sis_gemm256_set_trace.argtypes = [ctypes.c_void_p]
lib.sis_gemm256_set_trace.restype = None
trace = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
for _ in range(20):
    S.gemm_bf16(a, w, S.GEMM_NT, tile=tile, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    S.gemm_bf16(a, w, S.GEMM_NT, tile=tile, **kw)
e1.record()
torch.cuda.synchronize()
print(f"untraced pointer (stamps skipped): {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, tile code {tile}")
lib.sis_gemm256_set_trace(trace.data_ptr())
S.gemm_bf16(a, w, S.GEMM_NT, tile=tile, **kw)
torch.cuda.synchronize()
trace.zero_()
e0.record()
S.gemm_bf16(a, w, S.GEMM_NT, tile=tile, **kw)
e1.record()
torch.cuda.synchronize()
lib.sis_gemm256_set_trace(None)
t = trace.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
print(f"traced launch: {e0.elapsed_time(e1) * 1e3:.1f} us, {len(t)} workgroups")
s0, s1, s2, s3 = (t[:, i].astype(np.float64) for i in range(4))
xcc = (t[:, 4] >> 32) & 0xF
base = s0.min()
def q(x):
    return f"median {np.median(x):9.0f}  p10 {np.percentile(x, 10):9.0f}  p90 {np.percentile(x, 90):9.0f}"
print("(cycles of s_memtime; per-XCD clocks are not synchronised: 'start' only within an XCD)")
print("prologue (entry -> first stage landed)   ", q(s1 - s0))
print("main loop                                ", q(s2 - s1))
print("epilogue (-> stores retired)             ", q(s3 - s2))
print("whole workgroup                          ", q(s3 - s0))
for x in range(8):
    m = xcc == x
    if m.any():
        st = s0[m] - s0[m].min()
        en = s3[m] - s0[m].min()
        print(f"xcc {x}: {m.sum():4d} workgroups, starts at {np.sort(st)[:4].astype(int)} ... {np.sort(st)[-4:].astype(int)}, last end {en.max():.0f}")
