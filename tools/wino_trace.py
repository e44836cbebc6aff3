"""Development tool (GPU box): per-wave timeline of modconv_wino2_kernel's chunk loop from the trace build
(tools/wino_trace.sh).  usage: python tools/wino_trace.py [h=64] [cin=512] [cout=512] [batch=32]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

sis_hip.LIB_PATH = os.path.join(ROOT, "synthesis-in-style_amd", "lib", "libsis_hip_trace%s.so" % os.environ.get("SIS_TRACE_SUFFIX", ""))
h = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 512
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0")
x = torch.randn(B, cin, h, h, device=dev)
w = torch.randn(1, cout, cin, 3, 3, device=dev)
s = 1 + 0.1 * torch.randn(B, cin, device=dev)
wpk, wsq = sis_hip.modconv_prepack(w)
ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
noise = torch.randn(1, 1, h, h, device=dev)
nw = torch.full((1,), 0.1, device=dev)
bias = torch.zeros(cout, device=dev)
u = sis_hip.modconv_prepack_wino(w)
f = lambda: sis_hip.modconv2d(x, wpk, s, ds, 3, noise, nw, bias, fuse_act=True, wino_u=u)  # noqa: E731
for _ in range(20):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{ms:.3f} ms per launch, {2.0 * B * cout * cin * 9 * h * h / ms / 1e9:.1f} TF algorithmic")
L = sis_hip.lib()
L.sis_wino_trace_read.argtypes = [ctypes.c_void_p]
buf = np.zeros((4, 8, 64, 4), dtype=np.uint32)
rc = L.sis_wino_trace_read(buf.ctypes.data)
assert rc == 0, rc
nch = min(64, cin // 8)
for g in range(4):
    t = buf[g, :, :nch].astype(np.int64)
    t0 = t[:, 0, 0].min()
    t = (t - t0) & 0xFFFFFFFF
    per_chunk = np.diff(t[:, :, 0], axis=1)            # [wave][chunk] loop-top to loop-top
    print(f"workgroup {g}: cycles per chunk (median over chunks 4..{nch - 2}) by wave:", np.median(per_chunk[:, 4:nch - 1], axis=1).astype(int))
    seg = np.stack([t[:, :, 1] - t[:, :, 0], t[:, :, 2] - t[:, :, 1], t[:, :, 3] - t[:, :, 2]], axis=-1)  # first operand reads, the 32 slots, tail
    wait = t[:, 1:, 0] - t[:, :-1, 3]                 # barrier wait (incl. vmcnt(0))
    for wv in range(8):
        m = np.median(seg[wv, 4:nch - 1], axis=0).astype(int)
        print(f"  wave {wv}: top to first MFMA {m[0]:5d}  32 MFMA slots {m[1]:5d}  tail {m[2]:5d}  barrier wait {int(np.median(wait[wv, 4:nch - 2])):5d}")
    if g == 0:
        print("  chunk 10 raw (wave x stamp), relative to the earliest stamp of the chunk:")
        r = t[:, 10, :] - t[:, 10, :].min()
        for wv in range(8):
            print("   ", wv, r[wv].tolist(), " next top:", int(t[wv, 11, 0] - t[:, 10, :].min()))

tb = np.zeros((4, 8, 16, 8), dtype=np.uint32)
L.sis_wino_trace_tile_read.argtypes = [ctypes.c_void_p]
assert L.sis_wino_trace_tile_read(tb.ctypes.data) == 0
t = tb[0].astype(np.int64)
print("tile-level (workgroup 0), cycles, wave 0 / wave 7: chunk loop | next tile's setup + first DMA issue | reduce + LDS send | barrier | "
      "finalise + stores | barrier (DMA landed) | tail store + first transform + barrier")
order = [0, 1, 3, 4, 5, 6, 7, 2]  # stamp slots in program order
for kk in range(16):
    if t[0, kk, 1] == 0 or t[0, kk, 2] == 0:
        break
    segs = []
    for a, b in zip(order[:-1], order[1:]):
        segs.append(f"{(t[0, kk, b] - t[0, kk, a]) & 0xFFFFFFFF}/{(t[7, kk, b] - t[7, kk, a]) & 0xFFFFFFFF}")
    print(f"  tile {kk}: " + " | ".join(segs))
