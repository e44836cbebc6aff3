"""Development tool (GPU box): per-wave timeline of the transposed-convolution kernel's chunk loop (build with
tools/build_variant.sh WORK v2trace -DSIS_V2_TRACE).  usage: python tools/v2_trace.py [h=64] [cin=512] [cout=256] [batch=32]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

sis_hip.LIB_PATH = os.path.join(ROOT, "synthesis-in-style_amd", "lib", "libsis_hip_v2trace.so")
h = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cin = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cout = int(sys.argv[3]) if len(sys.argv) > 3 else 256
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0")
x = torch.randn(B, cin, h, h, device=dev)
w = torch.randn(1, cout, cin, 3, 3, device=dev)
s = 1 + 0.1 * torch.randn(B, cin, device=dev)
wpk, wsq = sis_hip.modconv_prepack(w)
ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
f = lambda: sis_hip.modconv2d_up(x, wpk, s, ds)  # noqa: E731
for _ in range(10):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{ms:.3f} ms per launch, {2.0 * B * cout * cin * 9 * h * h / ms / 1e9:.1f} TF")
L = sis_hip.lib()
L.sis_v2_trace_read.argtypes = [ctypes.c_void_p]
buf = np.zeros((4, 8, 64, 4), dtype=np.uint32)
assert L.sis_v2_trace_read(buf.ctypes.data) == 0
nch = min(64, cin // 8)
for g in range(2):
    t = buf[g, :, :nch].astype(np.int64)
    t = (t - t[:, 0, 0].min()) & 0xFFFFFFFF
    per_chunk = np.diff(t[:, :, 0], axis=1)
    print(f"workgroup {g}: cycles per chunk by wave:", np.median(per_chunk[:, 4:nch - 1], axis=1).astype(int))
    for wv in range(8):
        dma = np.median(t[wv, 4:nch - 1, 1] - t[wv, 4:nch - 1, 0])
        mf = np.median(t[wv, 4:nch - 1, 2] - t[wv, 4:nch - 1, 1])
        bw = np.median(t[wv, 5:nch - 1, 0] - t[wv, 4:nch - 2, 2])
        print(f"  wave {wv}: DMA issue {int(dma):5d}  loads+MFMAs {int(mf):5d}  barrier wait {int(bw):5d}")
