"""One up-convolution on the fast-FIR kernel, a few launches (for rocprofv3 --pmc passes).  usage: bench_upfir_one.py cin cout h [batch] [fir: 1 | 0 = the 4-phase gather kernel on the same layer]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

cin, cout, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 32
use_fir = (int(sys.argv[5]) if len(sys.argv) > 5 else 1) != 0
dev = torch.device("cuda:0")
x = torch.randn(B, cin, h, h, device=dev)
w = torch.randn(1, cout, cin, 3, 3, device=dev)
s = 1 + 0.1 * torch.randn(B, cin, device=dev)
wpk, wsq = sis_hip.modconv_prepack(w)
fir = sis_hip.modconv_prepack_up_fir(w)
ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
for _ in range(6):
    sis_hip.modconv2d_up(x, wpk, s, ds, padded_rows=True, fir_u=fir if use_fir else None)
torch.cuda.synchronize()
