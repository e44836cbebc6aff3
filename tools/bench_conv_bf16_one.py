"""One TransUNet layer on the bf16 MFMA convolution kernels, repeated (PMC passes: tools/pmc_conv_bf16.sh).
python tools/bench_conv_bf16_one.py <fwd|wgrad> cin cout h w k [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip  # noqa: E402

mode, cin, cout, h, w, k = sys.argv[1], *map(int, sys.argv[2:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda:0")
x = torch.randn(8, cin, h, w, device=dev).bfloat16()
if mode == "fwd":
    wt = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).bfloat16()
    packed = sis_hip.conv_bf16_pack(wt, h, w, 1)
    for _ in range(reps):
        sis_hip.conv_bf16(x, packed, cout, k, 1)
else:
    gy = torch.randn(8, cout, h, w, device=dev).bfloat16()
    for _ in range(reps):
        sis_hip.conv_bf16_wgrad(x, gy, torch.float32)
torch.cuda.synchronize()
print(sis_hip.lib().sis_last_kernel().decode())
