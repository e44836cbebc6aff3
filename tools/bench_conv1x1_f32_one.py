"""One fp32 1x1 layer (csrc/conv1x1_f32.hip), forward or data gradient, a few launches: the target of tools/pmc_conv1x1_f32.sh.
usage: bench_conv1x1_f32_one.py cin cout side [dgrad]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip  # noqa: E402

cin, cout, side = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dgrad = len(sys.argv) > 4 and sys.argv[4] == "dgrad"
dev = torch.device("cuda:0")
x = torch.randn(16, cout if dgrad else cin, side, side, device=dev)
w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
for _ in range(5):
    y = sis_hip.conv1x1_f32(x, w, data_gradient=dgrad)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    y = sis_hip.conv1x1_f32(x, w, data_gradient=dgrad)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"{cin}->{cout} @{side}^2 {'dgrad' if dgrad else 'fwd'}: {ms * 1e3:.1f} us, {2.0 * 16 * cin * cout * side * side / ms / 1e9:.1f} TF")
