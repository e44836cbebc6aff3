"""Development tool (GPU box): a few launches of the 3x3 Winograd weight-gradient kernel on one shape, for rocprofv3 --pmc passes
(tools/pmc_wgrad.sh).  usage: python tools/bench_wgrad_one.py [cin=2048] [cout=512] [hw=32] [batch=16] [launches=5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

cin, cout, hw, B, n = [int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 2048), (2, 512), (3, 32), (4, 16), (5, 5))]
dev = torch.device("cuda")
x = torch.randn(B, cin, hw, hw, device=dev)
gy = torch.randn(B, cout, hw, hw, device=dev)
for _ in range(n):
    sis_hip.conv3x3_wgrad(x, gy)
torch.cuda.synchronize()
print("done")
