"""Development tool (GPU box): the 3x3 Winograd weight-gradient kernel run repeatedly on small and large shapes: every repeat must
be bit-identical to the first and close to the library's gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

dev = torch.device("cuda")
bad = 0
for (b, cin, cout, h, w) in [(8, 64, 64, 4, 4), (8, 128, 128, 8, 8), (4, 64, 128, 16, 16), (8, 64, 64, 32, 32), (2, 64, 64, 4, 8),
                             (16, 512, 512, 32, 32), (3, 64, 64, 64, 48), (16, 128, 64, 2, 2), (8, 256, 256, 4, 4), (5, 64, 64, 8, 8),
                             (8, 512, 512, 16, 16), (4, 512, 512, 32, 32), (8, 512, 512, 8, 8), (4, 256, 512, 16, 16), (6, 64, 64, 4, 4)]:
    if not sis_hip.conv3x3_wgrad_supported(b, cin, cout, h, w, min_work=0):
        print("unsupported", b, cin, cout, h, w)
        continue
    x = torch.randn(b, cin, h, w, device=dev)
    gy = torch.randn(b, cout, h, w, device=dev)
    wt = torch.zeros(cout, cin, 3, 3, device=dev)
    ref = torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
    first = sis_hip.conv3x3_wgrad(x, gy).clone()
    err = ((first - ref).abs().max() / ref.abs().max()).item()
    diff = 0
    for _ in range(30):
        junk = torch.randn(1 << 22, device=dev)  # perturb timing / cache state
        got = sis_hip.conv3x3_wgrad(x, gy)
        diff += int((got != first).sum().item())
        del junk
    print(f"{b}x{cin}->{cout} {h}x{w}: rel err {err:.2e}, differing elements over 30 repeats: {diff}")
    bad += diff > 0 or err > 1e-3
print("FAIL" if bad else "OK")
