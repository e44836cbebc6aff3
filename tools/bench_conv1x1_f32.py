"""fp32 pointwise convolution (csrc/conv1x1_f32.hip) vs the library on EMANet-50's 1x1 shapes at 256^2, B = 16."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip  # noqa: E402

B = 16
LAYERS = [("l1 64->256", 64, 256, 64), ("l1 256->64", 256, 64, 64), ("l2 256->128", 256, 128, 64), ("l2 128->512", 128, 512, 32),
          ("l2 512->128", 512, 128, 32), ("l3 512->256", 512, 256, 32), ("l3 256->1024", 256, 1024, 32), ("l3 1024->256", 1024, 256, 32),
          ("l4 1024->512", 1024, 512, 32), ("l4 512->2048", 512, 2048, 32), ("l4 2048->512", 2048, 512, 32), ("fc0 2048->512", 2048, 512, 32)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dev = torch.device("cuda:0")
print(f"{'layer':16s} {'GF':>6s} | fwd mine / lib ms (TF/s) | dgrad mine / lib ms")
tm = tl = 0.0
for name, cin, cout, s in LAYERS:
    x = torch.randn(B, cin, s, s, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
    gy = torch.randn(B, cout, s, s, device=dev)
    gf = 2.0 * B * cin * cout * s * s / 1e9
    a = timeit(lambda: sis_hip.conv1x1_f32(x, w))
    b = timeit(lambda: F.conv2d(x, w))
    c = timeit(lambda: sis_hip.conv1x1_f32(gy, w, data_gradient=True))
    d = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False)))
    tm += a + c
    tl += b + d
    print(f"{name:16s} {gf:6.1f} | {a:.3f} ({gf / a:5.1f}) / {b:.3f} ({gf / b:5.1f}) | {c:.3f} ({gf / c:5.1f}) / {d:.3f} ({gf / d:5.1f})", flush=True)
print(f"total mine {tm:.2f} ms, library {tl:.2f} ms")
