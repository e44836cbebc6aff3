"""Development tool (GPU box): modulated / plain 3x3 convolutions and the transposed convolution run repeatedly on small and large
shapes with the allocator's reuse pattern perturbed: every repeat must be bit-identical to the first."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402
from networks.hip_conv import conv3x3  # noqa: E402

dev = torch.device("cuda")
bad = 0


def repeat(tag, fn, n=20):
    global bad
    first = fn().clone()
    diff = 0
    for i in range(n):
        junk = torch.full((1 << (18 + i % 5),), float("nan"), device=dev)  # freed blocks hold NaNs: a read of unwritten memory shows
        got = fn()
        diff += int((got != first).sum().item()) + int(torch.isnan(got).sum().item())
        del junk
    print(f"{tag}: differing / NaN elements over {n} repeats: {diff}")
    bad += diff > 0


for (b, cin, cout, h) in [(5, 64, 64, 16), (4, 512, 512, 4), (4, 512, 512, 8), (8, 512, 512, 16), (8, 512, 512, 32), (4, 256, 128, 64), (3, 128, 64, 32),
                          (32, 128, 128, 256)]:
    x = torch.randn(b, cin, h, h, device=dev)
    w = torch.randn(1, cout, cin, 3, 3, device=dev)
    s = 1 + 0.1 * torch.randn(b, cin, device=dev)
    wpk, wsq = sis_hip.modconv_prepack(w)
    ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
    noise = torch.randn(1, 1, h, h, device=dev)
    nw = torch.full((1,), 0.1, device=dev)
    bias = torch.randn(cout, device=dev)
    u = sis_hip.modconv_prepack_wino(w)
    repeat(f"modconv fused {b}x{cin}->{cout} @{h}", lambda: sis_hip.modconv2d(x, wpk, s, ds, 3, noise, nw, bias, fuse_act=True, wino_u=u))
    repeat(f"modconv plain {b}x{cin}->{cout} @{h}", lambda: sis_hip.modconv2d(x, wpk, s, ds, 3, wino_u=u))
    if h <= 128:
        repeat(f"modconv up    {b}x{cin}->{cout} @{h}", lambda: sis_hip.modconv2d_up(x, wpk, s, ds, padded_rows=True)[..., :2 * h + 1])
    if h <= 64:
        wt = torch.randn(cout, cin, 3, 3, device=dev, requires_grad=True)
        xg = x.clone().requires_grad_(True)
        gy = torch.randn(b, cout, h, h, device=dev)

        def fwd_bwd():
            xg.grad = None
            wt.grad = None
            y = conv3x3(xg, wt, 1)
            y.backward(gy)
            return y, xg.grad, wt.grad
        for part, name in enumerate(("y", "dx", "dw (library below the policy threshold: atomics allowed there)")):
            repeat(f"conv3x3 {name} {b}x{cin}->{cout} @{h}", lambda: fwd_bwd()[part], 10)
print("FAIL" if bad else "OK")
