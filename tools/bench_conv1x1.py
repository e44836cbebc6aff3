"""EMANet 1x1 convolutions (B=16): ATen convolution (MIOpen) vs plain batched GEMMs on the NCHW tensors, per direction."""
import time
import torch
import torch.nn.functional as F

B = 16
dev = torch.device("cuda")
shapes = [(128, 64, 64), (64, 256, 64), (256, 64, 64), (256, 128, 64), (128, 512, 32), (512, 128, 32), (512, 256, 32),
          (256, 1024, 32), (1024, 256, 32), (1024, 512, 32), (512, 2048, 32), (2048, 512, 32), (512, 512, 32)]
counts = [1, 4, 2, 1, 4, 3, 1, 7, 5, 1, 3, 3, 2]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tot = [0.0] * 6
print(f"{'shape':22s} | fwd aten / mm | dgrad aten / mm | wgrad aten / bmm+sum")
for (cin, cout, hw), cnt in zip(shapes, counts):
    x = torch.randn(B, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev)
    gy = torch.randn(B, cout, hw, hw, device=dev)
    w2 = w.view(cout, cin)
    cb = lambda mask: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, mask)
    t = [timeit(lambda: F.conv2d(x, w)), timeit(lambda: torch.matmul(w2, x.view(B, cin, -1))),
         timeit(lambda: cb((True, False, False))), timeit(lambda: torch.matmul(w2.t(), gy.view(B, cout, -1))),
         timeit(lambda: cb((False, True, False))),
         timeit(lambda: torch.bmm(gy.view(B, cout, -1), x.view(B, cin, -1).transpose(1, 2)).sum(0))]
    ref = cb((False, True, False))[1].view(cout, cin)
    got = torch.bmm(gy.view(B, cout, -1), x.view(B, cin, -1).transpose(1, 2)).sum(0)
    assert (ref - got).abs().max() < 1e-3 * ref.abs().max()
    for i in range(6):
        tot[i] += t[i] * cnt
    print(f"{cin:4d}->{cout:4d} @{hw:3d} x{cnt}    | {t[0]:.3f} / {t[1]:.3f} | {t[2]:.3f} / {t[3]:.3f} | {t[4]:.3f} / {t[5]:.3f}")
print("per step (ms): fwd %.2f / %.2f, dgrad %.2f / %.2f, wgrad %.2f / %.2f" % tuple(tot))
