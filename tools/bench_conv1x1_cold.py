"""1x1 convolution forward / data gradient: ATen convolution vs one strided-batched GEMM with the weight expanded over
the batch, on ROTATING buffers (16 copies, > the 256 MB Infinity Cache for the big shapes) so nothing is cache-resident."""
import time
import torch
import torch.nn.functional as F

B, dev, R = 16, torch.device("cuda"), 12
shapes = [(64, 256, 64), (256, 64, 64), (128, 512, 32), (512, 128, 32), (256, 1024, 32), (1024, 256, 32), (512, 2048, 32),
          (2048, 512, 32)]
counts = [4, 2, 4, 3, 7, 5, 3, 3]


def timeit(fn, n=3 * R):
    for i in range(R):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i % R)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tot = [0.0] * 4
for (cin, cout, hw), cnt in zip(shapes, counts):
    xs = [torch.randn(B, cin, hw, hw, device=dev) for _ in range(R)]
    gs = [torch.randn(B, cout, hw, hw, device=dev) for _ in range(R)]
    w = torch.randn(cout, cin, 1, 1, device=dev)
    w2 = w.view(cout, cin)
    t_fa = timeit(lambda i: F.conv2d(xs[i], w))
    t_fb = timeit(lambda i: torch.bmm(w2.unsqueeze(0).expand(B, cout, cin).contiguous(), xs[i].view(B, cin, -1)))
    t_da = timeit(lambda i: torch.ops.aten.convolution_backward(gs[i], xs[i], w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                                (True, False, False)))
    t_db = timeit(lambda i: torch.bmm(w2.t().unsqueeze(0).expand(B, cin, cout).contiguous(), gs[i].view(B, cout, -1)))
    for k, t in enumerate((t_fa, t_fb, t_da, t_db)):
        tot[k] += t * cnt
    print(f"{cin:4d}->{cout:4d} @{hw:3d} x{cnt}: fwd aten {t_fa:.3f} bmm {t_fb:.3f} | dgrad aten {t_da:.3f} bmm {t_db:.3f}")
print("per step: fwd aten %.2f bmm %.2f | dgrad aten %.2f bmm %.2f" % tuple(tot))
