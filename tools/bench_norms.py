"""Effective HBM bandwidth of the TransUNet norm / upsampling kernels (B=8, 512^2 input shapes), bf16 and fp32.
Bytes counted: forward = read x + write y; backward = read g + read x (twice: reduce + apply) + write dx."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch
import sis_hip

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"{'kernel':14s} {'shape':24s} {'dtype':6s} {'fwd ms':>8s} {'GB/s':>7s} {'bwd ms':>8s} {'GB/s':>7s}")
for dtype in (torch.bfloat16, torch.float32):
    es = 2 if dtype == torch.bfloat16 else 4
    for name, shape, groups in [("group_norm", (8, 64, 256, 256), 32), ("group_norm", (8, 256, 127, 127), 32),
                                ("group_norm", (8, 512, 64, 64), 32), ("group_norm", (8, 1024, 32, 32), 32),
                                ("batch_norm", (8, 16, 512, 512), 0), ("batch_norm", (8, 64, 256, 256), 0),
                                ("batch_norm", (8, 256, 64, 64), 0)]:
        x = torch.randn(*shape, device=dev).to(dtype)
        g = torch.randn(*shape, device=dev).to(dtype)
        c = shape[1]
        gamma, beta = torch.ones(c, device=dev), torch.zeros(c, device=dev)
        mb = x.numel() * es / 1e6
        if name == "group_norm":
            y, mean, rstd = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True)
            tf = timeit(lambda: sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True))
            tb = timeit(lambda: sis_hip.group_norm_bwd(g, x, mean, rstd, gamma, beta, groups, True))
        else:
            y, mean, rstd = sis_hip.batch_norm_train_fwd(x, gamma, beta, None, None, 1e-5, 0.1, True)
            tf = timeit(lambda: sis_hip.batch_norm_train_fwd(x, gamma, beta, None, None, 1e-5, 0.1, True))
            tb = timeit(lambda: sis_hip.batch_norm_train_bwd(g, x, mean, rstd, gamma, beta, True))
        print(f"{name:14s} {str(shape):24s} {str(dtype)[6:]:6s} {tf:8.3f} {2 * mb / tf:7.0f} {tb:8.3f} {4 * mb / tb:7.0f}")
    for shape in [(8, 64, 256, 256), (8, 256, 64, 64)]:
        x = torch.randn(*shape, device=dev).to(dtype)
        oh, ow = 2 * shape[2], 2 * shape[3]
        g = torch.randn(shape[0], shape[1], oh, ow, device=dev).to(dtype)
        mb = x.numel() * es / 1e6
        tf = timeit(lambda: sis_hip.upsample_bilinear(x, oh, ow))
        tb = timeit(lambda: sis_hip.upsample_bilinear(x, oh, ow, grad_output=g))
        print(f"{'upsample x2':14s} {str(shape):24s} {str(dtype)[6:]:6s} {tf:8.3f} {5 * mb / tf:7.0f} {tb:8.3f} {5 * mb / tb:7.0f}")
