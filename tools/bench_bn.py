"""Effective HBM bandwidth of the fused batch-norm kernels on EMANet-50 activation shapes (B=16)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch
import sis_hip

dev = torch.device("cuda")
shapes = [(16, 64, 128, 128), (16, 128, 128, 128), (16, 64, 64, 64), (16, 256, 64, 64), (16, 128, 32, 32), (16, 512, 32, 32),
          (16, 256, 32, 32), (16, 1024, 32, 32), (16, 2048, 32, 32)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"{'shape':24s} {'MB':>6s} | stats ms GB/s | fwd ms GB/s | fwd+res ms GB/s | bwd ms GB/s | bwd+res ms GB/s")
for shp in shapes:
    x = torch.randn(*shp, device=dev)
    res = torch.randn(*shp, device=dev)
    dy = torch.randn(*shp, device=dev)
    c = shp[1]
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    mb = x.numel() * 4 / 1e6
    mean, invstd = sis_hip.bn_stats(x, rm, rv, 1e-5, 3e-4)
    y = sis_hip.bn_act_fwd(x, res, mean, invstd, g, b, True)
    t_s = timeit(lambda: sis_hip.bn_stats(x, rm, rv, 1e-5, 3e-4))
    t_f = timeit(lambda: sis_hip.bn_act_fwd(x, None, mean, invstd, g, b, True))
    t_fr = timeit(lambda: sis_hip.bn_act_fwd(x, res, mean, invstd, g, b, True))
    t_b = timeit(lambda: sis_hip.bn_act_bwd(dy, y, x, mean, invstd, g, True, False))
    t_br = timeit(lambda: sis_hip.bn_act_bwd(dy, y, x, mean, invstd, g, True, True))
    gb = lambda passes, t: passes * mb / t  # MB/ms = GB/s
    print(f"{str(shp):24s} {mb:6.1f} | {t_s:6.3f} {gb(1, t_s):5.0f} | {t_f:6.3f} {gb(2, t_f):5.0f} | {t_fr:6.3f} {gb(3, t_fr):5.0f} | "
          f"{t_b:6.3f} {gb(7, t_b):5.0f} | {t_br:6.3f} {gb(8, t_br):5.0f}")
