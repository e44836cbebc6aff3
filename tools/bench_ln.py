import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch, torch.nn.functional as F, sis_hip
dev = torch.device("cuda")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
x = torch.randn(8, 1024, 768, device=dev); g = torch.randn(768, device=dev); b = torch.randn(768, device=dev)
gy = torch.randn(8, 1024, 768, device=dev); gyb = gy.bfloat16()
y, mean, rstd = sis_hip.layer_norm_fwd(x, g, b, 1e-6, torch.bfloat16)
print("ours fwd f32->bf16 %.4f ms" % timeit(lambda: sis_hip.layer_norm_fwd(x, g, b, 1e-6, torch.bfloat16)))
print("ours fwd f32->f32  %.4f ms" % timeit(lambda: sis_hip.layer_norm_fwd(x, g, b, 1e-6)))
print("aten fwd f32 + cast %.4f ms" % timeit(lambda: F.layer_norm(x, (768,), g, b, 1e-6).bfloat16()))
print("ours bwd (bf16 g)  %.4f ms" % timeit(lambda: sis_hip.layer_norm_bwd(gyb, x, mean, rstd, g)))
print("ours bwd (f32 g)   %.4f ms" % timeit(lambda: sis_hip.layer_norm_bwd(gy, x, mean, rstd, g)))
xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
out = F.layer_norm(xr, (768,), gr, br, 1e-6)
print("aten bwd           %.4f ms" % timeit(lambda: torch.autograd.grad(out, (xr, gr, br), gy, retain_graph=True)))
