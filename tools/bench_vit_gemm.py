"""Development tool (GPU box): the library GEMMs of TransUNet's ViT-B/16 encoder at 512^2, B = 8 (8192 tokens), bf16:
forward / data gradient / weight gradient (fp32 or bf16 output) per Linear shape."""
import torch

dev = torch.device("cuda")
M = 8192
shapes = [("qkv", 768, 2304), ("proj", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot = {}
for name, k, n in shapes:
    x = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, device=dev, dtype=torch.bfloat16)
    g = torch.randn(M, n, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * n * k
    cases = {
        "fwd addmm": lambda: torch.addmm(b, x, w.t()),
        "dgrad mm": lambda: torch.mm(g, w),
        "wgrad f32 out": lambda: torch.mm(g.t(), x, out_dtype=torch.float32),
        "wgrad bf16 out": lambda: torch.mm(g.t(), x),
        "wgrad bf16 out + float()": lambda: torch.mm(g.t(), x).float(),
        "wgrad^T f32 out (x^T g)": lambda: torch.mm(x.t(), g, out_dtype=torch.float32),
    }
    for cname, fn in cases.items():
        ms = timeit(fn)
        tot[cname] = tot.get(cname, 0.0) + ms
        print(f"{name:5s} [{M} x {k}] -> {n:5d}  {cname:28s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF", flush=True)
print("sum over the four shapes (x12 layers per step):", {k: round(v * 12, 3) for k, v in tot.items()}, "ms")
