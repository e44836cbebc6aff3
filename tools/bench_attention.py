"""Times the fused attention kernels (csrc/attention_bf16.hip) at configs[4] size next to torch's scaled_dot_product_attention
(forward and backward) on the same data."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip as S  # noqa: E402

dev = torch.device("cuda:0")
B, N, H = 8, 1024, 12
gen = torch.Generator().manual_seed(0)
qkv = torch.randn(B, N, 3 * H * 64, generator=gen).bfloat16().to(dev)
d_ctx = torch.randn(B, N, H * 64, generator=gen).bfloat16().to(dev)


def timed(fn, rounds=5, iters=10):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters)
    return best


ctx, lse = S.attention_fwd(qkv, H)
t_f = timed(lambda: S.attention_fwd(qkv, H))
t_b = timed(lambda: S.attention_bwd(d_ctx, qkv, ctx, lse, H))
fl = 4.0 * B * H * N * N * 64
print(f"fused fwd {t_f*1e3:7.1f} us {fl/t_f/1e9:6.0f} TF   bwd {t_b*1e3:7.1f} us {2.5*fl/t_b/1e9:6.0f} TF (5-product count)")
q, k, v = (t.view(B, N, H, 64).permute(0, 2, 1, 3) for t in qkv.split(H * 64, dim=-1))
q, k, v = (t.detach().requires_grad_() for t in (q, k, v))
t_lf = timed(lambda: F.scaled_dot_product_attention(q, k, v))
o = F.scaled_dot_product_attention(q, k, v)
g = d_ctx.view(B, N, H, 64).permute(0, 2, 1, 3)
t_lb = timed(lambda: torch.autograd.grad(o, (q, k, v), g, retain_graph=True))
print(f"library fwd {t_lf*1e3:7.1f} us   bwd {t_lb*1e3:7.1f} us")
