"""Forward / data-gradient GEMM shapes of a ViT-B/16 block at 8 192 tokens: the 256-row tiles (csrc/gemm256_bf16.hip) against
the 128-wide tiles (csrc/gemm_bf16.hip) with the same epilogues, interleaved rounds in one process, random operands."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip as S  # noqa: E402

dev = torch.device("cuda:0")
M = int(os.environ.get("TOKENS", "8192"))
gen = torch.Generator().manual_seed(0)


def r(*shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).bfloat16().to(dev)


seed = S.dropout_seed(dev)
cases = []   # (name, n, k, epilogue kwargs factory, old tile, old layout is NN?)
for name, n, k, epi in (("qkv fwd", 2304, 768, "bias3"), ("proj fwd", 768, 768, "resid"), ("fc1 fwd", 3072, 768, "gelu"),
                        ("fc2 fwd", 768, 3072, "resid"), ("qkv dgrad", 768, 2304, "none"), ("proj dgrad", 768, 768, "none"),
                        ("fc1 dgrad", 768, 3072, "none"), ("fc2 dgrad", 3072, 768, "gelu_bwd")):
    a, w = r(M, k), r(n, k, scale=k ** -0.5)
    wt = w.t().contiguous()   # [k, n]: the NN operand of the 128-wide data-gradient path
    bias, resid, pre = torch.randn(n, device=dev), torch.randn(M, n, device=dev), r(M, n)
    kw = {"bias3": dict(epilogue=S.EPI_BIAS, bias=(bias[:n // 3].contiguous(), bias[n // 3:2 * n // 3].contiguous(), bias[2 * n // 3:].contiguous())),
          "resid": dict(epilogue=S.EPI_BIAS_DROP_RESID, bias=bias, resid=resid, seed=seed, site=1, drop_p=0.1),
          "gelu": dict(epilogue=S.EPI_BIAS_GELU_DROP, bias=bias, seed=seed, site=2, drop_p=0.1),
          "none": dict(epilogue=S.EPI_NONE), "gelu_bwd": dict(epilogue=S.EPI_GELU_BWD, pre=pre, seed=seed, site=2, drop_p=0.1)}[epi]
    dgrad = "dgrad" in name
    old_tile = 0 if (dgrad or epi == "gelu") else 8
    new_tile = S.gemm_tile_256(M, n, k)
    old = (lambda a=a, w=w, wt=wt, kw=kw, t=old_tile, d=dgrad: S.gemm_bf16(a, wt if d else w, S.GEMM_NN if d else S.GEMM_NT, tile=t, **kw))
    new = (lambda a=a, w=w, kw=kw, t=new_tile: S.gemm_bf16(a, w, S.GEMM_NT, tile=t, **kw)) if new_tile is not None else None
    cases.append((name, n, k, old, new, new_tile))


def timeit(fn, iters=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, n, k, old, new, tile in cases:
    for fn in (old, new):
        if fn is not None:
            for _ in range(3):
                fn()
torch.cuda.synchronize()
tot_old = tot_new = 0.0
for name, n, k, old, new, tile in cases:
    ts_old, ts_new = [], []
    for _ in range(5):
        ts_old.append(timeit(old))
        if new is not None:
            ts_new.append(timeit(new))
    flops = 2.0 * M * n * k
    o = sorted(ts_old)[2]
    nn = sorted(ts_new)[2] if ts_new else float("nan")
    tot_old += o
    tot_new += nn if ts_new else o
    print(f"{name:12s} n={n:5d} k={k:5d} tile {tile}: 128-wide {o:7.1f} us = {flops / o / 1e6:7.0f} TF   256-row {nn:7.1f} us = {flops / nn / 1e6:7.0f} TF   x{o / nn:.2f}")
print(f"sum per block: {tot_old:.1f} -> {tot_new:.1f} us")
