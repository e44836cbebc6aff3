"""Times the bf16 GEMM (csrc/gemm_bf16.hip) on the twelve GEMM shapes of one ViT-B/16 block at 8 192 tokens (configs[4]) for
every tile / split choice, next to torch's library GEMM on the same operands (same process, interleaved rounds)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip as S  # noqa: E402

dev = torch.device("cuda:0")
M = int(os.environ.get("TOKENS", "8192"))
gen = torch.Generator().manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape, generator=gen).bfloat16().to(dev)


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


seed = torch.tensor([1], dtype=torch.int64, device=dev)
rows = []
if os.environ.get("PLAN") == "1":
    # the choices networks/trans_u_net/vit_encoder.py ships (forward 128 x 96 where the width divides by 96, data gradient
    # 128 x 128, weight gradient by _wgrad_plan), one line per Linear
    sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
    from networks.trans_u_net.vit_encoder import _wgrad_plan
    total = 0.0
    for name, n, k in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
        x, w, b = rnd(M, k), rnd(n, k), torch.randn(n, device=dev)
        g, resid, pre = rnd(M, n), torch.randn(M, n, device=dev), rnd(M, k)
        flops = 2.0 * M * n * k
        t96 = 8 if n % 96 == 0 and name != "fc1" else 0
        if name == "qkv":
            f = lambda: S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS, bias=b, tile=t96)
        elif name == "fc1":
            f = lambda: S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=b, seed=seed, site=1, drop_p=0.1, tile=t96)
        else:
            f = lambda: S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=b, resid=resid, seed=seed, site=1, drop_p=0.1, tile=t96)
        if name == "fc2":
            d = lambda: S.gemm_bf16(g, w, S.GEMM_NN, S.EPI_GELU_BWD, pre=pre, seed=seed, site=1, drop_p=0.1)
        else:
            d = lambda: S.gemm_bf16(g, w, S.GEMM_NN, S.EPI_NONE)
        splits, tile = _wgrad_plan(n, k)
        wg = lambda: S.gemm_bf16(g, x, S.GEMM_TN, S.EPI_F32, splits=splits, tile=tile)
        t_f, t_d, t_w = timed(f), timed(d), timed(wg)
        l_f = timed(lambda: torch.addmm(b.bfloat16(), x, w.t()))
        l_d = timed(lambda: torch.mm(g, w))
        l_w = timed(lambda: torch.mm(g.t(), x, out_dtype=torch.float32))
        total += t_f + t_d + t_w
        print(f"{name:5s} fwd tile {t96}: {t_f*1e3:6.1f} us {flops/t_f/1e9:5.0f} TF (library bare GEMM {l_f*1e3:6.1f})   "
              f"dgrad: {t_d*1e3:6.1f} us {flops/t_d/1e9:5.0f} TF ({l_d*1e3:6.1f})   "
              f"wgrad tile {tile} x{splits}: {t_w*1e3:6.1f} us {flops/t_w/1e9:5.0f} TF ({l_w*1e3:6.1f})", flush=True)
    x, w, b = rnd(M, 768), rnd(3072, 768), torch.randn(3072, device=dev)
    t = timed(lambda: S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=b, seed=seed, site=1, drop_p=0.1, tile=8))
    print(f"(fc1 forward on the 128 x 96 tile instead: {t*1e3:.1f} us)")
    print(f"one block, 12 GEMMs: {total*1e3:.1f} us; x12 blocks = {total*12:.2f} ms per step")
    sys.exit(0)
for name, n, k in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    x, w, b = rnd(M, k), rnd(n, k), torch.randn(n, device=dev)
    g = rnd(M, n)
    resid = torch.randn(M, n, device=dev)
    pre = rnd(M, k)
    flops = 2.0 * M * n * k
    lib_fwd = timed(lambda: torch.addmm(b.bfloat16(), x, w.t()))
    lib_dx = timed(lambda: torch.mm(g, w))
    lib_dw = timed(lambda: torch.mm(g.t(), x, out_dtype=torch.float32))
    for tile in (0, 4):
        epi = {"qkv": S.EPI_BIAS, "fc1": S.EPI_BIAS_GELU_DROP}.get(name, S.EPI_BIAS_DROP_RESID)
        kw = dict(bias=b, tile=tile)
        if epi == S.EPI_BIAS_DROP_RESID:
            kw.update(resid=resid, seed=seed, site=1, drop_p=0.1)
        if epi == S.EPI_BIAS_GELU_DROP:
            kw.update(seed=seed, site=1, drop_p=0.1)
        t_f = timed(lambda: S.gemm_bf16(x, w, S.GEMM_NT, epi, **kw))
        t_plain = timed(lambda: S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS, bias=b, tile=tile))
        if tile == 8:
            print(f"{name:5s} tile {tile}: fwd {t_f*1e3:7.1f} us {flops/t_f/1e9:7.0f} TF (bias only {t_plain*1e3:7.1f} us)", flush=True)
            continue
        if name == "fc2":
            t_dx = timed(lambda: S.gemm_bf16(g, w, S.GEMM_NN, S.EPI_GELU_BWD, pre=pre, seed=seed, site=1, drop_p=0.1, tile=tile))
        else:
            t_dx = timed(lambda: S.gemm_bf16(g, w, S.GEMM_NN, S.EPI_NONE, tile=tile))
        line = f"{name:5s} tile {tile}: fwd {t_f*1e3:7.1f} us {flops/t_f/1e9:7.0f} TF (bias only {t_plain*1e3:7.1f} us; lib {lib_fwd*1e3:7.1f})  " \
               f"dx {t_dx*1e3:7.1f} us {flops/t_dx/1e9:7.0f} TF (lib {lib_dx*1e3:7.1f})  dw"
        for splits in (1, 2, 4, 8):
            t_dw = timed(lambda: S.gemm_bf16(g, x, S.GEMM_TN, S.EPI_F32, splits=splits, tile=tile))
            line += f" s{splits}:{t_dw*1e3:6.1f}"
        line += f" us (lib {lib_dw*1e3:7.1f})"
        print(line, flush=True)
