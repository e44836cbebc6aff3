"""The encoder's weight-gradient GEMMs at 8 192 tokens: twelve separate split-K launches (+ their slab reductions) against one
pointer-table launch per Linear shape (csrc/gemm_bf16.hip, sis_gemm_bf16_wgrad_bias_multi), per tile code."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip  # noqa: E402
from networks.trans_u_net.vit_encoder import _wgrad_plan  # noqa: E402

dev = torch.device("cuda:0")
T, L = 8192, 12


def timeit(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot_a = 0.0
tot_b = {}
for name, out_f, in_f in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    grads = [torch.randn(T, out_f, device=dev).bfloat16() for _ in range(L)]
    xs = [torch.randn(T, in_f, device=dev).bfloat16() for _ in range(L)]
    dws = [torch.empty(out_f, in_f, device=dev) for _ in range(L)]
    dbs = [torch.empty(out_f, device=dev) for _ in range(L)]
    splits, tile = _wgrad_plan(out_f, in_f)

    def separate():
        for g, x, dw in zip(grads, xs, dws):
            sis_hip.gemm_bf16_wgrad_bias(g, x, splits, tile, dw=dw)

    ref = [sis_hip.gemm_bf16_wgrad_bias(g, x, splits, tile) for g, x in zip(grads[:2], xs[:2])]
    t_a = timeit(separate)
    tot_a += t_a
    fl = 2.0 * T * out_f * in_f * L
    line = f"{name:5s} {out_f:4d}x{in_f:4d}: separate (splits {splits}, tile {tile}) {t_a:7.3f} ms = {fl / t_a / 1e9:6.1f} TF |"
    for mt in (0, 4, 5, 6):
        jobs = list(zip(grads, xs, dws, dbs))
        t_b = timeit(lambda: sis_hip.gemm_bf16_wgrad_bias_multi(jobs, tile=mt))
        tot_b[mt] = tot_b.get(mt, 0.0) + t_b
        err = max(((dws[i] - ref[i][0]).abs().max() / ref[i][0].abs().max()).item() for i in range(2))
        errb = max(((dbs[i] - ref[i][1]).abs().max() / ref[i][1].abs().max()).item() for i in range(2))
        line += f" multi tile {mt}: {t_b:6.3f} ms = {fl / t_b / 1e9:6.1f} TF (dw {err:.1e} db {errb:.1e}) |"
    print(line, flush=True)
print(f"sum: separate {tot_a:.3f} ms; multi " + ", ".join(f"tile {k}: {v:.3f} ms" for k, v in tot_b.items()))
