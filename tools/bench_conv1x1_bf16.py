"""bf16 1x1 convolutions of TransUNet's trunk at B = 8: the convolution kernel (csrc/conv_bf16.hip) vs the batched GEMM
(csrc/gemm_bf16.hip via sis_gemm_bf16_batched), forward / data gradient / weight gradient (library bmm for the latter)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip as S  # noqa: E402

dev = torch.device("cuda:0")
B = 8


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
    return best * 1e3


for cin, cout, side in ((512, 128, 64), (128, 512, 64), (512, 256, 64), (1024, 256, 32), (256, 1024, 32), (1024, 768, 32)):
    hw = side * side
    x = torch.randn(B, cin, side, side, device=dev).bfloat16()
    w = (torch.randn(cout, cin, 1, 1, device=dev) * cin ** -0.5).bfloat16()
    gy = torch.randn(B, cout, side, side, device=dev).bfloat16()
    pk = S.conv_bf16_pack(w, side, side, 1)
    pa = S.conv_bf16_pack(w, side, side, 1, adjoint=True)
    t_cf = timed(lambda: S.conv_bf16(x, pk, cout, 1, 1))
    t_cd = timed(lambda: S.conv_bf16(gy, pa, cin, 1, 1))
    t_cw = timed(lambda: torch.bmm(gy.view(B, cout, hw), x.view(B, cin, hw).transpose(1, 2)).sum(0, dtype=torch.float32))
    w2, x3, g3 = w.view(cout, cin), x.view(B, cin, hw), gy.view(B, cout, hw)
    fl = 2.0 * B * cin * cout * hw
    line = f"{cin:5d}->{cout:5d} @{side}^2: conv fwd {t_cf:6.1f} dgrad {t_cd:6.1f} bmm-wgrad {t_cw:6.1f} us |"
    for tile in (0, 4):
        t_f = timed(lambda: S.gemm_bf16_batched(w2, x3, S.GEMM_NN, tile=tile))
        t_d = timed(lambda: S.gemm_bf16_batched(w2, g3, S.GEMM_TN, tile=tile))
        t_w = timed(lambda: S.gemm_bf16_batched(g3, x3, S.GEMM_NT, S.EPI_F32, sum_over_batches=True, tile=tile))
        line += f" gemm tile {tile}: fwd {t_f:6.1f} ({fl / t_f / 1e6:4.0f} TF) dgrad {t_d:6.1f} wgrad {t_w:6.1f} |"
    print(line, flush=True)
