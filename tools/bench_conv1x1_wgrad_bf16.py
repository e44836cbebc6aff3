"""Times the bf16 1x1 weight gradient (csrc/conv_bf16_wgrad.hip, conv1x1_wgrad_bf16_kernel) on the 1x1 layers of TransUNet's
ResNetV2 trunk at batch 8 / 512^2 next to the library formulation it replaces (bmm over the images + sum + cast)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip as S  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, rounds=5, iters=20):
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


tot_own = tot_lib = 0.0
for cin, cout, hw, count in ((64, 64, 127, 1), (64, 256, 127, 4), (256, 64, 127, 2), (256, 128, 127, 1), (128, 512, 64, 4), (512, 128, 64, 3),
                             (256, 512, 64, 1), (512, 256, 64, 1), (256, 1024, 32, 9), (1024, 256, 32, 8), (512, 1024, 32, 1)):
    x = torch.randn(8, cin, hw, hw, device=dev).bfloat16()
    g = torch.randn(8, cout, hw, hw, device=dev).bfloat16()
    own = timed(lambda: S.conv1x1_bf16_wgrad(x, g, torch.bfloat16))
    lib = timed(lambda: torch.bmm(g.view(8, cout, hw * hw), x.view(8, cin, hw * hw).transpose(1, 2)).sum(0, dtype=torch.float32).bfloat16())
    mb = (x.numel() + g.numel()) * 2 / 1e6
    tot_own += own * count; tot_lib += lib * count
    print(f"{cin:5d}->{cout:5d} @{hw:3d}^2 x{count}: own {own*1e3:7.1f} us ({mb/own/1e3:5.2f} TB/s of unique operand bytes)   library {lib*1e3:7.1f} us", flush=True)
print(f"per step: own {tot_own:.3f} ms, library {tot_lib:.3f} ms")
