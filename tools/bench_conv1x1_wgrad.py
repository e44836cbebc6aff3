"""Development tool (GPU box): own fp32 1x1 weight-gradient kernel vs the batched library GEMM on EMANet-50's shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

B = 16
dev = torch.device("cuda")
shapes = [(64, 256, 64, 3), (256, 64, 64, 2), (256, 128, 64, 1), (128, 512, 32, 4), (512, 128, 32, 3), (256, 512, 64, 1), (512, 256, 32, 1),
          (256, 1024, 32, 6), (1024, 256, 32, 5), (512, 1024, 32, 1), (1024, 512, 32, 1), (512, 2048, 32, 3), (2048, 512, 32, 3), (512, 512, 32, 2)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tot_o = tot_l = 0.0
for cin, cout, hw, count in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev)
    gy = torch.randn(B, cout, hw, hw, device=dev)
    fl = 2.0 * B * cin * cout * hw * hw
    lib = timeit(lambda: torch.bmm(gy.view(B, cout, -1), x.view(B, cin, -1).transpose(1, 2)).sum(0))
    ours = timeit(lambda: sis_hip.conv1x1_wgrad_f32(gy, x)) if sis_hip.conv1x1_wgrad_f32_supported(gy, x) else float("nan")
    tot_o += ours * count
    tot_l += lib * count
    print(f"{cin:5d} -> {cout:5d} @{hw:3d} x{count}: library {lib * 1e3:7.1f} us {fl / lib / 1e9:6.1f} TF | ours {ours * 1e3:7.1f} us {fl / ours / 1e9:6.1f} TF", flush=True)
print(f"per step: library {tot_l:.3f} ms, ours {tot_o:.3f} ms")
