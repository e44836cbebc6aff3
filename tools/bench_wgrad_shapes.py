"""ATen (MIOpen) weight-gradient and data-gradient time for EMANet-50's 3x3 convolutions (B=16)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip
from networks.hip_conv import _space_to_batch

if os.environ.get("SIS_HIP_LIB"):  # same-box A/B of kernel builds (tools/build_variant.sh)
    sis_hip.LIB_PATH = os.path.join(ROOT, "synthesis-in-style_amd", "lib", os.environ["SIS_HIP_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ABLATION = bool(os.environ.get("SIS_ABLATION"))  # timing of an ablation build: no ATen columns, no correctness check
dev = torch.device("cuda")
shapes = [(2048, 512, 32, 1, 1), (512, 256, 32, 1, 1), (64, 64, 128, 1, 1), (64, 128, 128, 1, 1), (64, 64, 64, 1, 3),
          (128, 128, 32, 1, 3), (256, 256, 32, 1, 1), (256, 256, 32, 2, 5), (512, 512, 32, 2, 1), (512, 512, 32, 8, 1),
          (512, 512, 32, 16, 1)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tot_w = tot_d = tot_o = 0.0
print(f"{'shape':30s} {'count':>5s} {'wgrad ms':>9s} {'TF':>7s} {'dgrad ms':>9s} {'TF':>7s}")
for cin, cout, hw, dil, count in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev)
    gy = torch.randn(B, cout, hw, hw, device=dev)
    flops = 2.0 * B * cout * cin * 9 * hw * hw
    tw = td = float("nan")
    if not ABLATION:
        tw = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (dil, dil), (dil, dil), False, (0, 0), 1,
                                                                (False, True, False)))
        td = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (dil, dil), (dil, dil), False, (0, 0), 1,
                                                                (True, False, False)))
    ours = float("nan")
    if sis_hip.conv3x3_wgrad_supported(B * dil * dil, cin, cout, hw // dil, hw // dil, min_work=0):  # (capability only: the policy threshold is what this tool is for)
        ref = torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (dil, dil), (dil, dil), False, (0, 0), 1, (False, True, False))[1]
        got = sis_hip.conv3x3_wgrad(_space_to_batch(x, dil), _space_to_batch(gy, dil))
        err = ((got - ref).abs().max() / ref.abs().max()).item()
        assert ABLATION or err < 5e-4, err
        xs, gs = _space_to_batch(x, dil), _space_to_batch(gy, dil)
        ours = timeit(lambda: sis_hip.conv3x3_wgrad(xs, gs))  # (the kernel alone: the sub-image copies of a dilated layer are made once per step)
        tot_o += ours * count
    else:
        tot_o += tw * count
    tot_w += tw * count
    tot_d += td * count
    print(f"3x3 {cin:4d}->{cout:4d} @{hw:3d} d{dil:<2d}      {count:5d} {tw:9.3f} {flops/tw/1e9:7.1f} {td:9.3f} {flops/td/1e9:7.1f}   ours wgrad {ours:7.3f} ms {flops/ours/1e9:7.1f} TF")
print(f"per step: wgrad ATen {tot_w:.2f} ms, wgrad ours(+ATen where ineligible) {tot_o:.2f} ms, dgrad (ATen) {tot_d:.2f} ms")
