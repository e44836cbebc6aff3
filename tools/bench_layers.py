"""Per-layer timing of the generator's convolution kernels (development tool, GPU box).
usage: python tools/bench_layers.py [batch] [reps]   -> one line per layer: ms, TFLOP/s"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

if os.environ.get("SIS_HIP_LIB"):  # same-box A/B of kernel builds (tools/build_variant.sh)
    sis_hip.LIB_PATH = os.path.join(ROOT, "synthesis-in-style_amd", "lib", os.environ["SIS_HIP_LIB"])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = sys.argv[3] if len(sys.argv) > 3 else ""
WINO = os.environ.get("SIS_WINOGRAD", "1") != "0"
dev = torch.device("cuda:0")
layers = [("conv", 512, 512, 4), ("up", 512, 512, 4), ("conv", 512, 512, 8), ("up", 512, 512, 8),
          ("conv", 512, 512, 16), ("up", 512, 512, 16), ("conv", 512, 512, 32), ("up", 512, 512, 32),
          ("conv", 512, 512, 64), ("up", 512, 256, 64), ("conv", 256, 256, 128), ("up", 256, 128, 128),
          ("conv", 128, 128, 256)]
tot = 0.0
for kind, cin, cout, h in layers:
    if only and only != f"{kind}{h}":
        continue
    x = torch.randn(B, cin, h, h, device=dev)
    w = torch.randn(1, cout, cin, 3, 3, device=dev)
    s = 1 + 0.1 * torch.randn(B, cin, device=dev)
    wpk, wsq = sis_hip.modconv_prepack(w)
    ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
    oh = h if kind == "conv" else 2 * h
    noise = torch.randn(1, 1, oh, oh, device=dev)
    nw = torch.full((1,), 0.1, device=dev)
    bias = torch.zeros(cout, device=dev)
    u = sis_hip.modconv_prepack_wino(w) if (WINO and kind == "conv") else None
    f = (lambda: sis_hip.modconv2d(x, wpk, s, ds, 3, noise, nw, bias, fuse_act=True, wino_u=u)) if kind == "conv" else \
        (lambda: sis_hip.modconv2d_up(x, wpk, s, ds))
    f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / REPS
    fl = 2.0 * B * cout * cin * 9 * h * h
    tot += ms
    print(f"{kind:5s} {cin:4d}->{cout:4d} @{h:3d}  {ms:8.3f} ms  {fl / ms / 1e9:7.2f} TFLOP/s", flush=True)
print(f"total {tot:.3f} ms")
