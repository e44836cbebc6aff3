"""Per-layer timing of the bf16 MFMA convolution (csrc/conv_bf16.hip) against the library's bf16 path on TransUNet's
shapes at 512^2, B = 8 (BASELINE.json configs[4]).  python tools/bench_conv_bf16.py [fwd|dgrad]"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip  # noqa: E402

B = 8
LAYERS = [  # name, cin, cout, h, w, k, stride
    ("b1.conv1 1x1", 256, 64, 127, 127, 1, 1), ("b1.conv2 3x3", 64, 64, 127, 127, 3, 1), ("b1.conv3 1x1", 64, 256, 127, 127, 1, 1),
    ("b2.conv1 1x1", 512, 128, 64, 64, 1, 1), ("b2.conv2 3x3", 128, 128, 64, 64, 3, 1), ("b2.conv3 1x1", 128, 512, 64, 64, 1, 1),
    ("b2.u1 conv2 s2", 128, 128, 127, 127, 3, 2), ("b2.u1 down s2", 256, 512, 127, 127, 1, 2),
    ("b3.conv1 1x1", 1024, 256, 32, 32, 1, 1), ("b3.conv2 3x3", 256, 256, 32, 32, 3, 1), ("b3.conv3 1x1", 256, 1024, 32, 32, 1, 1),
    ("patch emb 1x1", 1024, 768, 32, 32, 1, 1), ("conv_more", 768, 512, 32, 32, 3, 1),
    ("dec0.conv1", 1024, 256, 64, 64, 3, 1), ("dec0.conv2", 256, 256, 64, 64, 3, 1),
    ("dec1.conv1", 512, 128, 128, 128, 3, 1), ("dec1.conv2", 128, 128, 128, 128, 3, 1),
    ("dec2.conv1", 192, 64, 256, 256, 3, 1), ("dec2.conv2", 64, 64, 256, 256, 3, 1),
    ("dec3.conv1", 64, 16, 512, 512, 3, 1), ("dec3.conv2", 16, 16, 512, 512, 3, 1), ("head", 16, 3, 512, 512, 3, 1),
]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    dev = torch.device("cuda:0")
    tot_mine = tot_lib = 0.0
    print(f"{'layer':18s} {'GF':>7s} {'mine ms':>8s} {'TF/s':>7s} {'lib ms':>8s} {'TF/s':>7s}  kernel")
    for name, cin, cout, h, w, k, s in LAYERS:
        if mode == "wgrad":
            if k != 3 or s != 1 or not sis_hip.conv_bf16_wgrad_supported(B, cin, cout, h, w):
                continue
            x = torch.randn(B, cin, h, w, device=dev).bfloat16()
            gy = torch.randn(B, cout, h, w, device=dev).bfloat16()
            wt = torch.zeros(cout, cin, 3, 3, device=dev).bfloat16()
            gf = 2.0 * B * cout * cin * 9 * h * w / 1e9
            t_mine = timeit(lambda: sis_hip.conv_bf16_wgrad(x, gy, torch.float32))
            kern = sis_hip.lib().sis_last_kernel().decode()
            t_lib = float("nan") if os.environ.get("SIS_BENCH_NO_LIB") else timeit(
                lambda: torch.ops.aten.convolution_backward(gy, x, wt, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
            tot_mine += t_mine
            tot_lib += t_lib
            print(f"{name:18s} {gf:7.1f} {t_mine:8.3f} {gf / t_mine:7.1f} {t_lib:8.3f} {gf / t_lib:7.1f}  {kern}", flush=True)
            continue
        if mode == "dgrad":
            if s != 1:
                continue
            cin, cout = cout, cin
        x = torch.randn(B, cin, h, w, device=dev).bfloat16()
        wt = (torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5).bfloat16()
        packed = sis_hip.conv_bf16_pack(wt, h, w, s)
        ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
        gf = 2.0 * B * cout * cin * k * k * ho * wo / 1e9
        t_mine = timeit(lambda: sis_hip.conv_bf16(x, packed, cout, k, s))
        kern = sis_hip.lib().sis_last_kernel().decode()
        t_lib = timeit(lambda: F.conv2d(x, wt, None, s, k // 2))
        tot_mine += t_mine
        tot_lib += t_lib
        print(f"{name:18s} {gf:7.1f} {t_mine:8.3f} {gf / t_mine:7.1f} {t_lib:8.3f} {gf / t_lib:7.1f}  {kern}", flush=True)
    print(f"total: mine {tot_mine:.2f} ms, library {tot_lib:.2f} ms")


if __name__ == "__main__":
    main()
