"""EMANet-50 convolution shapes (B=16, 256^2 input): this library's plain-conv path (modulated kernels with unit
style / demodulation) against ATen (MIOpen / hipBLASLt) forward.  usage: python tools/bench_conv_shapes.py [batch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch
import torch.nn.functional as F
import sis_hip

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda")
shapes = [  # (k, cin, cout, hw, dilation)
    (3, 2048, 512, 32, 1), (3, 512, 256, 32, 1), (3, 64, 64, 128, 1), (3, 64, 128, 128, 1), (3, 64, 64, 64, 1),
    (3, 128, 128, 32, 1), (3, 256, 256, 32, 2), (3, 512, 512, 32, 2), (3, 512, 512, 32, 4), (3, 512, 512, 32, 8),
    (1, 128, 64, 64, 1), (1, 64, 256, 64, 1), (1, 256, 64, 64, 1), (1, 128, 512, 32, 1), (1, 512, 128, 32, 1),
    (1, 512, 256, 32, 1), (1, 256, 1024, 32, 1), (1, 1024, 256, 32, 1), (1, 1024, 512, 32, 1), (1, 512, 2048, 32, 1),
    (1, 2048, 512, 32, 1), (1, 512, 512, 32, 1),
]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"{'shape':34s} {'aten ms':>8s} {'ours ms':>8s} {'TF aten':>8s} {'TF ours':>8s}")
for k, cin, cout, hw, dil in shapes:
    x = torch.randn(B, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * (cin * k * k) ** -0.5
    flops = 2.0 * B * cout * cin * k * k * hw * hw
    t_aten = timeit(lambda: F.conv2d(x, w, padding=dil * (k // 2), dilation=dil))
    ours = float("nan")
    if dil > 1 and k == 3:
        d = dil
        u = sis_hip.conv3x3_prepack(w)

        def s2b(t):
            b, c, h, ww = t.shape
            return t.view(b, c, h // d, d, ww // d, d).permute(0, 3, 5, 1, 2, 4).reshape(b * d * d, c, h // d, ww // d)

        def b2s(t, b):
            _, c, h, ww = t.shape
            return t.view(b, d, d, c, h, ww).permute(0, 3, 4, 1, 5, 2).reshape(b, c, h * d, ww * d)

        def run():
            return b2s(sis_hip.conv3x3(s2b(x).contiguous(), u), B)
        y = run()
        ref = F.conv2d(x, w, padding=dil, dilation=dil)
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        assert err < 1e-4, err
        ours = timeit(run)
    if dil == 1:
        wpk, _ = sis_hip.modconv_prepack(w.view(1, cout, cin, k, k))
        u = sis_hip.modconv_prepack_wino(w.view(1, cout, cin, k, k)) if k == 3 else None
        s = torch.ones(B, cin, device=dev)
        d = torch.ones(B, cout, device=dev)
        y = sis_hip.modconv2d(x, wpk, s, d, k, wino_u=u)
        ref = F.conv2d(x, w, padding=k // 2)
        err = ((y - ref).abs().max() / ref.abs().max()).item()
        assert err < 1e-4, err
        ours = timeit(lambda: sis_hip.modconv2d(x, wpk, s, d, k, wino_u=u))
    print(f"{k}x{k} {cin:4d}->{cout:4d} @{hw:3d} d{dil}           {t_aten:8.3f} {ours:8.3f} {flops/t_aten/1e9:8.1f} {flops/ours/1e9:8.1f}")
