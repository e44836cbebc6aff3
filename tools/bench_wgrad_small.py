import sys, time
sys.path.insert(0, "synthesis-in-style_amd")
import torch, sis_hip
dev = torch.device("cuda")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (b, cin, cout, h) in [(8, 512, 512, 4), (8, 512, 512, 8), (8, 512, 512, 16), (8, 512, 512, 32), (8, 256, 256, 64), (8, 128, 128, 128), (24, 512, 512, 4), (24, 512, 512, 8), (24, 512, 512, 16), (16, 64, 64, 64), (5, 64, 64, 16), (16, 128, 128, 32)]:
    if not sis_hip.conv3x3_wgrad_supported(b, cin, cout, h, h, min_work=0):
        print("ineligible", b, cin, cout, h); continue
    x = torch.randn(b, cin, h, h, device=dev); gy = torch.randn(b, cout, h, h, device=dev); w = torch.zeros(cout, cin, 3, 3, device=dev)
    tl = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False)))
    to = timeit(lambda: sis_hip.conv3x3_wgrad(x, gy))
    print(f"{b}x{cin}->{cout} @{h}: work {b*h*h*cin*cout:.1e}  library {tl:.3f} ms  own {to:.3f} ms")
