"""The three large up-convolutions of Generator(256) at B = 32: fast-FIR kernel (csrc/modconv_upfir.hip) against the 4-phase
gather kernel (csrc/modconv_mfma2.hip), padded rows as the forward uses them, interleaved rounds in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import torch  # noqa: E402
import sis_hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot_old = tot_new = 0.0
for cin, cout, h in ((512, 512, 32), (512, 256, 64), (256, 128, 128)):
    x = torch.randn(B, cin, h, h, device=dev)
    w = torch.randn(1, cout, cin, 3, 3, device=dev)
    s = 1 + 0.1 * torch.randn(B, cin, device=dev)
    wpk, wsq = sis_hip.modconv_prepack(w)
    fir = sis_hip.modconv_prepack_up_fir(w)
    ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
    old = lambda: sis_hip.modconv2d_up(x, wpk, s, ds, padded_rows=True)               # noqa: E731
    new = lambda: sis_hip.modconv2d_up(x, wpk, s, ds, padded_rows=True, fir_u=fir)    # noqa: E731
    a, b = old(), new()
    err = ((a[..., :2 * h + 1] - b[..., :2 * h + 1]).abs().max() / a[..., :2 * h + 1].abs().max()).item()
    t_old, t_new = [], []
    for _ in range(4):
        t_old.append(timeit(old))
        t_new.append(timeit(new))
    o, n = sorted(t_old)[1], sorted(t_new)[1]
    fl = 2.0 * B * cout * cin * 9 * h * h
    tot_old += o
    tot_new += n
    print(f"up {cin:4d}->{cout:4d} @{h:3d}: 4-phase {o:7.3f} ms = {fl / o / 1e9:6.1f} TF   fast-FIR {n:7.3f} ms = {fl / n / 1e9:6.1f} TF direct-form "
          f"({fl * 25 / 36 / n / 1e9:6.1f} executed)   x{o / n:.2f}   max rel diff {err:.1e}", flush=True)
print(f"sum: {tot_old:.3f} -> {tot_new:.3f} ms")
