"""k-means label pass of the dataset loop, kernel by kernel (rocprofv3 --kernel-trace --stats around this, or the wall times
printed here): the four catalogued layers at B = 32, first pass on the matrix cores (SIS_KMEANS_MFMA=1) or on the VALU (0)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import sis_hip  # noqa: E402

dev = torch.device("cuda:0")
B = 32
for c, hw in ((512, 64), (128, 256)):
    x = torch.randn(B, c, hw, hw, device=dev)
    cen = torch.randn(24, c, device=dev)
    for mode in ("1", "0", "1", "0"):
        os.environ["SIS_KMEANS_MFMA"] = mode
        for _ in range(3):
            lab = sis_hip.kmeans_assign(x, cen)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            lab = sis_hip.kmeans_assign(x, cen)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        ws = torch.empty(sis_hip.lib().sis_kmeans_workspace_ints(B, hw * hw), dtype=torch.int32, device=dev)
        labels = torch.empty((B, hw, hw), dtype=torch.int64, device=dev)
        sis_hip.lib().sis_kmeans_assign_ws(sis_hip._ptr(labels), sis_hip._ptr(x), sis_hip._ptr(cen), B, c, hw * hw, 24, sis_hip._ptr(ws), ws.numel(),
                                           sis_hip._stream())
        torch.cuda.synchronize()
        print(f"C {c} HW {hw}x{hw} mfma={mode}: {ms:.3f} ms  ({x.numel() * 4 / ms / 1e9:.2f} TB/s of x)  open pixels {int(ws[0])} of {B * hw * hw}"
              f" ({100.0 * int(ws[0]) / (B * hw * hw):.2f} %)", flush=True)
