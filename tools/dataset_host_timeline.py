"""Host-side timeline of the dataset loop (bench.py --workload dataset): where does the CPU spend a batch, and is it ahead of
the GPU?  Prints per-stage host milliseconds (median over the batches) and the GPU time per batch."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import bench  # noqa: E402
from segmentation.gan_local_edit.factor_catalog import FactorCatalog  # noqa: E402
from utils.dataset_creation import label_and_encode, seeded_latents  # noqa: E402

dev = torch.device("cuda:0")
g = bench.build_generator(dev)
rng = np.random.RandomState(7)
catalogs = {k: FactorCatalog(cluster_centers=rng.randn(24, c).astype(np.float32)) for k, c in {8: 512, 9: 512, 12: 128, 13: 128}.items()}
B = 32
stages = {"latents": [], "h2d": [], "noise": [], "forward": [], "labels": []}
N = 14
with torch.no_grad():
    for it in range(N + 4):
        if it == 4:
            torch.cuda.synchronize()
            t_start = time.perf_counter()
        t0 = time.perf_counter()
        z = seeded_latents(B, g.style_dim, dev)
        t1 = time.perf_counter()
        zd = z.to(dev, non_blocking=True)
        t2 = time.perf_counter()
        noise = g.make_noise()
        t3 = time.perf_counter()
        image, acts = g([zd], noise=noise, return_intermediate_activations=True)
        t4 = time.perf_counter()
        out = label_and_encode(image, acts, catalogs)
        t5 = time.perf_counter()
        if it >= 4:
            for k, v in zip(stages, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
                stages[k].append(v * 1e3)
    t_issue = time.perf_counter()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
print(f"{N} batches: host issue {1e3 * (t_issue - t_start) / N:.3f} ms per batch, wall {1e3 * (t_end - t_start) / N:.3f} ms per batch")
for k, v in stages.items():
    print(f"  {k:8s} median {np.median(v):7.3f} ms   max {max(v):7.3f} ms")
print("allocator:", {k: v for k, v in torch.cuda.memory_stats(dev).items() if k in ("num_alloc_retries", "num_device_alloc", "num_device_free", "reserved_bytes.all.peak")})
