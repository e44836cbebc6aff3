"""Which part of the dataset loop costs what: wall ms per batch of 32 for the loop with pieces removed (interleaved rounds)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
import bench  # noqa: E402
import sis_hip  # noqa: E402
from segmentation.gan_local_edit.factor_catalog import FactorCatalog  # noqa: E402
from utils.dataset_creation import label_and_encode, seeded_latents  # noqa: E402

dev = torch.device("cuda:0")
g = bench.build_generator(dev)
rng = np.random.RandomState(7)
catalogs = {k: FactorCatalog(cluster_centers=rng.randn(24, c).astype(np.float32)) for k, c in {8: 512, 9: 512, 12: 128, 13: 128}.items()}
B = 32
z_dev = torch.randn(B, g.style_dim, device=dev)
noise_fixed = g.make_noise()


def full():
    z = seeded_latents(B, g.style_dim, dev).to(dev, non_blocking=True)
    image, acts = g([z], noise=g.make_noise(), return_intermediate_activations=True)
    return label_and_encode(image, acts, catalogs)


def labels_main_stream():
    os.environ["SIS_LABEL_STREAM"] = "0"
    try:
        return full()
    finally:
        os.environ["SIS_LABEL_STREAM"] = "1"


def no_labels():
    z = seeded_latents(B, g.style_dim, dev).to(dev, non_blocking=True)
    return g([z], noise=g.make_noise(), return_intermediate_activations=True)


def no_activations():
    z = seeded_latents(B, g.style_dim, dev).to(dev, non_blocking=True)
    return g([z], noise=g.make_noise())


def device_latents():
    return g([z_dev], noise=g.make_noise())


def fixed_noise():
    return g([z_dev], noise=noise_fixed)


def labels_only_image():
    z = seeded_latents(B, g.style_dim, dev).to(dev, non_blocking=True)
    image, acts = g([z], noise=g.make_noise(), return_intermediate_activations=True)
    return sis_hip.make_image_u8(image)


variants = [full, labels_main_stream, no_labels, no_activations, device_latents, fixed_noise, labels_only_image]
with torch.no_grad():
    for fn in variants:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for rnd in range(3):
        for fn in variants:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                out = fn()
            torch.cuda.synchronize()
            print(f"round {rnd} {fn.__name__:20s} {1e3 * (time.perf_counter() - t0) / 20:7.3f} ms per batch", flush=True)
