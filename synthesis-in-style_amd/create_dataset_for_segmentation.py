"""Synthetic (image, label) dataset generation: the hot loop of the reference's
create_dataset_for_segmentation.py:109-148, sharded over GPUs.

What runs per batch: seeded latents (CPU RNG, as utils/dataset_creation.py:32-37) -> ``Generator.forward`` with
intermediate activations on the MI355X kernels -> nearest k-means centre per pixel of the configured activation
layers on the device (FactorCatalog.predict) -> uint8 images on the device -> side-by-side ``[image | label]`` PNGs
in the reference's directory layout ``<id // 100000>/<id // 1000>/<id>.png`` (save_image, :84-90).

Not reproduced (CPU post-processing outside the hot path): class merging, OpenCV contour
extraction / filtering / rendering of the label image, COCO ground truth, train/val split.  The label half of the
PNG is therefore the raw cluster-id map of ``label_layer`` (id * 255 // (K-1) grey levels) instead of the rendered
class colours.

Multi-GPU (BASELINE.json configs[2]: 100k images on 8 GPUs): one process per GPU (``torch.distributed.run`` or
manual RANK/WORLD_SIZE), rank r generates the image-id range ``shard_range(num_images, r, world)``; no collective.
The latent stream is one global seeded stream: every rank draws the whole stream in batch order and keeps its own
rows; the per-batch noise maps ([1,1,h,h], shared by the batch's samples) are likewise drawn from the seeded device RNG
for EVERY batch on every rank, also the batches a rank skips.  So the image an id maps to does not depend on the world
size: the union over ranks equals the single-GPU dataset (tests/test_dataset_ops_gpu.py).
"""
import argparse
import json
import os
from pathlib import Path

import numpy
import torch

import sis_hip  # noqa: F401  (fails loudly at import when libsis_hip.so is missing: there is no CPU path)
from networks import get_stylegan2_generator
from segmentation.gan_local_edit.factor_catalog import FactorCatalog
from utils.dataset_creation import label_and_encode, seeded_latents, shard_range


def save_image(image: numpy.ndarray, image_id: int, base_dir: Path, name_format: str = "{id}.png"):
    from PIL import Image
    dest = base_dir / str(image_id // 100000) / str(image_id // 1000) / name_format.format(id=image_id)
    dest.parent.mkdir(exist_ok=True, parents=True)
    Image.fromarray(image).save(str(dest))


def save_generated_images(generated_images, label_images, first_id: int, base_dir: Path, num_images: int):
    images = numpy.concatenate([generated_images, label_images], axis=2)
    fmt = f"{{id:0{max(4, len(str(num_images)))}d}}.png"
    for idx, image in enumerate(images):
        save_image(image, first_id + idx, base_dir, name_format=fmt)


def load_generator(checkpoint, size, latent_size, n_mlp, channel_multiplier, device):
    """Generator-only branch of ``load_autoencoder_or_generator`` (networks/__init__.py:415-423: key 'g_ema', strict)."""
    g = get_stylegan2_generator(size, latent_size, n_mlp=n_mlp, channel_multiplier=channel_multiplier,
                                init_ckpt=checkpoint or None, ckpt_key='g_ema', strict=True)
    return g.to(device).eval()


def build_dataset(args, creation_config, rank=0, world_size=1):
    device = torch.device('cuda', rank % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(device)
    g = load_generator(args.checkpoint, creation_config.get('image_size', 256), creation_config.get('latent_size', 512),
                       creation_config.get('n_mlp', 8), creation_config.get('channel_multiplier', 2), device)
    catalogs = {}
    for layer, path in creation_config.get('catalogs', {}).items():  # {"13": "centres_13.npy", ...}
        catalogs[int(layer)] = FactorCatalog(cluster_centers=numpy.load(path))
    label_layer = int(creation_config.get('label_layer', max(catalogs) if catalogs else -1))
    mean_latent = g.mean_latent(4096) if args.truncate else None
    lo, hi = shard_range(args.num_images, rank, world_size)
    save_dir = Path(args.save_to) if args.save_to else None
    torch.random.manual_seed(creation_config.get('seed', 1))
    done = 0

    def flush(job):
        """Host side of a finished batch: wait for its label pass, then encode / write the files."""
        first_id, pixels, labels, ready = job
        if ready is not None:
            ready.synchronize()
        if save_dir is None:
            return
        rgb = pixels.cpu().numpy()
        if label_layer in labels:
            k = catalogs[label_layer].cluster_centers.shape[0]
            lab = labels[label_layer]
            if lab.shape[-1] != rgb.shape[2]:
                lab = torch.nn.functional.interpolate(lab[:, None].float(), size=rgb.shape[1:3], mode='nearest')[:, 0].long()
            grey = (lab * 255 // max(k - 1, 1)).to(torch.uint8).cpu().numpy()
            lab_img = numpy.repeat(grey[..., None], 3, axis=3)
        else:
            lab_img = numpy.zeros_like(rgb)
        save_generated_images(rgb, lab_img, first_id, save_dir, args.num_images)

    pending = None
    with torch.no_grad():
        for first in range(0, args.num_images, args.batch_size):
            n = min(args.batch_size, args.num_images - first)
            z = seeded_latents(args.batch_size, g.style_dim, device)[:n]  # the whole stream is drawn on every rank (pinned) ...
            noise = g.make_noise()  # ... and so are the batch's noise maps (device RNG, same seed on every rank)
            a, b = max(first, lo), min(first + n, hi)
            if a >= b:
                continue
            # A batch that straddles a shard boundary is synthesised WHOLE by both of its owners and then cut: the kernels'
            # split-K / tile plans follow the batch size, so only the same batch gives the same bits -- the bytes an image id
            # maps to must not depend on the world size (at most one redundant partial batch per shard boundary).
            image, acts = g([z.to(device, non_blocking=True)], noise=noise, return_intermediate_activations=True,
                            truncation=0.7 if mean_latent is not None else 1, truncation_latent=mean_latent)
            if (a, b) != (first, first + n):
                image = image[a - first:b - first]
                acts = {k: v[a - first:b - first] for k, v in acts.items()}
            job = (a,) + label_and_encode(image, acts, catalogs)  # side stream; the next batch's forward is issued first
            if pending is not None:
                flush(pending)
            pending = job
            done += b - a
        if pending is not None:
            flush(pending)
    torch.cuda.synchronize()
    return done, (lo, hi)


def main(args):
    creation_config = json.load(open(args.config)) if args.config else {}
    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    done, (lo, hi) = build_dataset(args, creation_config, rank, world)
    print(f"rank {rank}/{world}: generated image ids [{lo}, {hi}) = {done} images", flush=True)


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Generate a synthetic dataset with a StyleGAN2 generator on MI355X")
    parser.add_argument("checkpoint", nargs='?', default=None, help="generator checkpoint holding 'g_ema' (omit: random weights)")
    parser.add_argument("config", nargs='?', default=None, help="json: image_size, latent_size, seed, catalogs{layer: centres.npy}, label_layer")
    parser.add_argument("-n", "--num-images", type=int, default=100)
    parser.add_argument("-s", "--save-to", help="directory for the PNG pairs (omit: generate only)")
    parser.add_argument("-b", "--batch-size", default=10, type=int)
    parser.add_argument("--truncate", action='store_true', default=False, help="truncation trick (psi 0.7, mean of 4096 latents)")
    main(parser.parse_args())
