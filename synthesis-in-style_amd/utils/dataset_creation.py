"""Synthesis driver: the seeded latent/noise stream and the batched generator call.

Mirrors utils/dataset_creation.py:32-58 of the reference (the two functions the hot loop of
create_dataset_for_segmentation.py:129-148 calls).  Same observable behaviour:

* ``build_latent_and_noise_generator``: ``torch.random.manual_seed(seed)`` once, then per batch
  ``z = randn(B, latent_size)`` from the CPU RNG and ``decoder.make_noise()`` on the generator's device;
* ``generate_images``: move the batch to the device, run ``decoder([z], input_is_latent=False,
  noise=..., return_intermediate_activations=True, truncation=0.7 iff a mean latent is given)`` under
  ``torch.no_grad()``, return ``(activations, image)``.

The image-encoding branch of the reference (``autoencoder.encode`` for non-``Latents`` batches) belongs
to the projection research code and is out of scope.
"""
from typing import Dict, Iterable, Optional, Tuple

import torch

from latent_projecting import Latents


def build_latent_and_noise_generator(autoencoder, config: Dict, seed=1) -> Iterable:
    torch.random.manual_seed(seed)
    while True:
        yield Latents(torch.randn(config['batch_size'], config['latent_size']), autoencoder.decoder.make_noise())


def shard_range(num_images: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Image-id range [lo, hi) of one rank: contiguous blocks, remainder spread over the first ranks
    (multi-GPU synthesis is embarrassingly parallel -- SURVEY.md §8e; the reference is single-GPU)."""
    base, rem = divmod(num_images, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def generate_images(batch: Latents, autoencoder, device='cuda', mean_latent: Optional[torch.Tensor] = None) \
        -> Tuple[Dict[int, torch.Tensor], torch.Tensor]:
    if not isinstance(batch, Latents):
        raise NotImplementedError("only Latents batches are supported (image encoding is out of scope)")
    latents = batch.to(device)
    with torch.no_grad():
        image, activations = autoencoder.decoder(
            [latents.latent], input_is_latent=False, noise=latents.noise, return_intermediate_activations=True,
            truncation=0.7 if mean_latent is not None else 1, truncation_latent=mean_latent)
    return activations, image
