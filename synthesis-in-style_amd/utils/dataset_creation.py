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
import os
from typing import Dict, Iterable, Optional, Tuple

import torch

import sis_hip

from latent_projecting import Latents


def seeded_latents(batch: int, dim: int, device) -> torch.Tensor:
    """``torch.randn(batch, dim)`` from the CPU generator (the reference's latent stream, same values) drawn into PINNED host
    memory when it is headed for a HIP device: the copy that follows (``.to(device, non_blocking=True)``) is then a true
    asynchronous transfer (from pageable memory the runtime stages it through a bounce buffer on the calling thread).  Measured
    on the dataset loop: no difference (15.88 ms per batch of 32 either way, profiles/r05_dataset_vs_synthesis.txt) -- the loop's
    0.9 ms over bare synthesis is the label pass itself: 2.7 GB of activations re-read while the next batch's convolutions
    run, which slows those by what the pass would have cost alone (profiles/r05_dataset_timeline.txt)."""
    device = torch.device(device)
    return torch.randn(batch, dim, pin_memory=device.type == 'cuda' and torch.cuda.is_available())


def build_latent_and_noise_generator(autoencoder, config: Dict, seed=1) -> Iterable:
    torch.random.manual_seed(seed)
    decoder = autoencoder.decoder
    while True:
        device = next(decoder.parameters()).device
        yield Latents(seeded_latents(config['batch_size'], config['latent_size'], device), decoder.make_noise())


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


_LABEL_STREAMS = {}  # device -> side stream of the label / uint8 pass


def label_and_encode(image: torch.Tensor, activations: Dict[int, torch.Tensor], catalogs: Dict) \
        -> Tuple[torch.Tensor, Dict[int, torch.Tensor], Optional[torch.cuda.Event]]:
    """What the reference does with a batch after ``generate_images`` (create_dataset_for_segmentation.py:131-135 ->
    ``predict_clusters`` -> ``FactorCatalog.predict``; ``make_image``): nearest-centre label maps of the catalogued
    activation layers and the uint8 NHWC image, here issued on a side stream.  Both passes are HBM-bound
    readers of finished tensors, so they overlap the MFMA-bound first layers of the NEXT batch, whose small grids
    leave most compute units idle.  Returns (pixels u8 [B,H,W,3], {layer: labels}, event): wait for / synchronise on
    the event before touching the results from another stream or the host.  SIS_LABEL_STREAM=0 keeps everything on the
    current stream (event None)."""
    device = image.device
    if os.environ.get("SIS_LABEL_STREAM", "1") == "0" or not image.is_cuda:
        return sis_hip.make_image_u8(image), {k: cat.predict(activations[k]) for k, cat in catalogs.items()}, None
    if device not in _LABEL_STREAMS:
        _LABEL_STREAMS[device] = sis_hip.side_stream(device)
    side, main = _LABEL_STREAMS[device], torch.cuda.current_stream(device)
    side.wait_event(main.record_event())
    with torch.cuda.stream(side):
        labels = {k: cat.predict(activations[k]) for k, cat in catalogs.items()}
        pixels = sis_hip.make_image_u8(image)
    for t in [image] + [activations[k] for k in catalogs]:
        t.record_stream(side)  # the caching allocator must not hand these blocks out again before the side pass has read them
    return pixels, labels, side.record_event()
