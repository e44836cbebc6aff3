"""Tensor hand-off from synthesis to segmentation training.

The reference writes every synthetic sample as a side-by-side ``[image | label]`` PNG
(create_dataset_for_segmentation.py:84-99) and ``SegmentationDataset.__getitem__``
(data/segmentation_dataset.py:44-63) splits it again: left half -> ``ToTensor`` + ``Normalize(0.5, 0.5)``
(u8 / 255, then (x - 0.5) / 0.5), right half -> colours -> class ids -> nearest-neighbour resize to ``image_size`` ->
int64 ``[1, S, S]``.  PNG is lossless, so the batch the trainer sees is a pure function of the uint8 pixels and the
label map that ``utils.dataset_creation.label_and_encode`` leaves ON THE DEVICE.  ``encode_batch`` is that function
(same arithmetic, same order, bit for bit -- tests/test_dataset_ops_gpu.py round-trips a PNG through PIL against it)
and ``SynthesisSegmentationLoader`` feeds an updater straight from a generator: no PNG, no PIL, no host copy.
"""
from typing import Dict, Iterator, Optional

import torch
import torch.nn.functional as F

from utils.dataset_creation import label_and_encode


def encode_batch(pixels: torch.Tensor, labels: torch.Tensor, class_of_cluster: Optional[torch.Tensor] = None,
                 image_size: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """pixels uint8 [B,H,W,3] (``make_image``), labels int64 [B,h,w] cluster ids -> the reference loaders' batch contract
    (``images`` float32 [B,3,H,W] in [-1,1], ``segmented`` int64 [B,1,S,S]).  ``class_of_cluster`` [K] maps cluster ids to
    class ids (the reference's cluster -> class merge + colour map, applied as a lookup); ``image_size`` resizes the label
    map with nearest neighbours exactly as ``class_image_to_tensor`` does (segmentation_dataset.py:37-42)."""
    # divisors as tensors: ATen turns a division by a Python scalar into a multiplication by its reciprocal on the device,
    # which is one ulp off the true division ToTensor performs on the host for some of the 256 byte values
    d255 = torch.full((), 255.0, dtype=torch.float32, device=pixels.device)
    images = pixels.permute(0, 3, 1, 2).to(torch.float32).div(d255).sub(0.5).div(0.5).contiguous()
    classes = labels if class_of_cluster is None else class_of_cluster.to(labels.device)[labels]
    classes = classes.unsqueeze(1)
    size = image_size if image_size is not None else images.shape[-1]
    if classes.shape[-1] != size or classes.shape[-2] != size:
        classes = F.interpolate(classes.to(torch.float32), (size, size)).to(torch.int64)  # default mode: nearest
    return {"images": images, "segmented": classes.to(torch.int64)}


class SynthesisSegmentationLoader:
    """Endless iterable of training batches synthesised on the fly: seeded latents (CPU RNG stream of
    utils/dataset_creation.py:32-37) -> ``Generator.forward`` with activations -> k-means label map of ``label_layer`` +
    uint8 pixels on the side stream -> ``encode_batch``.  Everything after the latents stays in HBM."""

    def __init__(self, generator, catalogs: Dict, label_layer: int, batch_size: int, class_of_cluster=None,
                 image_size: Optional[int] = None, seed: int = 1, truncation_latent=None, num_batches: Optional[int] = None):
        self.generator, self.catalogs, self.label_layer = generator, catalogs, label_layer
        self.batch_size, self.class_of_cluster, self.image_size = batch_size, class_of_cluster, image_size
        self.seed, self.truncation_latent, self.num_batches = seed, truncation_latent, num_batches

    def __len__(self):
        return self.num_batches if self.num_batches is not None else 1 << 30

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        g = self.generator
        device = g.input.input.device
        rng = torch.Generator().manual_seed(self.seed)
        i = 0
        while self.num_batches is None or i < self.num_batches:
            z = torch.randn(self.batch_size, g.style_dim, generator=rng).to(device, non_blocking=True)
            with torch.no_grad():
                image, acts = g([z], noise=g.make_noise(), return_intermediate_activations=True,
                                truncation=0.7 if self.truncation_latent is not None else 1,
                                truncation_latent=self.truncation_latent)
                pixels, labels, ready = label_and_encode(image, {self.label_layer: acts[self.label_layer]},
                                                         {self.label_layer: self.catalogs[self.label_layer]})
                if ready is not None:
                    torch.cuda.current_stream(device).wait_event(ready)
                batch = encode_batch(pixels, labels[self.label_layer], self.class_of_cluster, self.image_size)
            # yielded OUTSIDE the no_grad block: a generator suspended inside it would leave grad mode off in the consumer
            # (the updater keeps this iterator alive across its forward / backward)
            yield batch
            i += 1
