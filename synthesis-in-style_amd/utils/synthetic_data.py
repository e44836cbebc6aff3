"""Synthetic batches with the reference loaders' contract (data/segmentation_dataset.py:60-63,
utils/data_loading.py:38-42): ``images`` float32 [B,3,S,S] in [-1,1], ``segmented`` int64 [B,1,S,S] in [0,C).
The real input pipeline (JSON-listed PNG pairs, imgaug) is host-side and outside the hot path; benchmarks and
tests feed the training step from here (per-rank seed 1234 + rank)."""
import torch


class SyntheticSegmentationLoader:
    def __init__(self, batch_size, image_size, num_classes, seed=1234, device=None, num_batches=None, distinct=4):
        gen = torch.Generator().manual_seed(seed)
        self.batches = []
        for _ in range(distinct):
            images = torch.rand(batch_size, 3, image_size, image_size, generator=gen) * 2 - 1
            labels = torch.randint(0, num_classes, (batch_size, 1, image_size, image_size), generator=gen)
            if device is not None:
                images, labels = images.to(device), labels.to(device)
            self.batches.append({'images': images, 'segmented': labels})
        self.num_batches = num_batches

    def __len__(self):
        return self.num_batches if self.num_batches is not None else 1 << 30

    def __iter__(self):
        i = 0
        while self.num_batches is None or i < self.num_batches:
            yield dict(self.batches[i % len(self.batches)])
            i += 1
