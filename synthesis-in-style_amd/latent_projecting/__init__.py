"""``Latents`` container of the synthesis driver (reference: latent_projecting/__init__.py:15-37).
Only the container is on the hot path; the projection losses/optimisers of the reference module are
out of scope."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import torch


@dataclass
class Latents:
    latent: torch.Tensor
    noise: List[torch.Tensor]

    def to(self, device) -> Latents:
        # (non_blocking: a latent drawn into pinned memory -- utils.dataset_creation.seeded_latents -- crosses asynchronously)
        self.latent = self.latent.to(device, non_blocking=True)
        self.noise = [n.to(device, non_blocking=True) for n in self.noise]
        return self

    def __getitem__(self, key: int) -> Latents:
        return Latents(self.latent[key].unsqueeze(0), [n[key].unsqueeze(0) for n in self.noise])

    def detach(self):
        self.latent = self.latent.detach()
        self.noise = [n.detach() for n in self.noise]

    def numpy(self) -> Latents:
        return Latents(self.latent.cpu().numpy(), [n.cpu().numpy() for n in self.noise])
