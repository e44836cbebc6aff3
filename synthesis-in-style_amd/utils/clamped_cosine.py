"""Cosine learning-rate schedule that stops at ``T_max`` and then holds ``eta_min``
(reference: utils/clamped_cosine.py:8-19; stepped once per iteration, train.py:56)."""
import math


class ClampedCosineAnnealingLR:
    """Closed form lr_t = eta_min + (base - eta_min) * (1 + cos(pi * t / T_max)) / 2 for t <= T_max, eta_min after.
    Works on any optimizer exposing ``param_groups`` (incl. training.fused_sgd.FusedSGD)."""

    def __init__(self, optimizer, T_max, eta_min=0.0, last_epoch=-1):
        self.optimizer, self.T_max, self.eta_min = optimizer, T_max, eta_min
        self.base_lrs = [g['lr'] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def get_lr(self):
        t = self.last_epoch
        if t > self.T_max:
            return [self.eta_min for _ in self.base_lrs]
        return [self.eta_min + (b - self.eta_min) * (1 + math.cos(math.pi * t / self.T_max)) / 2 for b in self.base_lrs]

    def step(self):
        self.last_epoch += 1
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group['lr'] = lr
