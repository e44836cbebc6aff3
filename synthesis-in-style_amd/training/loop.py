"""Minimal training-loop plumbing the updaters need.

The reference drives its updaters with the un-vendored, un-pinned third-party package ``pytorch_training``
(``Updater``, ``GradientApplier``, ``get_current_reporter``; requirements.txt:3) whose source is not part of
the reference tree (parity unpinned at that boundary).  This module states the semantics the
updaters rely on, as observed at their call sites (updater/segmentation_updater.py:2-4,19-39,47-73,83-106,
training_builder/ema_net_train_builder.py:50-59):

* ``Updater(iterators, networks, optimizers, device, copy_to_device)`` holds the three dicts; ``update()``
  runs ``update_core()`` once and counts iterations; ``next_batch(name)`` restarts a finite loader at its end (epoch
  boundary);
* ``GradientApplier(networks, optimizers)`` is a context manager: ``zero_grad`` on entry, ``step`` on exit;
* ``get_current_reporter().add_observation(dict, prefix)`` records scalars under ``prefix/key``;
* ``UpdateDisabler(network)`` (updater/stylegan_2_updater.py:130,164) freezes a network's parameters inside the block;
* ``reduce_sum`` (distributed/__init__.py:4-14) / ``get_world_size``: sum over ranks, identity without a process group.
"""
from typing import Dict, Iterable

import torch.distributed as dist


class Reporter:
    def __init__(self):
        self.observations: Dict[str, object] = {}

    def add_observation(self, values: Dict[str, object], prefix: str = ''):
        for key, value in values.items():
            self.observations[f"{prefix}/{key}" if prefix else key] = value

    def scalars(self) -> Dict[str, float]:
        return {k: float(v) for k, v in self.observations.items()}


_reporter = Reporter()


def get_current_reporter() -> Reporter:
    return _reporter


class GradientApplier:
    def __init__(self, networks: Iterable, optimizers: Iterable):
        self.networks, self.optimizers = list(networks), list(optimizers)

    def __enter__(self):
        for opt in self.optimizers:
            opt.zero_grad()
        return self

    def __exit__(self, exc_type, exc, tb):
        if exc_type is None:
            for opt in self.optimizers:
                opt.step()
        return False


class UpdateDisabler:
    """``requires_grad = False`` on every parameter of ``network`` inside the block, restored afterwards."""

    def __init__(self, network):
        self.network = network
        self._saved = None

    def __enter__(self):
        self._saved = [(p, p.requires_grad) for p in self.network.parameters()]
        for p, _ in self._saved:
            p.requires_grad = False
        return self

    def __exit__(self, exc_type, exc, tb):
        for p, flag in self._saved:
            p.requires_grad = flag
        return False


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def reduce_sum(tensor):
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    tensor = tensor.clone()
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor


class Updater:
    def __init__(self, iterators: Dict, networks: Dict, optimizers: Dict, device='cuda', copy_to_device=True):
        self.loaders = dict(iterators)  # kept so that an exhausted (one-epoch) iterator can be re-created
        self.iterators = {k: iter(v) for k, v in iterators.items()}
        self.networks = networks
        self.optimizers = optimizers
        self.device = device
        self.copy_to_device = copy_to_device
        self.iteration = 0

    def update(self):
        self.update_core()
        self.iteration += 1

    def next_batch(self, name: str):
        """Next batch of ``iterators[name]``; a finite loader (one pass = one epoch) is restarted at its end, as the
        reference's trainer re-creates its iterators per epoch."""
        try:
            return next(self.iterators[name])
        except StopIteration:
            self.iterators[name] = iter(self.loaders[name])
            try:
                return next(self.iterators[name])
            except StopIteration:
                raise RuntimeError(f"data loader '{name}' yields no batches") from None

    def update_core(self):
        raise NotImplementedError
