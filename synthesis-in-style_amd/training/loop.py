"""Minimal training-loop plumbing the updaters need.

The reference drives its updaters with the un-vendored, un-pinned third-party package ``pytorch_training``
(``Updater``, ``GradientApplier``, ``get_current_reporter``; requirements.txt:3) whose source is not part of
the reference tree (parity unpinned at that boundary).  This module states the semantics the
updaters rely on, as observed at their call sites (updater/segmentation_updater.py:2-4,19-39,47-73,83-106,
training_builder/ema_net_train_builder.py:50-59):

* ``Updater(iterators, networks, optimizers, device, copy_to_device)`` holds the three dicts; ``update()``
  runs ``update_core()`` once and counts iterations;
* ``GradientApplier(networks, optimizers)`` is a context manager: ``zero_grad`` on entry, ``step`` on exit;
* ``get_current_reporter().add_observation(dict, prefix)`` records scalars under ``prefix/key``.
"""
from typing import Dict, Iterable


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


class Updater:
    def __init__(self, iterators: Dict, networks: Dict, optimizers: Dict, device='cuda', copy_to_device=True):
        self.iterators = {k: iter(v) for k, v in iterators.items()}
        self.networks = networks
        self.optimizers = optimizers
        self.device = device
        self.copy_to_device = copy_to_device
        self.iteration = 0

    def update(self):
        self.update_core()
        self.iteration += 1

    def update_core(self):
        raise NotImplementedError
