"""Builder base: prepares the network (device, optional fine-tune weights, DistributedDataParallel) and hands
out optimizer / updater / snapshot pieces.  Mirrors training_builder/base_train_builder.py:21-102.

Multi-GPU: one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm).  The wrap
keeps the reference's semantics -- ``broadcast_buffers=False`` (batch-norm statistics and EMANet's ``emau.mu``
stay per-rank), ``find_unused_parameters`` as the concrete builder sets it -- and adds what suits the fabric:
25 MB buckets reduced while backward is still producing earlier layers' gradients (an EMANet-50 step has
139 MB of fp32 gradients: ~6 buckets, each ~0.3 ms on a 153 GB/s xGMI link, hidden under >= 20 ms of backward),
and ``gradient_as_bucket_view=True`` so gradients live in the buckets: no copy in or out, and stable storage
for the fused optimizer's pointer table.
"""
import functools
from pathlib import Path
from typing import Dict, Union

import torch
from torch.nn.parallel import DistributedDataParallel as DDP
from torch.optim import Optimizer
from torch.utils.data import DataLoader

from networks.base_segmenter import BaseSegmenter


def strip_parallel_module(network):
    return network.module if isinstance(network, DDP) else network


def load_weights(network, checkpoint_path, key='segmentation_network', strict=True):
    """networks/__init__.py:22-29 of the reference: a checkpoint dict holding a state_dict under ``key``."""
    checkpoint = torch.load(checkpoint_path, map_location='cpu')
    state = checkpoint[key] if key in checkpoint else checkpoint
    strip_parallel_module(network).load_state_dict(state, strict=strict)
    return network


class BaseTrainBuilder:
    def __init__(self, config: dict, train_data_loader: Union[DataLoader, None] = None,
                 val_data_loader: Union[DataLoader, None] = None, rank: int = 0, world_size: int = 1):
        self.segmentation_network = None
        self.train_data_loader = train_data_loader
        self.val_data_loader = val_data_loader
        self.config = config
        self.fine_tune = config.get('fine_tune')
        self.rank = rank
        self.world_size = world_size
        self.find_unused_params = False
        self._optimizers = None

    def device(self):
        if torch.cuda.is_available():
            return torch.device('cuda', self.rank % max(torch.cuda.device_count(), 1))
        return torch.device('cpu')  # gloo rehearsals of the data-parallel plumbing

    def _prepare_segmentation_network(self, segmentation_network: BaseSegmenter,
                                      network_name: str = 'segmentation_network') -> BaseSegmenter:
        assert segmentation_network is not None, 'Segmentation network was not properly initialized!'
        device = self.device()
        segmentation_network.to(device)
        if self.fine_tune is not None:
            load_weights(segmentation_network, self.fine_tune, key=network_name)
        if self.world_size > 1:
            wrap = functools.partial(DDP, find_unused_parameters=self.find_unused_params, broadcast_buffers=False,
                                     bucket_cap_mb=self.config.get('bucket_cap_mb', 25), gradient_as_bucket_view=True)
            if device.type == 'cuda':
                segmentation_network = wrap(segmentation_network, device_ids=[device.index], output_device=device.index)
            else:
                segmentation_network = wrap(segmentation_network)
        return segmentation_network

    def _initialize_segmentation_network(self):
        raise NotImplementedError

    def get_network(self) -> BaseSegmenter:
        return self.segmentation_network

    def get_networks_for_updater(self) -> Dict[str, BaseSegmenter]:
        raise NotImplementedError

    def get_optimizers(self) -> Dict[str, Optimizer]:
        raise NotImplementedError

    def get_updater(self):
        raise NotImplementedError

    def get_snapshotter(self):
        raise NotImplementedError

    def get_evaluator(self, logger):
        return None

    def get_image_plotter(self):
        return None  # image grids are wandb/PIL-side visualisation, outside the training step


class Snapshotter:
    """Rank-0 checkpoint writer with the reference's dict layout: ``{'segmentation_network': state_dict,
    'main': optimizer.state_dict()}`` (base_train_builder.py:91-102)."""

    def __init__(self, targets: Dict, log_dir, every: int):
        self.targets, self.log_dir, self.every = targets, Path(log_dir), every

    def maybe_save(self, iteration: int):
        if self.every and iteration % self.every == 0:
            self.log_dir.mkdir(parents=True, exist_ok=True)
            torch.save({k: strip_parallel_module(v).state_dict() for k, v in self.targets.items()},
                       self.log_dir / f"{iteration:06d}.pt")


class BaseSingleNetworkTrainBuilder(BaseTrainBuilder):
    def get_networks_for_updater(self) -> Dict[str, BaseSegmenter]:
        return {'segmentation': self.segmentation_network}

    def get_snapshotter(self):
        if self.rank != 0:
            return None
        return Snapshotter({'segmentation_network': self.segmentation_network, **self.get_optimizers()},
                           self.config.get('log_dir', 'logs'), self.config.get('snapshot_save_iter', 0))
