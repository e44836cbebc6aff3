"""Train builders: network preparation (device, optional fine-tune weights, DistributedDataParallel), optimizer,
updater and snapshot wiring for the segmentation networks on the MI355X hot path.

Role of training_builder/base_train_builder.py:21-102 in the reference, organised differently: a concrete builder
only *declares* what is specific to its network (``build_network``, ``parameter_groups``, ``optimizer_defaults``,
``updater_class``, ``updater_options``, ``find_unused_params``); everything else lives here once.  The public
method names the reference's ``train.py`` calls (``get_updater``, ``get_optimizers``, ``get_snapshotter``,
``get_network``, ``get_evaluator``, ``get_image_plotter``) are kept.

Multi-GPU: one process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm).  The wrap keeps the
reference's semantics -- parameters broadcast from rank 0, ``broadcast_buffers=False`` (batch-norm statistics and EMANet's
``emau.mu`` stay per-rank), parameters without a gradient tolerated where the concrete builder declares
``find_unused_params`` -- and adds what suits the fabric: 25 MB buckets reduced while backward is still producing earlier
layers' gradients (an EMANet-50 step has 139 MB of fp32 gradients: ~6 buckets, each ~0.3 ms on one 153 GB/s xGMI link,
hidden under >= 20 ms of backward), gradients that LIVE in the buckets (stable storage for the fused optimizer's pointer
table).  Two implementations, ``data_parallel`` in the config: ``buckets`` (default; training/grad_exchange.py: reduce-scatter
+ all-gather per bucket on a side stream, plan fixed after one backward, capturable in the step's hipGraph) and ``ddp``
(torch's DistributedDataParallel with ``gradient_as_bucket_view=True``; eager only).  ``force_data_parallel: true`` wraps
even a single rank (world_size 1 over RCCL: the rehearsal a one-GPU box allows).

Reference quirk not reproduced: its ``get_optimizers()`` constructs a NEW optimizer on every call, so the LR
scheduler built in train.py:39-56 schedules an optimizer that never trains.  Here the optimizer is created once.
"""
from pathlib import Path
from typing import Dict, Optional

import torch
from torch.nn.parallel import DistributedDataParallel as DDP

from training.fused_sgd import FusedSGD
from training.grad_exchange import BucketedDataParallel


def strip_parallel_module(network):
    return network.module if isinstance(network, (DDP, BucketedDataParallel)) else network


def load_weights(network, checkpoint_path, key='segmentation_network', strict=True):
    """A checkpoint dict holding a state_dict under ``key`` (reference: networks/__init__.py:22-29)."""
    checkpoint = torch.load(checkpoint_path, map_location='cpu')
    strip_parallel_module(network).load_state_dict(checkpoint.get(key, checkpoint), strict=strict)
    return network


class Snapshotter:
    """Rank-0 checkpoint writer, dict layout of the reference: ``{'segmentation_network': ..., 'main': optimizer}``."""

    def __init__(self, targets: Dict, log_dir, every: int):
        self.targets, self.log_dir, self.every = targets, Path(log_dir), every

    def maybe_save(self, iteration: int):
        if self.every and iteration % self.every == 0:
            self.log_dir.mkdir(parents=True, exist_ok=True)
            torch.save({k: strip_parallel_module(v).state_dict() for k, v in self.targets.items()},
                       self.log_dir / f"{iteration:06d}.pt")


class BaseTrainBuilder:
    # ---- what a concrete builder declares ---------------------------------------------------------------
    find_unused_params = False
    updater_class = None

    def build_network(self):
        raise NotImplementedError

    def parameter_groups(self, network):
        return list(network.parameters())

    def optimizer_defaults(self) -> dict:
        raise NotImplementedError

    def updater_options(self) -> dict:
        return {}

    # ---- shared machinery -------------------------------------------------------------------------------
    def __init__(self, config: dict, train_data_loader=None, val_data_loader=None, rank: int = 0, world_size: int = 1,
                 build: bool = True):
        self.config = config
        self.train_data_loader, self.val_data_loader = train_data_loader, val_data_loader
        self.rank, self.world_size = rank, world_size
        self.fine_tune = config.get('fine_tune')
        if config.get('miopen_search'):
            # let MIOpen time its solvers for the library convolutions (stride-2 / bf16 layers) instead of taking the
            # heuristic pick: minutes of search at the first iteration, cached per machine in MIOpen's user database;
            # TransUNet bf16 200 -> 216 images/s, EMANet unchanged (its 3x3 layers are on the HIP kernels anyway)
            torch.backends.cudnn.benchmark = True
        self.segmentation_network = None
        self._optimizers: Optional[Dict] = None
        if build and type(self).build_network is not BaseTrainBuilder.build_network:
            self.segmentation_network = self._prepare_segmentation_network(self.build_network())

    def device(self):
        if torch.cuda.is_available():
            return torch.device('cuda', self.rank % max(torch.cuda.device_count(), 1))
        return torch.device('cpu')  # gloo rehearsals of the data-parallel plumbing

    def _prepare_segmentation_network(self, network, network_name: str = 'segmentation_network'):
        assert network is not None, 'Segmentation network was not properly initialized!'
        device = self.device()
        network.to(device)
        if self.fine_tune is not None:
            load_weights(network, self.fine_tune, key=network_name)
        if self.world_size > 1 or self.config.get('force_data_parallel'):
            flavour = self.config.get('data_parallel', 'buckets')
            if flavour == 'buckets':
                network = BucketedDataParallel(network, bucket_cap_mb=self.config.get('bucket_cap_mb', 25),
                                               collective=self.config.get('collective'))
            elif flavour == 'ddp':
                kwargs = dict(find_unused_parameters=self.find_unused_params, broadcast_buffers=False,
                              bucket_cap_mb=self.config.get('bucket_cap_mb', 25), gradient_as_bucket_view=True)
                if device.type == 'cuda':
                    kwargs.update(device_ids=[device.index], output_device=device.index)
                network = DDP(network, **kwargs)
                if device.type == 'cuda':
                    # torch's reducer copies a gradient into its bucket the moment autograd hands it over: nothing may be
                    # completed later (sis_hip's deferred reductions / batched weight gradients fill their tensors at the end)
                    import sis_hip
                    sis_hip.block_deferral(network)
            else:
                raise ValueError(f"data_parallel: '{flavour}' (buckets | ddp)")
        return network

    def get_network(self):
        return self.segmentation_network

    def get_networks_for_updater(self) -> Dict:
        return {'segmentation': self.segmentation_network}

    def get_optimizers(self) -> Dict:
        if self._optimizers is None:
            groups = self.parameter_groups(strip_parallel_module(self.segmentation_network))
            optimizer = FusedSGD(groups, **self.optimizer_defaults())
            # layers holding bf16 copies of their fp32 weights (TransUNet's encoder Linear layers) hand them to the
            # optimizer: its one launch per step then writes master weight and copy together
            for module in strip_parallel_module(self.segmentation_network).modules():
                register = getattr(module, 'register_weight_shadows', None)
                if register is not None:
                    register(optimizer)
            self._optimizers = {'main': optimizer}
        return self._optimizers

    def get_updater(self):
        return self.updater_class(iterators={'images': self.train_data_loader}, networks=self.get_networks_for_updater(),
                                  optimizers=self.get_optimizers(), device=self.device(),
                                  copy_to_device=(self.world_size == 1),
                                  **{'hip_graph': self.config.get('hip_graph', True), **self.updater_options()})

    def get_snapshotter(self):
        if self.rank != 0:
            return None
        return Snapshotter({'segmentation_network': self.segmentation_network, **self.get_optimizers()},
                           self.config.get('log_dir', 'logs'), self.config.get('snapshot_save_iter', 0))

    def get_evaluator(self, logger):
        return None

    def get_image_plotter(self):
        return None  # image grids are wandb / PIL visualisation, outside the training step


BaseSingleNetworkTrainBuilder = BaseTrainBuilder  # the reference's name for the single-network flavour
