"""Segmentation training entry point: one process per GPU, data-parallel over RCCL.

Mirrors stylegan_code_finder/train.py:59-187 for the networks on the MI355X hot path (EMANet, TransUNet):
config (YAML merged with CLI flags) -> process group -> loaders -> train builder -> updater loop with a
per-iteration clamped-cosine LR, rank-0 snapshots, barrier before the first iteration.

Launch either like the reference (``python train.py CONFIG ...`` spawns ``torch.cuda.device_count()`` workers)
or under ``torch.distributed.run`` (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* taken from the environment).
The third-party trainer / wandb / image-plotter extensions of the reference are not part of the step and are
replaced by a plain loop with stdout logging.
"""
import argparse
import json
import logging
import os
import time

import torch
import torch.distributed as dist
import yaml

from training.loop import get_current_reporter
from training_builder.train_builder_selection import get_train_builder_class
from utils.clamped_cosine import ClampedCosineAnnealingLR
from utils.synthetic_data import SyntheticSegmentationLoader


def load_yaml_config(path):
    with open(path) as f:
        return yaml.safe_load(f)


def merge_config_and_args(config: dict, args: argparse.Namespace) -> dict:
    for key, value in vars(args).items():
        if not key.startswith('_') and (value is not None or key not in config):
            config[key] = value
    return config


def sanity_check_config(config: dict):
    assert config.get('network') in ('TransUNet', 'EMANet'), 'The network must be one of: TransUNet, EMANet'
    if config.get('class_to_color_map'):
        with open(config['class_to_color_map']) as f:
            assert len(json.load(f)) == config['num_classes'], \
                'The number of classes in the class_to_color_map must be equal to the num_classes in the config'


def setup_distributed(mpi_backend: str, rank: int, world_size: int):
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '12355')
    kwargs = {}
    if mpi_backend == 'nccl' and torch.cuda.is_available():
        kwargs['device_id'] = torch.device('cuda', rank % torch.cuda.device_count())
    dist.init_process_group(mpi_backend, rank=rank, world_size=world_size, **kwargs)


def get_scheduler(config: dict, iterations_per_epoch: int, optimizers: dict):
    if 'cosine_max_update_epoch' in config:
        end = config['cosine_max_update_epoch'] * iterations_per_epoch
    elif 'cosine_max_update_iter' in config:
        end = config['cosine_max_update_iter']
    else:
        end = config['epochs']
    eta_min = config.get('end_lr', 0.0)
    if config.get('warm_restarts'):  # train.py:50-52 of the reference
        return {name: torch.optim.lr_scheduler.CosineAnnealingWarmRestarts(opt, end, eta_min=eta_min)
                for name, opt in optimizers.items()}
    return {name: ClampedCosineAnnealingLR(opt, end, eta_min=eta_min) for name, opt in optimizers.items()}


def get_data_loader(config: dict, rank: int, device):
    if config.get('train_json') and not config.get('synthetic'):
        raise NotImplementedError("the JSON/PNG + imgaug input pipeline is host-side and out of the hot path; "
                                  "pass --synthetic or supply your own iterable of {'images','segmented'} batches")
    return SyntheticSegmentationLoader(config['batch_size'], config['image_size'], config['num_classes'],
                                       seed=1234 + rank, device=device, num_batches=config.get('iterations_per_epoch'))


def main(rank: int, args: argparse.Namespace, world_size: int):
    config = merge_config_and_args(load_yaml_config(args.config), args)
    sanity_check_config(config)
    if world_size > 1:
        setup_distributed(args.mpi_backend, rank, world_size)
    device = torch.device('cuda', rank % torch.cuda.device_count()) if torch.cuda.is_available() else torch.device('cpu')
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    loader = get_data_loader(config, rank, device)
    builder = get_train_builder_class(config)(config, loader, None, rank=rank, world_size=world_size)
    updater = builder.get_updater()
    per_epoch = config.get('iterations_per_epoch') or len(loader)
    max_iter = config['max_iter'] if 'max_iter' in config and config['max_iter'] else config['epochs'] * per_epoch
    schedulers = get_scheduler(config, per_epoch, builder.get_optimizers())
    snapshotter = builder.get_snapshotter()
    if world_size > 1:
        dist.barrier()
    logging.info('Setup complete. Starting training...')
    t0 = time.perf_counter()
    try:
        for it in range(1, max_iter + 1):
            updater.update()
            for sched in schedulers.values():
                sched.step()
            if snapshotter is not None:
                snapshotter.maybe_save(it)
            if rank == 0 and it % config.get('log_iter', 10) == 0:
                obs = get_current_reporter().scalars()
                rate = it * config['batch_size'] * world_size / (time.perf_counter() - t0)
                print(f"iter {it} " + " ".join(f"{k}={v:.5f}" for k, v in obs.items()) + f" images/s={rate:.1f}",
                      flush=True)
    finally:
        if world_size > 1:
            dist.destroy_process_group()
    logging.info('Training finished')


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description='Train a network for semantic segmentation of documents')
    parser.add_argument('config', help='path to config with common train settings, such as LR')
    parser.add_argument('--images', dest='train_json', help='path to json file with train images')
    parser.add_argument('--val-images', dest='validation_json', help='path to json file with validation images')
    parser.add_argument('--mpi-backend', default='nccl', choices=['nccl', 'gloo'], help='torch.distributed backend')
    parser.add_argument('--fine-tune', dest='fine_tune', help='Path to model to finetune from')
    parser.add_argument('-l', '--log-dir', default='logs', help='where to write snapshots')
    parser.add_argument('--synthetic', action='store_true', help='feed synthetic batches (benchmarks / smoke runs)')
    parser.add_argument('--max-iter', dest='max_iter', type=int, help='stop after this many iterations')
    return parser.parse_args(argv)


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO)
    cli = parse_args()
    if 'RANK' in os.environ:  # launched by torch.distributed.run: one process per GPU already
        main(int(os.environ['RANK']), cli, int(os.environ.get('WORLD_SIZE', '1')))
    else:
        n = max(torch.cuda.device_count(), 1)
        if n > 1:
            torch.multiprocessing.spawn(main, args=(cli, n), nprocs=n)
        else:
            main(0, cli, 1)
