"""GPU: the data-parallel training path with more than one rank on real kernels.  One-GPU boxes only allow a rehearsal --
two ranks SHARE cuda:0 and talk over gloo (RCCL refuses two ranks on one device) -- but everything else is the product
path: EMANetTrainBuilder -> DistributedDataParallel(gradient_as_bucket_view, find_unused_parameters) -> FusedSGD on the
bucket views (device kernel) -> EMANetUpdater with the step hipGraph off.  Multi-GPU throughput itself stays
"unmeasured on hardware" until the driver's SCALE run."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    for p in (ROOT, os.path.join(ROOT, "synthesis-in-style_amd")):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import yaml
        from training.fused_sgd import FusedSGD
        from training_builder.ema_net_train_builder import EMANetTrainBuilder
        from utils.synthetic_data import SyntheticSegmentationLoader
        torch.cuda.set_device(0)
        cfg = yaml.safe_load(open(os.path.join(ROOT, "synthesis-in-style_amd", "configs", "segmenter", "ema_net_resnet50_256.yaml")))
        cfg.update(fine_tune=None, batch_size=2, image_size=64)
        loader = SyntheticSegmentationLoader(2, 64, 3, seed=1234 + rank, device=torch.device("cuda:0"))
        torch.manual_seed(0)  # identical initial weights on every rank
        builder = EMANetTrainBuilder(cfg, loader, None, rank=0, world_size=world)  # rank 0 -> cuda:0 on both (shared GPU)
        net = builder.get_network()
        assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
        opt = builder.get_optimizers()["main"]
        assert isinstance(opt, FusedSGD)
        upd = builder.get_updater()
        assert not upd._step_graph.enabled  # whole-iteration capture stays off under DDP
        for _ in range(3):
            upd.update()
        torch.cuda.synchronize()
        flat = torch.cat([p.detach().flatten() for p in net.module.parameters()]).cpu()
        mu = net.module.emau.mu.detach().flatten().cpu()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        mus = [torch.zeros_like(mu) for _ in range(world)]
        dist.all_gather(mus, mu)
        out[rank] = (all(torch.equal(gathered[0], g) for g in gathered), bool(torch.isfinite(flat).all()),
                     not torch.equal(mus[0], mus[1]), upd.iteration)
    finally:
        dist.destroy_process_group()


def test_ema_net_ddp_two_ranks_share_one_gpu(device):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        lockstep, finite, mu_per_rank, iterations = out[rank]
        assert lockstep, "replicas diverged: gradients were not averaged into the buckets FusedSGD reads"
        assert finite and iterations == 3
        assert mu_per_rank, "emau.mu is a per-rank buffer (broadcast_buffers=False): different data, different bases"
