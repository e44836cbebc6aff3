"""CPU suite, part 4: the N > 1 paths, rehearsed with the gloo backend and world_size 2.

* synthesis shards by image id with no collective (utils/dataset_creation.shard_range);
* segmentation training wraps the network exactly as the builder does (DistributedDataParallel,
  broadcast_buffers=False, bucketed all-reduce): gradients on every rank equal the mean of the per-rank
  gradients, parameters stay in lock-step after an optimizer step, buffers stay per-rank.
"""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_ids():
    from utils.dataset_creation import shard_range
    for n, world in [(100000, 8), (10, 3), (7, 8), (0, 4)]:
        ranges = [shard_range(n, r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        assert max(hi - lo for lo, hi in ranges) - min(hi - lo for lo, hi in ranges) <= 1
    assert shard_range(100000, 3, 8) == (37500, 50000)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _TinySegmenter(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 3, padding=1)
        self.bn = nn.BatchNorm2d(4, momentum=3e-4)
        self.head = nn.Conv2d(4, 3, 1)
        self.register_buffer("mu", torch.zeros(1, 4))

    def forward(self, x):
        return self.head(torch.relu(self.bn(self.conv(x))))


def _worker(rank, world, port, out, flavour):
    sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from training.grad_exchange import BucketedDataParallel
        from training_builder.base_train_builder import BaseSingleNetworkTrainBuilder, strip_parallel_module
        torch.manual_seed(rank)  # DIFFERENT initial weights per rank: the wrap must broadcast rank 0's
        builder = BaseSingleNetworkTrainBuilder({"fine_tune": None, "bucket_cap_mb": 1, "data_parallel": flavour}, rank=rank,
                                                world_size=world, build=False)
        net = _TinySegmenter()
        with torch.no_grad():
            net.mu.fill_(float(rank))  # per-rank buffer must survive (broadcast_buffers=False)
        ddp = builder._prepare_segmentation_network(net)
        assert isinstance(ddp, BucketedDataParallel if flavour == "buckets" else nn.parallel.DistributedDataParallel)
        assert strip_parallel_module(ddp) is net
        gen = torch.Generator().manual_seed(100 + rank)
        x = torch.randn(2, 3, 8, 8, generator=gen)
        y = torch.randint(0, 3, (2, 8, 8), generator=gen)
        loss = nn.functional.cross_entropy(ddp(x), y)
        loss.backward()
        grads = torch.cat([p.grad.flatten() for p in net.parameters()])
        # single-process replay of this rank's gradient, then average across ranks by hand
        torch.manual_seed(0)
        solo = _TinySegmenter()
        nn.functional.cross_entropy(solo(x), y).backward()
        local = torch.cat([p.grad.flatten() for p in solo.parameters()])
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        expect = torch.stack(gathered).mean(0)
        opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
        opt.step()
        params = torch.cat([p.detach().flatten() for p in net.parameters()])
        all_params = [torch.zeros_like(params) for _ in range(world)]
        dist.all_gather(all_params, params)
        out[rank] = (torch.allclose(grads, expect, atol=1e-6), all(torch.equal(all_params[0], q) for q in all_params),
                     float(net.mu[0, 0]), float(net.bn.running_mean.abs().sum()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("flavour", ["buckets", "ddp"])
def test_ddp_wrap_world_size_2_gloo(flavour):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, flavour), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        grads_ok, params_ok, mu, bn_stat = out[rank]
        assert grads_ok, "DDP gradients are not the mean of the per-rank gradients"
        assert params_ok, "parameters diverged across ranks after one step"
        assert mu == float(rank), "buffers were broadcast (must stay per-rank)"
    assert out[0][3] != out[1][3], "batch-norm statistics must be per-rank"


class _TinyEmaShaped(nn.Module):
    """CPU stand-in with EMANet's optimizer-relevant shape: conv weights / norm scales / biases in three groups and
    one parameter that never receives a gradient (EMANet's ``emau.conv1``: find_unused_parameters)."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 8, 3, padding=1, bias=False)
        self.bn = nn.BatchNorm2d(8)
        self.unused = nn.Conv2d(8, 8, 1)
        self.head = nn.Conv2d(8, 3, 1)

    def forward(self, x):
        return self.head(torch.relu(self.bn(self.conv(x))))


def _host_sgd_kernel(table, n_chunks, lrs, wds, momentum, first_step):
    """Host stand-in for ``sis_sgd_momentum`` (csrc/seg_ops.hip ``sgd_kernel``): walks the SAME chunk table the device
    kernel gets -- raw (param, grad, momentum) addresses + count | group << 48 -- and applies torch.optim.SGD's update."""
    import ctypes
    import numpy as np
    rows = table.numpy()
    for p_ptr, g_ptr, b_ptr, packed, _shadow in rows[:n_chunks]:
        n, gi = int(packed) & ((1 << 48) - 1), int(packed) >> 48
        view = lambda ptr: np.ctypeslib.as_array((ctypes.c_float * n).from_address(int(ptr)))  # noqa: E731
        p, g, b = view(p_ptr), view(g_ptr), view(b_ptr)
        d = g + np.float32(wds[gi]) * p
        b[:] = d if first_step else np.float32(momentum) * b + d
        p -= np.float32(lrs[gi]) * b


def _fused_sgd_worker(rank, world, port, out, flavour="ddp"):
    sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sis_hip
        from training.fused_sgd import FusedSGD
        from training_builder.base_train_builder import BaseTrainBuilder
        from updater.segmentation_updater import _graphable
        sis_hip.sgd_momentum = _host_sgd_kernel          # the kernel only; table / bucket logic below is the product's
        sis_hip.require_device = lambda t, name: None
        sis_hip.sgd_chunk_elems = lambda: 64             # several chunks per tensor

        class Builder(BaseTrainBuilder):  # EMANetTrainBuilder's declarations on the stand-in network
            find_unused_params = True

            def build_network(self):
                torch.manual_seed(0)
                return _TinyEmaShaped()

            def parameter_groups(self, network):
                conv = [network.conv.weight, network.unused.weight, network.head.weight]
                return [{'params': conv, 'lr': 0.05, 'weight_decay': 1e-2}, {'params': [network.bn.weight], 'lr': 0.05,
                        'weight_decay': 0.0}, {'params': [network.bn.bias, network.unused.bias, network.head.bias],
                                               'lr': 0.1, 'weight_decay': 0.0}]

            def optimizer_defaults(self):
                return {'momentum': 0.9}

        from training.grad_exchange import BucketedDataParallel
        # buckets flavour: a 100-byte cap cuts the five live tensors (12, 96, 32 + 32, 864 bytes in readiness order) into four buckets
        builder = Builder({"fine_tune": None, "bucket_cap_mb": 1 if flavour == "ddp" else 100 / (1 << 20), "data_parallel": flavour},
                          rank=rank, world_size=world)
        ddp = builder.get_network()
        net = ddp.module
        opt = builder.get_optimizers()['main']
        assert isinstance(opt, FusedSGD)
        assert isinstance(ddp, nn.parallel.DistributedDataParallel if flavour == "ddp" else BucketedDataParallel)
        graph_off = not _graphable(ddp, opt, 'cuda:0')   # eager under DDP, and under gloo (host-synchronising collectives)
        # reference run: same data, plain torch.optim.SGD on the hand-averaged gradients
        torch.manual_seed(0)
        solo = _TinyEmaShaped()
        ref_opt = torch.optim.SGD(Builder.parameter_groups(builder, solo), momentum=0.9)
        uploads, table_ok, alias_ok, ptrs, spans = [], True, True, [], []

        def recording_allreduce(state, bucket):  # the default hook's arithmetic, plus a record of the bucket's storage
            buf = bucket.buffer()
            state.append((buf.data_ptr(), buf.data_ptr() + 4 * buf.numel()))
            fut = dist.all_reduce(buf.div_(world), async_op=True).get_future()
            return fut.then(lambda f: f.value()[0])

        if flavour == "ddp":
            ddp.register_comm_hook(spans, recording_allreduce)
        orig_upload = opt._upload
        opt._upload = lambda entries, capturing=False: (uploads.append(1), orig_upload(entries, capturing))[1]
        for it in range(3):
            gen = torch.Generator().manual_seed(100 * it + rank)
            x, y = torch.randn(2, 3, 8, 8, generator=gen), torch.randint(0, 3, (2, 8, 8), generator=gen)
            opt.zero_grad()  # set_to_none: DDP re-points .grad at the bucket views during backward
            nn.functional.cross_entropy(ddp(x), y).backward()
            live = [p for p in net.parameters() if p.grad is not None]
            if flavour == "buckets":
                spans = ddp.bucket_spans()
                assert len(spans) >= 3 and ddp.stats["discovery_backwards"] == 1
            assert net.unused.weight.grad is None and len(live) == 5 and spans
            alias_ok &= all(any(lo <= p.grad.data_ptr() < hi for lo, hi in spans) for p in live)
            ptrs.append(tuple(p.grad.data_ptr() for p in live))
            opt.step()
            rows = opt._table.numpy()
            first_chunk = {int(r[0]): int(r[1]) for r in rows}  # param address -> grad address of each tensor's chunk 0
            table_ok &= all(first_chunk[p.data_ptr()] == p.grad.data_ptr() for p in live)
            # reference: average of both ranks' local gradients
            ref_opt.zero_grad()
            for r in range(world):
                gen_r = torch.Generator().manual_seed(100 * it + r)
                xr, yr = torch.randn(2, 3, 8, 8, generator=gen_r), torch.randint(0, 3, (2, 8, 8), generator=gen_r)
                solo.bn.train()
                (nn.functional.cross_entropy(solo(xr), yr) / world).backward()
            ref_opt.step()
        close = all(torch.allclose(a, b, atol=2e-6) for a, b in zip(net.parameters(), solo.parameters()))
        flat = torch.cat([p.detach().flatten() for p in net.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        out[rank] = dict(graph_off=graph_off, uploads=len(uploads), stable=len(set(ptrs)) == 1, table_ok=bool(table_ok),
                         alias_ok=bool(alias_ok), close=bool(close), lockstep=all(torch.equal(gathered[0], q) for q in gathered),
                         untouched=bool(torch.equal(net.unused.weight, _TinyEmaShaped_init_unused())))
    finally:
        dist.destroy_process_group()


def _TinyEmaShaped_init_unused():
    torch.manual_seed(0)
    return _TinyEmaShaped().unused.weight.detach()


@pytest.mark.parametrize("flavour", ["buckets", "ddp"])
def test_fused_sgd_on_ddp_bucket_views_world_size_2_gloo(flavour):
    """(``buckets``: the product's own exchange, training/grad_exchange.py -- plan fixed by the first backward, several buckets,
    the unused parameter never enters a bucket; ``ddp``: torch's reducer.)  EMANetTrainBuilder's shape under DistributedDataParallel(gradient_as_bucket_view=True, find_unused_parameters=True)
    with the product's FusedSGD: gradients live in the all-reduce buckets at stable addresses (ONE pointer-table upload for
    three iterations), the table's gradient column points into the buckets, the update equals torch.optim.SGD on the
    rank-averaged gradients (3 groups: lr / 2 lr / weight decay), the gradient-less parameter is skipped like
    torch.optim.SGD skips it, replicas stay in lock-step, and the step hipGraph is off under DDP.  Only the kernel is a
    host stand-in (it consumes the same chunk table).  Unmeasured on hardware until a multi-GPU SCALE record exists."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_fused_sgd_worker, args=(world, _free_port(), out, flavour), nprocs=world, join=True)
    assert len(out) == world
    for rank in range(world):
        r = out[rank]
        assert r["graph_off"], "hipGraph capture must be disabled under DistributedDataParallel"
        assert r["stable"] and r["uploads"] == 1, f"gradient storage moved between iterations: {r}"
        assert r["alias_ok"] and r["table_ok"], f"pointer table does not alias the DDP buckets: {r}"
        assert r["close"], "FusedSGD on bucket views differs from torch.optim.SGD on the averaged gradients"
        assert r["lockstep"] and r["untouched"]


class _TinyGenerator(nn.Module):
    """CPU stand-in with the call surface Stylegan2Updater uses: ``g(styles, noise=..., return_latents=...)`` ->
    (image, latents | None), ``noises`` buffers, differentiable from the latents to the image."""

    def __init__(self, dim=8, n_latent=4):
        super().__init__()
        self.n_latent = n_latent
        self.mapping = nn.Linear(dim, dim)
        self.synthesis = nn.Linear(dim, 3 * 4 * 4)
        self.noises = nn.Module()
        self.noises.register_buffer("noise_0", torch.zeros(1, 1, 4, 4))

    def forward(self, styles, return_latents=False, noise=None, **_):
        w = [self.mapping(s) for s in styles]
        latent = w[0].unsqueeze(1).repeat(1, self.n_latent, 1) if len(w) == 1 else torch.cat(
            [w[0].unsqueeze(1).repeat(1, 2, 1), w[1].unsqueeze(1).repeat(1, self.n_latent - 2, 1)], 1)
        image = torch.tanh(self.synthesis(latent.mean(1))).view(-1, 3, 4, 4)
        return image, (latent if return_latents else None)


class _TinyDiscriminator(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 4, 3, padding=1)
        self.out = nn.Linear(4 * 4 * 4, 1)

    def forward(self, x):
        return self.out(nn.functional.leaky_relu(self.conv(x), 0.2).flatten(1))


def _gan_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "synthesis-in-style_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import copy
        import random
        from torch.nn.parallel import DistributedDataParallel as DDP
        from updater.stylegan_2_updater import Stylegan2Updater
        torch.manual_seed(0)  # identical initial weights on every rank
        g, d = _TinyGenerator(), _TinyDiscriminator()
        g_ema = copy.deepcopy(g)
        opts = {"generator": torch.optim.Adam(g.parameters(), lr=1e-2, betas=(0.0, 0.99)),
                "discriminator": torch.optim.Adam(d.parameters(), lr=1e-2, betas=(0.0, 0.99))}
        networks = {"generator": DDP(g, broadcast_buffers=False), "discriminator": DDP(d, broadcast_buffers=False)}
        torch.manual_seed(100 + rank)  # per-rank data and latents from here on
        random.seed(7)  # the style-mixing coin must agree across ranks: it decides which parameters take part

        def batches():
            while True:
                yield {"image": torch.rand(4, 3, 4, 4) * 2 - 1}

        up = Stylegan2Updater(iterators={"images": batches()}, networks=networks, optimizers=opts, device="cpu", g_ema=g_ema,
                              latent_size=8, regularization_options={"d_reg_interval": 2, "g_reg_interval": 2})
        up.accumulate(networks["generator"], 0)
        for _ in range(4):  # iterations 0 and 2 run both lazy regularisers (second-order backward through DDP)
            up.update()
        flat = torch.cat([p.detach().flatten() for net in (g, d) for p in net.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ema_start = [p.detach().clone() for p in _TinyGenerator().parameters()]  # a fresh init differs from the average
        ema_moved = any(not torch.equal(a, b) for a, b in zip(g_ema.parameters(), ema_start))
        out[rank] = (all(torch.allclose(gathered[0], q, atol=1e-6) for q in gathered), float(up.mean_path_length_avg),
                     bool(torch.isfinite(flat).all()), all(p.requires_grad for p in list(g.parameters()) + list(d.parameters())),
                     bool(ema_moved))
    finally:
        dist.destroy_process_group()


def test_stylegan2_updater_world_size_2_gloo():
    """The GAN iteration under DistributedDataParallel (gloo, CPU stand-in networks): freezing one network per sub-step
    (UpdateDisabler) and the lazy regularisers' second-order backward leave the replicas in lock-step, and the mean path
    length is averaged over ranks (the updater's only explicit collective, reduce_sum)."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gan_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert len(out) == world
    assert all(out[r][0] for r in range(world)), "replicas diverged"
    assert out[0][1] == pytest.approx(out[1][1], rel=1e-6) and out[0][1] > 0, "mean path length is not the all-rank average"
    assert all(out[r][2] and out[r][3] for r in range(world))


def test_shard_plan_matches_the_in_place_reduce_scatter_all_gather_convention():
    """The shard arithmetic of the direct RCCL path (training/grad_exchange.py::shard_plan / shard_span) for world sizes 1..8,
    checked against a host model of the two collectives as nccl.h defines their in-place forms: ncclReduceScatter(send, recv =
    send + rank * count) leaves rank r with the reduction of everybody's elements [r * count, (r + 1) * count) at that offset;
    ncclAllGather(send = recv + rank * count, recv) then gives every rank every shard.  A one-GPU box can never run this path
    with more than one rank, so what CAN be pinned is: every rank's shard has the same length, starts 16-byte aligned, the
    shards tile the padded bucket exactly, and the composition is the mean of the ranks' buckets (padding included)."""
    import numpy as np
    from training.grad_exchange import shard_plan, shard_span
    rng = np.random.RandomState(5)
    for world in range(1, 9):
        for numel in (1, 3, 4 * world, 4 * world + 1, 4099, 6553601):
            padded, per = shard_plan(numel, world)
            assert padded >= numel and padded - numel < 4 * world and padded == per * world and per % 4 == 0
            spans = [shard_span(r, per) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == padded
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:])) and all((4 * b) % 16 == 0 for b, _ in spans)
            if numel > 5000:
                continue
            send = [rng.randn(padded).astype(np.float32) for _ in range(world)]     # each rank's bucket before the exchange
            mean = np.mean(np.stack(send).astype(np.float64), axis=0)
            buf = [s.copy() for s in send]
            for r in range(world):                                                  # reduce-scatter, in place
                b, e = shard_span(r, per)
                buf[r][b:e] = np.mean(np.stack([s[b:e] for s in send]).astype(np.float64), axis=0).astype(np.float32)
            shards = [buf[r][slice(*shard_span(r, per))].copy() for r in range(world)]
            for r in range(world):                                                  # all-gather, in place
                for q in range(world):
                    buf[r][slice(*shard_span(q, per))] = shards[q]
            for r in range(world):
                np.testing.assert_allclose(buf[r], mean, rtol=1e-6, atol=1e-7)
    with pytest.raises(ValueError):
        shard_plan(0, 2)


def test_direct_rccl_switch_positions():
    """Which configurations take the direct librccl.so path: by default world size 1 only (until a multi-rank run
    has exercised it, world > 1 goes through torch.distributed); SIS_DP_DIRECT_RCCL=1 opts every world size in, 0 none; and a
    capture of the exchange additionally needs SIS_DP_GRAPH (auto: world size 1) -- work objects (direct off) never are."""
    import training.grad_exchange as GX

    class Fake:
        _on_gpu, backend = True, "nccl"
        capturable = GX.BucketedDataParallel.capturable

    fake = Fake()
    saved = GX._DP_GRAPH
    try:
        for graph, world, comm, want in (("auto", 1, 1, True), ("auto", 2, 1, False), ("1", 2, 1, True), ("0", 1, 1, False),
                                         ("1", 1, None, False), ("auto", 1, None, False)):
            GX._DP_GRAPH, fake.world, fake._comm = graph, world, comm
            assert fake.capturable() is want, (graph, world, comm)
    finally:
        GX._DP_GRAPH = saved
    src = open(GX.__file__).read()
    assert 'os.environ.get("SIS_DP_DIRECT_RCCL", "auto")' in src and '_DIRECT_RCCL == "auto" and self.world == 1' in src


def test_gradient_arena_hands_out_bucket_slices():
    """sis_hip's gradient arena (host logic, CPU tensors): after the data-parallel wrap has registered the bucket slices, a
    weight-gradient binding asking ``grad_out(param.data_ptr(), ...)`` gets a NEW view of the parameter's slice (autograd then
    takes it as ``.grad`` without a copy); it gets a fresh tensor when the parameter already holds a gradient (a second
    backward must accumulate), when the shape / dtype does not match, or for an unknown address; the fused query | key | value
    result is served only when the three slices lie back to back in that order; releasing the owner empties the registry."""
    import sis_hip
    owner = object()
    flat = torch.zeros(64)
    a, b, c = (nn.Parameter(torch.randn(2, 4)) for _ in range(3))
    sis_hip.grad_arena_register(owner, [a, b, c], [flat, flat, flat], [8, 16, 24])
    try:
        g = sis_hip.grad_out(a.data_ptr(), (2, 4), torch.float32, flat.device)
        assert g.data_ptr() == flat.data_ptr() + 4 * 8 and g.shape == (2, 4)
        # handed out: a second request in the same backward (a parameter with two uses in the graph, whose second kernel may run
        # before autograd has accumulated the first result) gets a FRESH tensor, never the same slice again
        t = sis_hip.grad_out(a.data_ptr(), (2, 4), torch.float32, flat.device)
        assert not (flat.data_ptr() <= t.data_ptr() < flat.data_ptr() + 4 * 64)
        sis_hip.grad_arena_reset(owner)                                                               # next backward
        g2 = sis_hip.grad_out(a.data_ptr(), (2, 4), torch.float32, flat.device)
        assert g2.data_ptr() == g.data_ptr() and g2 is not g                                          # a new view object per call
        sis_hip.grad_arena_reset(owner)
        g.fill_(3.0)
        assert float(flat[8:16].sum()) == 24.0 and float(flat.sum()) == 24.0
        for key, shape, dtype in ((a.data_ptr(), (4, 4), torch.float32), (a.data_ptr(), (2, 4), torch.float64),
                                  (12345, (2, 4), torch.float32), (None, (2, 4), torch.float32)):
            t = sis_hip.grad_out(key, shape, dtype, flat.device)
            assert not (flat.data_ptr() <= t.data_ptr() < flat.data_ptr() + 4 * 64)
        a.grad = torch.ones(2, 4)                                                                    # holds a gradient: accumulate
        t = sis_hip.grad_out(a.data_ptr(), (2, 4), torch.float32, flat.device)
        assert not (flat.data_ptr() <= t.data_ptr() < flat.data_ptr() + 4 * 64)
        a.grad = None
        fused = sis_hip.grad_out_fused((a.data_ptr(), b.data_ptr(), c.data_ptr()), [2, 2, 2], 4, flat.device)
        assert fused.data_ptr() == flat.data_ptr() + 4 * 8 and fused.shape == (6, 4)
        sis_hip.grad_arena_reset(owner)
        swapped = sis_hip.grad_out_fused((b.data_ptr(), a.data_ptr(), c.data_ptr()), [2, 2, 2], 4, flat.device)
        assert not (flat.data_ptr() <= swapped.data_ptr() < flat.data_ptr() + 4 * 64)
        # a parameter used TWICE in one graph: both results arrive, one through the slice, one fresh, and autograd adds them
        class Twice(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.key = w.data_ptr()
                return x * w.sum()
            @staticmethod
            def backward(ctx, gy):
                dw = sis_hip.grad_out(ctx.key, (2, 4), torch.float32, gy.device)
                dw.fill_(float(gy.sum()))
                return None, dw
        (Twice.apply(torch.ones(3), c).sum() + Twice.apply(2 * torch.ones(5), c).sum()).backward()
        assert torch.equal(c.grad, torch.full((2, 4), 8.0))   # (the engine sums the two into a new tensor: the wrap's gather copies it)
        sis_hip.grad_arena_reset(owner)
        # autograd's side of the contract: a gradient returned as such a view becomes .grad as it is (no clone)
        class Fn(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.key = w.data_ptr()
                return x * w.sum()
            @staticmethod
            def backward(ctx, gy):
                dw = sis_hip.grad_out(ctx.key, (2, 4), torch.float32, gy.device)
                dw.fill_(float(gy.sum()))
                return None, dw
        Fn.apply(torch.ones(3), b).sum().backward()
        assert b.grad.data_ptr() == flat.data_ptr() + 4 * 16 and float(flat[16:24].sum()) == 24.0
    finally:
        sis_hip.grad_arena_release(owner)
    assert sis_hip.grad_arena_slot(a.data_ptr()) is None
