"""GPU: a training iteration replayed as a hipGraph (training/graph_step.py) must reproduce the eager iterations:
same losses, same parameters, and a host-side learning-rate change between replays must take effect."""
import os

import numpy as np
import pytest
import torch

from oracle import ema_net_ref as E

pytestmark = pytest.mark.gpu


def test_graph_replay_matches_eager(device, golden_dir):
    """Two iterations from one saved state, once eager and once as capture + replays (EMANet-50 at a random init is
    chaotic across runs -- DESIGN.md §2 -- so whole trajectories cannot be compared; single steps from the same state
    can).  The second iteration runs at a quarter of the learning rate, as an LR scheduler would set it."""
    from networks.ema_net.network import EMANet
    from networks.ema_net.utils import get_params
    from training.fused_sgd import FusedSGD
    from training.loop import get_current_reporter
    from updater.segmentation_updater import EMANetUpdater
    g = np.load(os.path.join(golden_dir, "ema_net_step.npz"))
    n_layers, classes, wseed, bseed, batch, size = g["cfg"].tolist()
    net = EMANet(3, 50, use_pretrained_resnet=False)
    net.load_state_dict(E.seeded_state_dict(50, 3, seed=wseed), strict=True)
    net.fc1[1].p = 0.0
    net = net.to(device).train()
    lr = 2e-5
    opt = FusedSGD([{"params": list(get_params(net, "1x")), "lr": lr, "weight_decay": 1e-4},
                    {"params": list(get_params(net, "1y")), "lr": lr, "weight_decay": 0},
                    {"params": list(get_params(net, "2x")), "lr": 2 * lr, "weight_decay": 0.0}], momentum=0.9)
    b = [E.seeded_batch(batch, size, classes, seed=bseed + i) for i in range(5)]
    upd = EMANetUpdater(em_mom=0.9, iterators={"images": [b[0], b[1], b[3], b[4], b[3], b[4]]},
                        networks={"segmentation": net}, optimizers={"main": opt}, device=device)
    assert upd._step_graph.enabled
    watched = ("fc2.weight", "fc2.bias", "fc1.0.bn.weight", "emau.mu", "extractor.7.2.conv3.weight")

    def loss():
        return float(get_current_reporter().scalars()["loss/softmax"])

    def two_steps():
        out = []
        for scale in (1.0, 0.25):
            for group, base in zip(opt.param_groups, (lr, lr, 2 * lr)):
                group["lr"] = base * scale
            upd.update()
            out.append((loss(), {k: net.state_dict()[k].detach().cpu().clone() for k in watched}))
        return out

    for _ in range(2):
        upd.update()  # eager warm-up iterations
    assert upd._step_graph.graph is None
    saved_model = {k: v.detach().clone() for k, v in net.state_dict().items()}
    saved_mom = {p: opt.state[p]["momentum_buffer"].clone() for p in opt.state}
    upd._step_graph.enabled = False
    eager = two_steps()
    net.load_state_dict(saved_model, strict=True)
    for p, m in saved_mom.items():
        opt.state[p]["momentum_buffer"].copy_(m)
    upd._step_graph.enabled = True
    graphed = two_steps()
    assert upd._step_graph.graph is not None, "the iteration was never captured"

    start = {k: saved_model[k].cpu() for k in watched}
    np.testing.assert_allclose(graphed[0][0], eager[0][0], rtol=1e-4)
    np.testing.assert_allclose(graphed[1][0], eager[1][0], rtol=2e-2)
    for step, tol in ((0, 2e-2), (1, 5e-2)):
        for k in watched:
            ref = (eager[step][1][k] - start[k]).numpy()
            got = (graphed[step][1][k] - start[k]).numpy()
            # deltas of a 2e-5 learning rate sit a few ulps above the weights' own rounding
            atol = tol * np.abs(ref).max() + 4e-7 * float(start[k].abs().max())
            np.testing.assert_allclose(got, ref, rtol=tol, atol=atol, err_msg=f"step {step} {k}")


def test_lr_change_reaches_the_captured_optimizer(device):
    """FusedSGD on a capturing stream reads lr / weight decay / momentum from device memory."""
    from training.fused_sgd import FusedSGD
    p = torch.nn.Parameter(torch.ones(1000, device=device))
    opt = FusedSGD([p], lr=0.5, momentum=0.0)
    p.grad = torch.ones_like(p)
    opt.step()  # eager: p = 0.5, momentum buffers created
    opt.push_hyper()
    static_grad = torch.ones_like(p)
    p.grad = static_grad
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step()
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(p.detach().cpu().numpy(), 0.0, atol=1e-7)
    opt.param_groups[0]["lr"] = 0.125
    opt.push_hyper()
    graph.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(p.detach().cpu().numpy(), -0.125, atol=1e-7)


def test_partial_batch_runs_eagerly_and_graph_stays_valid(device):
    """a batch whose shape differs from the captured one (last batch of an epoch) must neither raise from the
    static-input copy nor be broadcast into it; it runs eagerly, later full batches replay the graph again, and the
    parameters equal an all-eager run."""
    from training.fused_sgd import FusedSGD
    from updater.segmentation_updater import _GraphedUpdater
    gen = torch.Generator().manual_seed(5)
    sizes = [4, 4, 4, 4, 2, 4, 1, 4]
    batches = [{"images": torch.randn(n, 16, generator=gen), "segmented": torch.randn(n, 8, generator=gen)} for n in sizes]

    class Tiny(_GraphedUpdater):
        def _iteration(self, batch):
            net, opt = self.networks["segmentation"], self.optimizers["main"]
            loss = ((net(batch["images"]) - batch["segmented"]) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            return {"mse": loss.detach()}

    def run(hip_graph):
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.Tanh(), torch.nn.Linear(32, 8)).to(device)
        opt = FusedSGD(list(net.parameters()), lr=0.05, momentum=0.9)
        upd = Tiny(iterators={"images": batches}, networks={"segmentation": net}, optimizers={"main": opt}, device=device,
                   hip_graph=hip_graph)
        for _ in sizes:
            upd.update()
        torch.cuda.synchronize()
        return [p.detach().cpu().clone() for p in net.parameters()], upd

    eager, _ = run(False)
    graphed, upd = run(True)
    assert upd._step_graph.graph is not None and upd.iteration == len(sizes)
    for a, b in zip(eager, graphed):
        np.testing.assert_allclose(b.numpy(), a.numpy(), rtol=1e-5, atol=1e-6)
