"""GPU parity, part 4: the EMANet training step (product: networks/ema_net + updater + FusedSGD on MI355X)
against the golden outputs of the reference (tests/golden/ema_net_step.npz) and the CPU oracle.

Tolerances.  Every convolution of the step except the 3-channel stem runs on this repo's own fp32 kernels (Winograd
F(2x2,3x3) forward / data / weight gradient, MFMA 1x1 kernels), which sum in a different order than the oracle's direct
CPU convolution, and the Winograd transforms add ~1e-6 relative error per layer; on the plain random init of this fixture
the 50-layer ReLU network amplifies such differences (a 1e-7 perturbation moves the stem gradient by 2-3 %
between the CPU oracle and the CPU reference themselves).  Hence: first-iteration loss 1e-4, every gradient norm 2 %,
label maps bit-exact wherever the reference's top-2 logit margin exceeds 5e-2 (twice the logit tolerance).  The
well-conditioned fixture (``test_ema_net_conditioned_fixture_tight``) is where the bounds are tight: loss 1e-5, gradient
norms 3e-3; the BASELINE batch size (B = 16, where the single-launch batch-norm kernels dispatch) is covered by
``test_ema_net_step_at_baseline_batch_vs_oracle``.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ema_net_ref as E

pytestmark = pytest.mark.gpu


def _net(device, wseed):
    from networks.ema_net.network import EMANet
    net = EMANet(3, 50, use_pretrained_resnet=False)
    net.load_state_dict(E.seeded_state_dict(50, 3, seed=wseed), strict=True)
    net.fc1[1].p = 0.0
    return net.to(device).train()


def test_ema_net_first_iteration_vs_golden(device, golden_dir):
    from networks.ema_net.utils import get_params
    from training.fused_sgd import FusedSGD
    from training.loop import get_current_reporter
    from updater.segmentation_updater import EMANetUpdater
    g = np.load(os.path.join(golden_dir, "ema_net_step.npz"))
    n_layers, classes, wseed, bseed, batch, size = g["cfg"].tolist()
    net = _net(device, wseed)
    assert (len(list(get_params(net, "1x"))), len(list(get_params(net, "1y"))), len(list(get_params(net, "2x")))) == (60, 58, 60)
    lr = 2e-5  # see tests/golden/make_golden_seg.py: at the config's 0.009 the second iteration is chaotic
    opt = FusedSGD([{"params": list(get_params(net, "1x")), "lr": lr, "weight_decay": 1e-4},
                    {"params": list(get_params(net, "1y")), "lr": lr, "weight_decay": 0},
                    {"params": list(get_params(net, "2x")), "lr": 2 * lr, "weight_decay": 0.0}], momentum=0.9)
    batches = [E.seeded_batch(batch, size, classes, seed=bseed + i) for i in range(2)]
    upd = EMANetUpdater(em_mom=0.9, iterators={"images": batches}, networks={"segmentation": net},
                        optimizers={"main": opt}, device=device)
    # iteration 1, instrumented: same order as update_core
    b0 = {k: v.to(device) for k, v in batches[0].items()}
    loss, mu = net(b0["images"], b0["segmented"].squeeze(1))
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g["loss"], rtol=1e-4)
    np.testing.assert_allclose(mu[:, ::32, ::8].cpu().numpy(), g["mu_slice"], rtol=2e-3, atol=1e-5)
    loss.mean().backward()
    grads = {n: p.grad for n, p in net.named_parameters()}
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        if ref < 0:
            assert grads[str(name)] is None
        else:
            np.testing.assert_allclose(grads[str(name)].double().norm().item(), ref, rtol=2e-2, err_msg=str(name))
    np.testing.assert_allclose(grads["fc2.weight"].cpu().numpy(), g["grad_fc2_weight"], rtol=5e-3,
                               atol=2e-3 * np.abs(g["grad_fc2_weight"]).max())
    # the stem gradient is the end of a 50-layer backward chain through batch-of-2 BN and hard EM assignments:
    # the CPU oracle and the CPU reference themselves drift apart by 2-3 % there after one 1e-7 perturbation
    # (tests/test_segmentation_oracle_cpu.py), so individual elements get 5 % of the largest one
    np.testing.assert_allclose(grads["extractor.0.0.weight"][::8].cpu().numpy(), g["grad_stem0_slice"], rtol=5e-2,
                               atol=5e-2 * np.abs(g["grad_stem0_slice"]).max())
    net.zero_grad(set_to_none=True)
    net.load_state_dict(E.seeded_state_dict(50, 3, seed=wseed), strict=True)  # undo BN running-stat updates
    # the real thing: two updater iterations
    upd.update()
    np.testing.assert_allclose(float(get_current_reporter().scalars()["loss/softmax"]), g["loss_mean_0"], rtol=1e-4)
    upd.update()
    # second iteration: even at lr 2e-5 the first update moves early layers by percents (gradients ~1000x the
    # weights at this random init), so its loss is only held to 5 %; the two-step parameter deltas of the
    # well-conditioned late layers pin momentum / weight decay / group learning rates instead
    np.testing.assert_allclose(float(get_current_reporter().scalars()["loss/softmax"]), g["loss_mean_1"], rtol=5e-2)
    assert upd.iteration == 2
    sd = net.state_dict()
    init = E.seeded_state_dict(50, 3, seed=wseed)
    for k in ("fc2.weight", "fc2.bias", "fc1.0.bn.weight"):
        ref = g["delta_" + k]
        np.testing.assert_allclose((sd[k].cpu() - init[k]).numpy(), ref, rtol=1e-1, atol=8e-2 * np.abs(ref).max(), err_msg=k)
    for name, ref in zip(g["after_names"], g["after_abs_sums"]):
        np.testing.assert_allclose(sd[str(name)].double().abs().sum().item(), ref, rtol=2e-3, err_msg=str(name))
    np.testing.assert_allclose(sd["emau.mu"][0, ::32, ::8].cpu().numpy(), g["after_emau_mu_slice"], rtol=5e-3, atol=1e-5)


def test_ema_net_label_maps_bit_exact(device, golden_dir):
    g = np.load(os.path.join(golden_dir, "ema_net_step.npz"))
    n_layers, classes, wseed, bseed, batch, size = g["cfg"].tolist()
    net = _net(device, wseed)
    images = E.seeded_batch(batch, size, classes, seed=10)["images"].to(device)
    with torch.no_grad():
        pred = net(images)
        labels = net.predict_classes(images)
    np.testing.assert_allclose(pred[:, :, ::16, ::16].cpu().numpy(), g["pred_slice"], rtol=2e-3, atol=2e-2)
    assert labels.dtype == torch.int64 and tuple(labels.shape) == (batch, 1, size, size)
    # logits agree to ~1e-2 absolute (measured 7e-3 on values of ~10 after 50 library-conv layers): the label
    # must be identical wherever the reference's top-2 margin is larger than twice that
    decided = g["pred_margin"].astype(np.float32) > 5e-2
    got = labels.cpu().numpy().astype(np.uint8)[:, 0]
    assert (got[decided] == g["pred_labels"][:, 0][decided]).all()
    assert decided.mean() > 0.95


def test_ema_net_step_vs_oracle_fresh_seed(device):
    """Same step on a seed the golden file has never seen, against the oracle run live on the host."""
    sd = E.seeded_state_dict(50, 3, seed=31)
    batch = E.seeded_batch(2, 128, 3, seed=32)
    total, loss_o, mu_o, grads_o = E.train_step(dict(sd), {}, batch)
    net = _net(device, 31)
    loss, mu = net(batch["images"].to(device), batch["segmented"].squeeze(1).to(device))
    np.testing.assert_allclose(loss.detach().cpu().numpy(), loss_o.numpy(), rtol=1e-4)
    loss.mean().backward()
    for name, p in net.named_parameters():
        if grads_o[name] is None:
            assert p.grad is None
        else:
            ref = grads_o[name].double().norm().item()
            assert abs(p.grad.double().norm().item() - ref) <= 2e-2 * ref + 1e-7, name


def _rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)) / np.linalg.norm(np.asarray(b, dtype=np.float64)))


def test_ema_net_conditioned_fixture_tight(device, golden_dir):
    """the same two iterations on the well-conditioned fixture (small residual branches, the SHIPPED
    learning rate 0.009), where tolerances mean something: loss 1e-5, every gradient norm 3e-3, gradient tensors 1e-4 (head)
    to 1e-2 (stem) in relative L2 (a ReLU network's gradient moves by ~sqrt(fraction of units whose pre-activation crossed
    zero): ~1e-3 is the fp32 floor for the deep layers), second-iteration loss 1e-4, two-step parameter deltas 1e-2."""
    from networks.ema_net.network import EMANet
    from networks.ema_net.utils import get_params
    from training.fused_sgd import FusedSGD
    from training.loop import get_current_reporter
    from updater.segmentation_updater import EMANetUpdater
    g = np.load(os.path.join(golden_dir, "ema_net_step_conditioned.npz"))
    n_layers, classes, wseed, bseed, batch, size = g["cfg"].tolist()

    def fresh():
        net = EMANet(3, 50, use_pretrained_resnet=False)
        net.load_state_dict(E.seeded_state_dict(50, 3, seed=wseed, residual_scale=0.1), strict=True)
        net.fc1[1].p = 0.0
        return net.to(device).train()

    net = fresh()
    batches = [E.seeded_batch(batch, size, classes, seed=bseed + i) for i in range(2)]
    b0 = {k: v.to(device) for k, v in batches[0].items()}
    loss, mu = net(b0["images"], b0["segmented"].squeeze(1))
    np.testing.assert_allclose(loss.detach().cpu().numpy(), g["loss"], rtol=1e-5)
    np.testing.assert_allclose(mu[:, ::32, ::8].cpu().numpy(), g["mu_slice"], rtol=1e-3, atol=1e-6)
    loss.mean().backward()
    grads = {n: p.grad for n, p in net.named_parameters()}
    norm_err = {str(name): abs(grads[str(name)].double().norm().item() - ref) / ref
                for name, ref in zip(g["grad_names"], g["grad_norms"]) if ref >= 0}
    measured = {"worst_grad_norm": max(norm_err.items(), key=lambda kv: kv[1]),
                "grad_fc2": _rel_l2(grads["fc2.weight"].cpu().numpy(), g["grad_fc2_weight"]),
                "grad_layer4_conv3": _rel_l2(grads["extractor.7.2.conv3.weight"][::64, ::16].cpu().numpy(), g["grad_layer4_conv3_slice"]),
                "grad_stem0": _rel_l2(grads["extractor.0.0.weight"][::8].cpu().numpy(), g["grad_stem0_slice"])}
    import json
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "ema_net_conditioned_parity.json"), "w") as f:
        json.dump(measured, f, indent=1)  # measured deviations, kept next to the stated tolerances
    assert measured["worst_grad_norm"][1] < 3e-3, measured  # (measured 1.0e-3 on a batch-norm scale of layer1)
    # (run-to-run spread on the GPU: layer4 2.1e-3 .. 2.9e-3, stem 3.8e-3 .. 5.1e-3 -- library kernels with atomics)
    assert measured["grad_fc2"] < 1e-4 and measured["grad_layer4_conv3"] < 6e-3 and measured["grad_stem0"] < 1e-2, measured
    # the updater, at the shipped learning rate
    net = fresh()
    lr = 0.009
    opt = FusedSGD([{"params": list(get_params(net, "1x")), "lr": lr, "weight_decay": 1e-4},
                    {"params": list(get_params(net, "1y")), "lr": lr, "weight_decay": 0},
                    {"params": list(get_params(net, "2x")), "lr": 2 * lr, "weight_decay": 0.0}], momentum=0.9)
    upd = EMANetUpdater(em_mom=0.9, iterators={"images": batches}, networks={"segmentation": net}, optimizers={"main": opt},
                        device=device)
    upd.update()
    np.testing.assert_allclose(float(get_current_reporter().scalars()["loss/softmax"]), g["loss_mean_0"], rtol=1e-5)
    upd.update()
    # a full-size step (lr 0.009) separates the iterations' states by the gradient differences above: loss 1e-3 (measured 2.8e-4)
    np.testing.assert_allclose(float(get_current_reporter().scalars()["loss/softmax"]), g["loss_mean_1"], rtol=1e-3)
    sd = net.state_dict()
    init = E.seeded_state_dict(50, 3, seed=wseed, residual_scale=0.1)
    after = {k: _rel_l2((sd[k].cpu() - init[k]).numpy(), g["delta_" + k]) for k in ("fc2.weight", "fc2.bias", "fc1.0.bn.weight")}
    after["layer4_conv3"] = _rel_l2((sd["extractor.7.2.conv3.weight"].cpu() - init["extractor.7.2.conv3.weight"])[::64, ::16].numpy(),
                                    g["delta_layer4_conv3_slice"])
    after["stem0"] = _rel_l2((sd["extractor.0.0.weight"].cpu() - init["extractor.0.0.weight"])[::8].numpy(), g["delta_stem0_slice"])
    after["abs_sums"] = max(abs(sd[str(n)].double().abs().sum().item() - r) / (abs(r) + 1e-12) for n, r in zip(g["after_names"], g["after_abs_sums"]))
    measured["two_step_deltas_rel_l2"] = after
    with open(os.path.join("gpurun_out", "ema_net_conditioned_parity.json"), "w") as f:
        json.dump(measured, f, indent=1)
    assert after["fc2.weight"] < 1e-2 and after["fc2.bias"] < 1e-2 and after["fc1.0.bn.weight"] < 2e-2, after  # momentum, wd, group lrs
    # the second step's gradients are taken at parameters that already differ by the first step's 3e-3: measured 3.0e-2 / 4.8e-2
    assert after["layer4_conv3"] < 6e-2 and after["stem0"] < 1e-1 and after["abs_sums"] < 1e-3, after


def test_ema_net_step_at_baseline_batch_vs_oracle(device):
    """BASELINE.json configs[3] at ITS batch: EMANet-50, 256 x 256, B = 16.  The batch decides the
    dispatch -- at B = 16 a channel of a 32 x 32 layer is exactly one 16 384-element slice, so the single-launch
    ``bn_fused_fwd/bwd_kernel`` replace the three-launch batch norm, the 1x1 / Winograd tile plans and split-K slabs change
    -- and before this test that step was only ever run by bench.py, which checks nothing.  One forward + backward on the
    conditioned seeded state (bn3 x 0.1, as ema_net_step_conditioned.npz) against the oracle run live on the host: per-sample
    losses, the EM bases, every gradient norm, and a record that the B = 16-only kernels were the ones dispatched."""
    import json
    import sis_hip
    from networks.ema_net.network import EMANet
    sd = E.seeded_state_dict(50, 3, seed=41, residual_scale=0.1)
    batch = E.seeded_batch(16, 256, 3, seed=42)
    total, loss_o, mu_o, grads_o = E.train_step(dict(sd), {}, batch)
    net = EMANet(3, 50, use_pretrained_resnet=False)
    net.load_state_dict(sd, strict=True)
    net.fc1[1].p = 0.0
    net = net.to(device).train()
    records = []
    sis_hip.library_calls(reset=True)
    sis_hip.set_profiler(records)
    try:
        loss, mu = net(batch["images"].to(device), batch["segmented"].squeeze(1).to(device))
        loss.mean().backward()
        torch.cuda.synchronize()
    finally:
        sis_hip.set_profiler(None)
    launched = {}
    for name, *_ in records:
        launched[name] = launched.get(name, 0) + 1
    calls = sis_hip.library_calls(reset=True)
    # the single-pass batch norm takes every 32 x 32 layer (42 of the 55 norms) in both directions
    # (backward: the 512-thread instance of bn_wide_bwd_kernel -- the name the library reports is the kernel rocprof lists)
    assert launched.get("bn_fused_fwd_kernel", 0) >= 40 and launched.get("bn_wide_bwd_kernel", 0) >= 40, launched
    assert launched.get("bn_wide_fwd_kernel", 0) >= 8, launched   # layer1's 64 x 64 norms: one 1 024-thread workgroup per channel
    assert calls["fallback"] == {}, calls       # nothing the kernels declined; the 3-channel stem is the one intended library layer
    assert set(calls["intended"]) <= {"hip_conv.HipConv2d.forward"}, calls
    np.testing.assert_allclose(loss.detach().cpu().numpy(), loss_o.numpy(), rtol=1e-5)
    mu_err = _rel_l2(mu.cpu().numpy(), mu_o.numpy())
    norm_err = {}
    for name, p in net.named_parameters():
        if grads_o[name] is None:
            assert p.grad is None, name
        else:
            ref = grads_o[name].double().norm().item()
            norm_err[name] = abs(p.grad.double().norm().item() - ref) / (ref + 1e-30)
    worst = max(norm_err.items(), key=lambda kv: kv[1])
    measured = {"loss_rel": float(np.abs(loss.detach().cpu().numpy() / loss_o.numpy() - 1).max()), "mu_rel_l2": mu_err,
                "worst_grad_norm": worst, "grad_fc2_rel_l2": _rel_l2(net.fc2.weight.grad.cpu().numpy(), grads_o["fc2.weight"].numpy()),
                "launched": launched}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "ema_net_b16_parity.json"), "w") as f:
        json.dump(measured, f, indent=1)
    # the bounds of the conditioned B = 2 fixture (test_ema_net_conditioned_fixture_tight): the floor of a ReLU network in fp32
    assert mu_err < 1e-4 and worst[1] < 3e-3 and measured["grad_fc2_rel_l2"] < 1e-4, measured


def test_ema_net_101_shipped_config_vs_oracle(device):
    """The configuration the reference actually SHIPS (configs/segmenter/stylegan2_ema_net_segmenter.yaml:16-24: EMANet on a
    ResNet-101 trunk, batch 8, 256 x 256; VERDICT r4 missing #4): one forward + backward on the conditioned seeded state against
    the oracle run live on the host.  The 23 units of layer3 run on the same kernels and tile plans as EMANet-50's six, but
    nothing had ever executed them: this pins losses, EM bases, every gradient norm and that no operator of the step falls
    back to the ROCm libraries."""
    import json
    import sis_hip
    from networks.ema_net.network import EMANet
    sd = E.seeded_state_dict(101, 3, seed=51, residual_scale=0.1)
    batch = E.seeded_batch(8, 256, 3, seed=52)
    total, loss_o, mu_o, grads_o = E.train_step(dict(sd), {}, batch, n_layers=101)
    net = EMANet(3, 101, use_pretrained_resnet=False)
    net.load_state_dict(sd, strict=True)
    net.fc1[1].p = 0.0
    net = net.to(device).train()
    sis_hip.library_calls(reset=True)
    loss, mu = net(batch["images"].to(device), batch["segmented"].squeeze(1).to(device))
    loss.mean().backward()
    torch.cuda.synchronize()
    calls = sis_hip.library_calls(reset=True)
    assert calls["fallback"] == {}, calls
    assert set(calls["intended"]) <= {"hip_conv.HipConv2d.forward"}, calls
    np.testing.assert_allclose(loss.detach().cpu().numpy(), loss_o.numpy(), rtol=2e-5)
    mu_err = _rel_l2(mu.cpu().numpy(), mu_o.numpy())
    norm_err = {}
    for name, p in net.named_parameters():
        if grads_o[name] is None:
            assert p.grad is None, name
        else:
            ref = grads_o[name].double().norm().item()
            norm_err[name] = abs(p.grad.double().norm().item() - ref) / (ref + 1e-30)
    worst = max(norm_err.items(), key=lambda kv: kv[1])
    measured = {"loss_rel": float(np.abs(loss.detach().cpu().numpy() / loss_o.numpy() - 1).max()), "mu_rel_l2": mu_err,
                "worst_grad_norm": worst, "grad_fc2_rel_l2": _rel_l2(net.fc2.weight.grad.cpu().numpy(), grads_o["fc2.weight"].numpy()),
                "parameters": len(norm_err)}
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", "ema_net_101_b8_parity.json"), "w") as f:
        json.dump(measured, f, indent=1)
    # EMANet-50's bounds at B = 16 (test_ema_net_step_at_baseline_batch_vs_oracle), with twice the depth between loss and stem
    assert mu_err < 1e-4 and worst[1] < 6e-3 and measured["grad_fc2_rel_l2"] < 1e-4, measured
