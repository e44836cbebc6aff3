"""GPU parity of the GAN-training row: the four sub-steps of ``Stylegan2Updater`` on the MI355X
kernels against golden values of the unmodified reference modules (tests/golden/make_golden_gan.py), the second-order
Winograd convolution against the library's double backward, and the shared-weight modulated convolution against
the reference's grouped formulation.

Tolerances (fp32 end to end; Winograd F(2x2,3x3) and a different summation order): losses / predictions 2e-4
relative, image gradient of R1 2e-3 in L2, gradient norms 2e-3 relative, gradient entries 2e-3 of the tensor's largest golden entry."""
import os

import numpy as np
import pytest
import torch
from torch import autograd
from torch.nn import functional as F

from test_gan_oracle_cpu import gan_inputs

pytestmark = pytest.mark.gpu


def _networks(cfg, device):
    from networks.stylegan2.model import Discriminator, Generator
    g = Generator(cfg["size"], cfg["style_dim"], cfg["n_mlp"], channel_multiplier=cfg["cm"])
    g.load_state_dict(cfg["g_state"], strict=True)
    d = Discriminator(cfg["size"], channel_multiplier=cfg["cm"])
    d.load_state_dict(cfg["d_state"], strict=True)
    return g.to(device).train(), d.to(device).train()


def _updater(g, d, device):
    import copy
    from updater.stylegan_2_updater import Stylegan2Updater
    return Stylegan2Updater(iterators={}, networks={"generator": g, "discriminator": d}, optimizers={}, device=device,
                            g_ema=copy.deepcopy(g), latent_size=g.style_dim)


def _check_grads(net, golden, tag):
    for name, p in net.named_parameters():
        want_norm, want_head = float(golden[f"{tag}/norm/{name}"]), golden[f"{tag}/head/{name}"]
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        np.testing.assert_allclose(got.double().norm().item(), want_norm, rtol=2e-3, atol=1e-5, err_msg=f"{tag} {name}")  # atol: cancelling sums (noise weights)
        np.testing.assert_allclose(got.flatten()[:16].cpu().numpy(), want_head, rtol=0,
                                   atol=2e-3 * max(np.abs(want_head).max(), want_norm / np.sqrt(p.numel())) + 1e-5,
                                   err_msg=f"{tag} {name}")
    net.zero_grad(set_to_none=True)


def test_four_substeps_match_reference_golden(device, golden_dir):
    from training.loop import UpdateDisabler
    gold = np.load(os.path.join(golden_dir, "gan32.npz"))
    cfg = gan_inputs(gold)
    g, d = _networks(cfg, device)
    up = _updater(g, d, device)
    z1, z2, real = cfg["z1"].to(device), cfg["z2"].to(device), cfg["real"].to(device)
    noise = [n.to(device) for n in cfg["noise"]]
    b = cfg["batch"]

    with UpdateDisabler(g):  # update_discriminator (:126-147): fakes come from the fused inference path
        fake, _ = g([z1, z2], inject_index=3, noise=noise)
        assert not fake.requires_grad
        fake_pred, real_pred = d(fake), d(real)
        d_loss = up.d_logistic_loss(real_pred, fake_pred)
        d_loss.backward()
    np.testing.assert_allclose(fake.cpu().numpy(), gold["fake"], rtol=0, atol=2e-4 * np.abs(gold["fake"]).max())
    np.testing.assert_allclose(fake_pred.detach().cpu().numpy(), gold["fake_pred"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(real_pred.detach().cpu().numpy(), gold["real_pred"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(d_loss.item(), gold["d_loss"], rtol=2e-4)
    _check_grads(d, gold, "d_step")

    img = real.clone().requires_grad_(True)  # regularize_discriminator (:150-161)
    real_pred = d(img)
    grad_real, = autograd.grad(outputs=real_pred.sum(), inputs=img, create_graph=True)
    r1 = up.d_r1_loss(real_pred, img)
    (up.r1_weight / 2 * r1 * up.d_reg_interval + 0 * real_pred[0]).backward()
    # entry-wise the image gradient is only piecewise smooth (a leaky-ReLU gate whose pre-activation is ~0 flips with
    # the summation order and moves its receptive field by a step), so it is compared in the L2 sense
    want = gold["r1_grad_real"].astype(np.float64)
    err = np.linalg.norm(grad_real.detach().cpu().numpy().astype(np.float64) - want) / np.linalg.norm(want)
    assert err < 2e-3, err
    np.testing.assert_allclose(r1.item(), gold["r1_loss"], rtol=5e-4)
    _check_grads(d, gold, "d_reg")

    with UpdateDisabler(d):  # update_generator (:163-176)
        fake, _ = g([z1, z2], inject_index=3, noise=noise)
        assert fake.requires_grad
        g_loss = up.g_nonsaturating_loss(d(fake))
        g_loss.backward()
    np.testing.assert_allclose(g_loss.item(), gold["g_loss"], rtol=2e-4)
    _check_grads(g, gold, "g_step")

    fake, latents = g([z1[:b // 2], z2[:b // 2]], return_latents=True, inject_index=3, noise=noise)  # regularize_generator
    penalty, mean, lengths = up.g_path_regularize(fake, latents, 0, noise=cfg["path_noise"].to(device))
    weighted = up.path_reg_weight * up.g_reg_interval * penalty
    weighted = weighted + 0 * fake[0, 0, 0, 0]
    weighted.backward()
    np.testing.assert_allclose(lengths.detach().cpu().numpy(), gold["path_lengths"], rtol=5e-4)
    np.testing.assert_allclose(mean.item(), gold["path_mean"], rtol=5e-4)
    np.testing.assert_allclose(penalty.item(), gold["path_penalty"], rtol=1e-3)
    _check_grads(g, gold, "g_reg")


@pytest.mark.parametrize("shape", [(4, 64, 64, 32, 32), (2, 128, 64, 16, 16), (8, 64, 128, 64, 64)])
def test_conv3x3_double_backward_matches_library(device, shape):
    """d/d(x, w, dy) of <dx, a> + <dw, b>: the three second-order products on the Winograd kernels."""
    import sis_hip
    from networks.hip_conv import conv3x3
    b, cin, cout, h, w = shape
    gen = torch.Generator(device="cpu").manual_seed(b + cin)
    x = torch.randn(b, cin, h, w, generator=gen).to(device).requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=gen) / (3 * cin ** 0.5)).to(device).requires_grad_(True)
    gy = torch.randn(b, cout, h, w, generator=gen).to(device).requires_grad_(True)
    a = torch.randn(b, cin, h, w, generator=gen).to(device)
    bb = torch.randn(cout, cin, 3, 3, generator=gen).to(device)
    assert sis_hip.conv3x3_supported(x, wt)

    def second_order(conv):
        y = conv(x, wt)
        dx, dw = autograd.grad(y, (x, wt), gy, create_graph=True)
        return (dx, dw) + autograd.grad((dx * a).sum() + (dw * bb).sum(), (x, wt, gy))

    got = second_order(conv3x3)
    want = second_order(lambda i, k: F.conv2d(i.double(), k.double(), padding=1).float())
    for name, u, v in zip(("dx", "dw", "d2/dx", "d2/dw", "d2/dgy"), got, want):
        np.testing.assert_allclose(u.detach().cpu().numpy(), v.detach().cpu().numpy(), rtol=0,
                                   atol=3e-4 * float(v.detach().abs().max()), err_msg=name)


@pytest.mark.parametrize("upsample,k", [(False, 3), (True, 3), (False, 1)])
def test_shared_weight_modconv_matches_grouped_formulation(device, upsample, k, monkeypatch):
    from networks.stylegan2.model import ModulatedConv2d
    torch.manual_seed(5)
    layer = ModulatedConv2d(64, 128, k, 32, demodulate=(k == 3), upsample=upsample).to(device)
    x = torch.randn(3, 64, 16, 16, device=device, requires_grad=True)
    style = torch.randn(3, 32, device=device, requires_grad=True)
    probe = None
    results = []
    for grouped in ("0", "1"):
        monkeypatch.setenv("SIS_MODCONV_GROUPED", grouped)
        y = layer(x, style)
        if probe is None:
            probe = torch.randn_like(y)
        grads = autograd.grad((y * probe).sum(), (x, style, layer.weight, layer.modulation.weight))
        results.append((y,) + grads)
    for u, v in zip(*results):
        np.testing.assert_allclose(u.detach().cpu().numpy(), v.detach().cpu().numpy(), rtol=0, atol=2e-4 * float(v.detach().abs().max()))


def test_update_core_runs_the_schedule(device):
    """Five iterations at 32^2: regularisers fire on their intervals, observations are finite, g_ema moves towards G."""
    from training.loop import get_current_reporter
    from networks.stylegan2.model import Discriminator, Generator
    from updater.stylegan_2_updater import Stylegan2Updater
    torch.manual_seed(0)
    g = Generator(32, 64, 2, channel_multiplier=1).to(device)
    g_ema = Generator(32, 64, 2, channel_multiplier=1).to(device)
    g_ema.load_state_dict(g.state_dict())
    d = Discriminator(32, channel_multiplier=1).to(device)
    opts = {"generator": torch.optim.Adam(g.parameters(), lr=2e-3 * 4 / 5, betas=(0.0, 0.99 ** (4 / 5))),
            "discriminator": torch.optim.Adam(d.parameters(), lr=2e-3 * 16 / 17, betas=(0.0, 0.99 ** (16 / 17)))}

    def batches():
        while True:
            yield {"image": torch.rand(8, 3, 32, 32) * 2 - 1}

    up = Stylegan2Updater(iterators={"images": batches()}, networks={"generator": g, "discriminator": d}, optimizers=opts,
                          device=device, g_ema=g_ema, latent_size=64, regularization_options={"d_reg_interval": 2,
                                                                                              "g_reg_interval": 4})
    before = [p.detach().clone() for p in g_ema.parameters()]
    seen = []
    for _ in range(5):
        get_current_reporter().observations.clear()
        up.update()
        obs = get_current_reporter().scalars()
        assert all(np.isfinite(v) for v in obs.values()), obs
        seen.append(set(obs))
    assert "discriminator/r1_loss" in seen[0] and "discriminator/r1_loss" not in seen[1] and "discriminator/r1_loss" in seen[2]
    assert "generator/perceputal_path_loss" in seen[0] and "generator/perceputal_path_loss" in seen[4]
    assert "generator/perceputal_path_loss" not in seen[1]
    assert up.iteration == 5 and up.mean_path_length_avg > 0
    assert any(not torch.equal(a, p) for a, p in zip(before, g_ema.parameters()))
    assert all(p.requires_grad for p in g.parameters()) and all(p.requires_grad for p in d.parameters())
