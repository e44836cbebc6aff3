"""CPU: surface of the GAN-training row -- the product Discriminator's state_dict is the
reference schema (oracle/stylegan2_ref.py::discriminator_schema, asserted equal to the reference's own module when
tests/golden/make_golden_gan.py ran), the golden file is self-consistent, and the updater's host logic."""
import os

import numpy as np
import pytest
import torch
from torch import nn

from oracle import stylegan2_ref as R
from oracle.load_reference import reference_available


def gan_inputs(g):
    size, style_dim, n_mlp, cm, b = g["cfg"].tolist()
    rng = np.random.RandomState(13)
    z1 = torch.from_numpy(rng.standard_normal((b, style_dim)).astype(np.float32))
    z2 = torch.from_numpy(rng.standard_normal((b, style_dim)).astype(np.float32))
    real = torch.from_numpy(rng.uniform(-1, 1, (b, 3, size, size)).astype(np.float32))
    path_noise = torch.from_numpy(rng.standard_normal((b // 2, 3, size, size)).astype(np.float32))
    _, noise = R.seeded_inputs(size, 1, style_dim, seed=14)
    return dict(size=size, style_dim=style_dim, n_mlp=n_mlp, cm=cm, batch=b, z1=z1, z2=z2, real=real,
                path_noise=path_noise, noise=noise,
                g_state=R.seeded_state_dict(size, style_dim, n_mlp, cm, seed=11),
                d_state=R.seeded_discriminator_state_dict(size, cm, seed=12))


@pytest.mark.parametrize("size,cm", [(32, 1), (64, 2), (256, 2)])
def test_product_discriminator_has_the_reference_schema(size, cm):
    from networks.stylegan2.model import Discriminator
    net = Discriminator(size, channel_multiplier=cm)
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == R.discriminator_schema(size, cm)
    net.load_state_dict(R.seeded_discriminator_state_dict(size, cm, seed=3), strict=True)
    assert net.stddev_group == 4 and net.stddev_feat == 1


@pytest.mark.skipif(not reference_available(), reason="reference tree not mounted")
def test_schema_is_the_reference_modules_own():
    from oracle.load_reference import load_reference_stylegan2
    ref = load_reference_stylegan2()
    for size, cm in [(32, 1), (128, 2)]:
        d = ref.Discriminator(size, channel_multiplier=cm)
        assert [(k, tuple(v.shape)) for k, v in d.state_dict().items()] == R.discriminator_schema(size, cm)


def test_golden_is_self_consistent(golden_dir):
    g = np.load(os.path.join(golden_dir, "gan32.npz"))
    real_pred, fake_pred = g["real_pred"].astype(np.float64), g["fake_pred"].astype(np.float64)
    softplus = lambda t: np.log1p(np.exp(t))  # noqa: E731
    np.testing.assert_allclose(softplus(-real_pred).mean() + softplus(fake_pred).mean(), g["d_loss"], rtol=1e-5)
    r1 = (g["r1_grad_real"].astype(np.float64) ** 2).reshape(real_pred.shape[0], -1).sum(1).mean()
    np.testing.assert_allclose(r1, g["r1_loss"], rtol=1e-5)
    lengths = np.sqrt((g["path_grad"].astype(np.float64) ** 2).sum(2).mean(1))
    np.testing.assert_allclose(lengths, g["path_lengths"], rtol=1e-5)
    np.testing.assert_allclose(0.01 * lengths.mean(), g["path_mean"], rtol=1e-5)
    np.testing.assert_allclose(((lengths - g["path_mean"]) ** 2).mean(), g["path_penalty"], rtol=1e-5)
    names = {k.split("/", 2)[2] for k in g.files if k.startswith("d_step/norm/")}
    assert names == {n for n, _ in R.discriminator_schema(32, 1) if not n.endswith(".kernel")}


def test_updater_host_logic():
    from updater.stylegan_2_updater import Stylegan2Updater

    class Tiny(nn.Module):
        def __init__(self, v):
            super().__init__()
            self.w = nn.Parameter(torch.full((3,), float(v)))
            self.noises = nn.Module()
            for i in range(3):
                self.noises.register_buffer(f"noise_{i}", torch.full((1, 1, 2, 2), float(i)))

    g, g_ema = Tiny(1.0), Tiny(0.0)
    with pytest.raises(AssertionError):
        Stylegan2Updater(iterators={}, networks={}, optimizers={}, device="cpu")
    up = Stylegan2Updater(iterators={}, networks={"generator": g}, optimizers={}, device="cpu", g_ema=g_ema, latent_size=8,
                          regularization_options={"d_reg_interval": 8, "r1_weight": 5}, freeze_stochastic_noise_layers=[1])
    assert (up.d_reg_interval, up.g_reg_interval, up.r1_weight, up.path_reg_weight) == (8, 4, 5.0, 2.0)
    assert abs(up.accumulation_decay - 0.5 ** (32 / 10000)) < 1e-15
    up.accumulate(g, 0.75)
    np.testing.assert_allclose(g_ema.w.detach().numpy(), 0.25)
    noise = up.make_stochastic_noise()
    assert noise[0] is None and noise[2] is None and float(noise[1].mean()) == 1.0
    assert Stylegan2Updater(iterators={}, networks={}, optimizers={}, device="cpu", g_ema=g_ema,
                            freeze_stochastic_noise_layers=True).stochastic_noise_layers_to_freeze == [0, 1, 2]
    assert [tuple(t.shape) for t in up.make_noise(5, 2)] == [(5, 8), (5, 8)]
    up.style_mixing_prob = 0
    assert len(up.mixing_styles(3)) == 1
    # loss arithmetic on plain tensors
    rp, fp = torch.tensor([[0.3], [-1.0]]), torch.tensor([[0.5], [2.0]])
    want = np.log1p(np.exp(-rp.numpy())).mean() + np.log1p(np.exp(fp.numpy())).mean()
    np.testing.assert_allclose(up.d_logistic_loss(rp, fp).item(), want, rtol=1e-6)
    np.testing.assert_allclose(up.g_nonsaturating_loss(fp).item(), np.log1p(np.exp(-fp.numpy())).mean(), rtol=1e-6)
    x = torch.randn(2, 3, 4, 4, requires_grad=True)
    np.testing.assert_allclose(up.d_r1_loss((x ** 2).sum((1, 2, 3)) / 2, x).item(),
                               (x.detach() ** 2).reshape(2, -1).sum(1).mean().item(), rtol=1e-6)
    lat = torch.randn(2, 4, 8, requires_grad=True)
    img = (lat.sum((1, 2))[:, None, None, None] * torch.ones(2, 3, 4, 4))
    pen, mean, lengths = up.g_path_regularize(img, lat, 0, noise=torch.ones(2, 3, 4, 4))
    np.testing.assert_allclose(lengths.detach().numpy(), np.sqrt(8 * (48 / 4) ** 2), rtol=1e-6)  # every d img / d latent = 1
    np.testing.assert_allclose(mean.item(), 0.01 * lengths.mean().item(), rtol=1e-6)


def test_update_disabler_restores_flags():
    from training.loop import UpdateDisabler
    net = nn.Linear(2, 2)
    net.bias.requires_grad = False
    with UpdateDisabler(net):
        assert not any(p.requires_grad for p in net.parameters())
    assert net.weight.requires_grad and not net.bias.requires_grad
