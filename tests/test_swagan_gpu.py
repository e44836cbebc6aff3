"""GPU parity: the SWAGAN generator on the MI355X kernels against the reference golden and the CPU oracle.
Tolerance: 2e-4 of the image range after the whole stack (same bound as the StyleGAN2 generator test)."""
import os

import numpy as np
import pytest
import torch

from oracle import swagan_ref as W
from test_swagan_oracle_cpu import _inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["swagan32.npz", "swagan64.npz"])
def test_generator_matches_reference_golden(device, golden_dir, name):
    from networks.swagan.model import Generator
    g = np.load(os.path.join(golden_dir, name))
    (size, style_dim, n_mlp, cm), sd, z, z2, ml = _inputs(g)
    net = Generator(size, style_dim, n_mlp, channel_multiplier=cm)
    net.load_state_dict(sd, strict=True)
    net = net.to(device).eval()
    with torch.no_grad():
        img, acts = net([z.to(device)], randomize_noise=False, return_intermediate_activations=True)
        mixed, _ = net([z.to(device), z2.to(device)], inject_index=3, truncation=0.7, truncation_latent=ml.to(device),
                       randomize_noise=False)
    assert tuple(img.shape) == (2, 3, size, size)
    np.testing.assert_allclose(img.cpu().numpy(), g["image"], rtol=0, atol=2e-4 * np.abs(g["image"]).max())
    np.testing.assert_allclose(mixed.cpu().numpy(), g["mixed"], rtol=0, atol=2e-4 * np.abs(g["mixed"]).max())
    for k, v in acts.items():
        ref = g[f"act_{k}_slice"]
        np.testing.assert_allclose(v[:, ::7, ::3, ::3].cpu().numpy(), ref, rtol=0, atol=1e-4 * np.abs(ref).max())


def test_haar_transforms_invert_each_other(device):
    from networks.swagan.model import HaarTransform, InverseHaarTransform
    x = torch.randn(2, 3, 32, 48, device=device)
    bands = HaarTransform(3).to(device)(x)
    assert tuple(bands.shape) == (2, 12, 16, 24)
    back = InverseHaarTransform(3).to(device)(bands)
    np.testing.assert_allclose(back.cpu().numpy(), x.cpu().numpy(), atol=1e-6)
    filters = W.haar_filters()
    np.testing.assert_allclose(bands.cpu().numpy(), W.dwt(x.cpu(), filters).numpy(), atol=1e-6)


def test_discriminator_matches_reference_golden(device, golden_dir):
    """SWAGAN discriminator (wavelet pyramid FromRGB + ConvBlocks) forward and the logistic-loss gradients against the
    unmodified reference module; 3x3 stride-1 layers run on the Winograd kernels."""
    import sys
    import torch.nn.functional as F
    sys.path.insert(0, golden_dir)
    from make_golden_swagan_d import seed_discriminator, seeded_images
    from networks.swagan import Discriminator
    g = np.load(os.path.join(golden_dir, "swagan_d32.npz"))
    size, cm, b = g["cfg"].tolist()
    net = Discriminator(size, channel_multiplier=cm)
    seed_discriminator(net, 31)
    net = net.to(device).train()
    real, fake = (t.to(device) for t in seeded_images(size, b, 32))
    real_pred, fake_pred = net(real), net(fake)
    loss = F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()
    loss.backward()
    np.testing.assert_allclose(real_pred.detach().cpu().numpy(), g["real_pred"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(fake_pred.detach().cpu().numpy(), g["fake_pred"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(loss.item(), g["d_loss"], rtol=2e-4)
    for name, p in net.named_parameters():
        want_norm, want_head = float(g[f"norm/{name}"]), g[f"head/{name}"]
        np.testing.assert_allclose(p.grad.double().norm().item(), want_norm, rtol=2e-3, atol=1e-5, err_msg=name)
        np.testing.assert_allclose(p.grad.flatten()[:16].cpu().numpy(), want_head, rtol=0,
                                   atol=2e-3 * max(np.abs(want_head).max(), want_norm / np.sqrt(p.numel())) + 1e-5, err_msg=name)
