"""CPU: the SWAGAN oracle (oracle/swagan_ref.py) against the golden outputs of the reference's own model, and the
product module's state_dict surface."""
import os

import numpy as np
import pytest
import torch

from oracle import swagan_ref as W


def _inputs(g):
    size, style_dim, n_mlp, cm, wseed, zseed = g["cfg"].tolist()
    sd = W.seeded_state_dict(size, style_dim, n_mlp, cm, seed=wseed)
    rng = np.random.RandomState(zseed)
    z = torch.from_numpy(rng.randn(2, style_dim).astype(np.float32))
    z2 = torch.from_numpy(rng.randn(2, style_dim).astype(np.float32))
    mean_latent = torch.from_numpy(rng.randn(1, style_dim).astype(np.float32)) * 0.1
    return (size, style_dim, n_mlp, cm), sd, z, z2, mean_latent


@pytest.mark.parametrize("name", ["swagan32.npz", "swagan64.npz"])
def test_oracle_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name))
    _, sd, z, z2, ml = _inputs(g)
    img, acts = W.generator_forward(sd, [z], return_intermediate_activations=True)
    np.testing.assert_allclose(img.numpy(), g["image"], rtol=1e-5, atol=1e-5 * np.abs(g["image"]).max())
    for k, v in acts.items():
        np.testing.assert_allclose(v[:, ::7, ::3, ::3].numpy(), g[f"act_{k}_slice"], rtol=1e-5,
                                   atol=1e-5 * np.abs(g[f"act_{k}_slice"]).max())
        np.testing.assert_allclose(v.double().sum().item(), g[f"act_{k}_sum"], rtol=1e-6, atol=1e-3)
    mixed, _ = W.generator_forward(sd, [z, z2], inject_index=3, truncation=0.7, truncation_latent=ml)
    np.testing.assert_allclose(mixed.numpy(), g["mixed"], rtol=1e-5, atol=1e-5 * np.abs(g["mixed"]).max())


def test_product_state_dict_is_the_reference_schema(golden_dir):
    from networks.swagan.model import Generator
    g = np.load(os.path.join(golden_dir, "swagan32.npz"))
    (size, style_dim, n_mlp, cm), sd, *_ = _inputs(g)
    net = Generator(size, style_dim, n_mlp, channel_multiplier=cm)
    assert sorted(net.state_dict().keys()) == sorted(str(k) for k in g["state_keys"])
    net.load_state_dict(sd, strict=True)
    assert (net.log_size, net.n_latent, net.num_layers) == (4, 6, 5)
    assert [tuple(n.shape[-2:]) for n in net.make_noise()] == [(4, 4), (8, 8), (8, 8), (16, 16), (16, 16)]


def test_product_discriminator_has_the_reference_schema(golden_dir):
    """networks.swagan.Discriminator: the same state_dict keys / shapes, in the same order, as the reference module
    whose golden run is stored in swagan_d32.npz (tests/golden/make_golden_swagan_d.py)."""
    from networks.swagan import Discriminator
    g = np.load(os.path.join(golden_dir, "swagan_d32.npz"))
    size, cm, _ = g["cfg"].tolist()
    net = Discriminator(size, channel_multiplier=cm)
    assert list(net.state_dict().keys()) == g["schema_names"].tolist()
    assert [",".join(map(str, v.shape)) for v in net.state_dict().values()] == g["schema_shapes"].tolist()
    ref_like = {k: v for k, v in net.state_dict().items() if k.endswith(("kernel", ".ll", ".lh", ".hl", ".hh"))}
    assert all(torch.isfinite(v).all() for v in ref_like.values()) and len(ref_like) > 0
