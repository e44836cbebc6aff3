"""Golden vectors of the SWAGAN generator from the UNMODIFIED reference networks/swagan/model.py, imported here with
the oracle's CPU ops as its ``.op`` (oracle/load_reference.py::load_reference_swagan).

    python tests/golden/make_golden_swagan.py      -> tests/golden/swagan32.npz, swagan64.npz
Weights / inputs are re-derived in the tests from oracle.swagan_ref.seeded_state_dict and numpy RandomState streams.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import swagan_ref as W  # noqa: E402
from oracle.load_reference import load_reference_swagan  # noqa: E402

ref = load_reference_swagan()
for size, style_dim, n_mlp, cm, wseed, zseed in [(32, 64, 2, 1, 5, 6), (64, 512, 8, 2, 7, 8)]:
    sd = W.seeded_state_dict(size, style_dim, n_mlp, cm, seed=wseed)
    g = ref.Generator(size, style_dim, n_mlp, channel_multiplier=cm)
    g.load_state_dict(sd, strict=True)
    g.eval()
    rng = np.random.RandomState(zseed)
    z = torch.from_numpy(rng.randn(2, style_dim).astype(np.float32))
    z2 = torch.from_numpy(rng.randn(2, style_dim).astype(np.float32))
    mean_latent = torch.from_numpy(rng.randn(1, style_dim).astype(np.float32)) * 0.1
    with torch.no_grad():
        img, acts = g([z], randomize_noise=False, return_intermediate_activations=True)
        mixed, _ = g([z, z2], inject_index=3, truncation=0.7, truncation_latent=mean_latent, randomize_noise=False)
    out = {"cfg": np.asarray([size, style_dim, n_mlp, cm, wseed, zseed]), "image": img.numpy(), "mixed": mixed.numpy(),
           "state_keys": np.asarray(list(g.state_dict().keys()))}
    for k, v in acts.items():
        out[f"act_{k}_sum"] = np.asarray(v.double().sum().item())
        out[f"act_{k}_slice"] = v[:, ::7, ::3, ::3].numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"swagan{size}.npz"), **out)
    print(size, img.shape, float(img.abs().max()))
