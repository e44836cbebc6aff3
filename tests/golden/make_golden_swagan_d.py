"""Golden vectors of the SWAGAN discriminator from the UNMODIFIED reference networks/swagan/model.py (imported with the
oracle's CPU ops as its ``.op``; oracle/load_reference.py::load_reference_swagan): state_dict schema, predictions, the
logistic D loss on a (real, fake) pair of seeded image batches and, per parameter, gradient norm + first 16 entries.

    python tests/golden/make_golden_swagan_d.py      -> tests/golden/swagan_d32.npz
Weights are re-derived in the tests by ``seed_discriminator`` below (numpy RandomState stream over the parameters in
state_dict order; buffers -- blur taps, Haar filters -- keep their constructor values).
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.load_reference import load_reference_swagan  # noqa: E402


def seed_discriminator(net, seed):
    """N(0,1) weights, N(0, 0.1^2) biases, drawn in ``named_parameters`` order from a frozen numpy stream."""
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            t = torch.from_numpy(rng.standard_normal(tuple(p.shape))).to(p.dtype)
            p.copy_(t * 0.1 if name.endswith("bias") else t)


def seeded_images(size, batch, seed):
    rng = np.random.RandomState(seed)
    return (torch.from_numpy(rng.uniform(-1, 1, (batch, 3, size, size)).astype(np.float32)),
            torch.from_numpy(rng.uniform(-1, 1, (batch, 3, size, size)).astype(np.float32)))


if __name__ == "__main__":
    SIZE, CM, B = 32, 1, 4
    ref = load_reference_swagan()
    d = ref.Discriminator(SIZE, channel_multiplier=CM)
    seed_discriminator(d, 31)
    d.train()
    real, fake = seeded_images(SIZE, B, 32)
    real_pred, fake_pred = d(real), d(fake)
    loss = F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()
    loss.backward()
    out = {"cfg": np.asarray([SIZE, CM, B]), "real_pred": real_pred.detach().numpy(), "fake_pred": fake_pred.detach().numpy(),
           "d_loss": np.asarray(loss.item()),
           "schema_names": np.asarray(list(d.state_dict().keys())),
           "schema_shapes": np.asarray([",".join(map(str, v.shape)) for v in d.state_dict().values()])}
    for name, p in d.named_parameters():
        out[f"norm/{name}"] = np.asarray(p.grad.double().norm().item())
        out[f"head/{name}"] = p.grad.flatten()[:16].numpy().copy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "swagan_d32.npz"), **out)
    print(loss.item(), real_pred.flatten().tolist(), len(out["schema_names"]))
