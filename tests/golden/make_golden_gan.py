"""Golden vectors of one StyleGAN2 training iteration's four sub-steps from the UNMODIFIED
reference networks/stylegan2/model.py (Generator + Discriminator), imported with the oracle's CPU ops as its ``.op``
(oracle/load_reference.py::load_reference_stylegan2; autograd differentiates those twice).

The loss arithmetic is restated from updater/stylegan_2_updater.py (the module itself needs the un-vendored
``pytorch_training`` package and ``collections.Iterable``, gone in Python 3.10): d_logistic_loss :82-86, d_r1_loss
:88-94, g_nonsaturating_loss :96-99, g_path_regularize :105-120, and the backward expressions of
regularize_discriminator :150-158 / regularize_generator :178-203.

    python tests/golden/make_golden_gan.py      -> tests/golden/gan32.npz
Weights / inputs are re-derived in the tests from oracle.stylegan2_ref seeded_* helpers and numpy RandomState streams.
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from torch import autograd

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import stylegan2_ref as R  # noqa: E402
from oracle.load_reference import load_reference_stylegan2  # noqa: E402

SIZE, STYLE_DIM, N_MLP, CM, B = 32, 64, 2, 1, 4
ref = load_reference_stylegan2()
g = ref.Generator(SIZE, STYLE_DIM, N_MLP, channel_multiplier=CM)
g.load_state_dict(R.seeded_state_dict(SIZE, STYLE_DIM, N_MLP, CM, seed=11), strict=True)
d = ref.Discriminator(SIZE, channel_multiplier=CM)
assert [(k, tuple(v.shape)) for k, v in d.state_dict().items()] == R.discriminator_schema(SIZE, CM)
d.load_state_dict(R.seeded_discriminator_state_dict(SIZE, CM, seed=12), strict=True)
g.train(), d.train()

rng = np.random.RandomState(13)
z1 = torch.from_numpy(rng.standard_normal((B, STYLE_DIM)).astype(np.float32))
z2 = torch.from_numpy(rng.standard_normal((B, STYLE_DIM)).astype(np.float32))
real = torch.from_numpy(rng.uniform(-1, 1, (B, 3, SIZE, SIZE)).astype(np.float32))
path_noise = torch.from_numpy(rng.standard_normal((B // 2, 3, SIZE, SIZE)).astype(np.float32))
_, noise = R.seeded_inputs(SIZE, 1, STYLE_DIM, seed=14)
out = {"cfg": np.asarray([SIZE, STYLE_DIM, N_MLP, CM, B])}


def grads(net, tag):
    for name, p in net.named_parameters():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"{tag}/norm/{name}"] = np.asarray(gr.double().norm().item())
        out[f"{tag}/head/{name}"] = gr.flatten()[:16].numpy().copy()
    net.zero_grad(set_to_none=True)


# D step: generator frozen
for p in g.parameters():
    p.requires_grad = False
fake, _ = g([z1, z2], inject_index=3, noise=noise)
fake_pred, real_pred = d(fake), d(real)
d_loss = F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()
d_loss.backward()
out.update(fake=fake.detach().numpy(), fake_pred=fake_pred.detach().numpy(), real_pred=real_pred.detach().numpy(),
           d_loss=np.asarray(d_loss.item()))
grads(d, "d_step")

# lazy R1
img = real.clone().requires_grad_(True)
real_pred = d(img)
grad_real, = autograd.grad(outputs=real_pred.sum(), inputs=img, create_graph=True)
r1 = grad_real.pow(2).view(B, -1).sum(1).mean()
(10 / 2 * r1 * 16 + 0 * real_pred[0]).backward()
out.update(r1_loss=np.asarray(r1.item()), r1_grad_real=grad_real.detach().numpy())
grads(d, "d_reg")

# G step: discriminator frozen
for p in g.parameters():
    p.requires_grad = True
for p in d.parameters():
    p.requires_grad = False
fake, _ = g([z1, z2], inject_index=3, noise=noise)
g_loss = F.softplus(-d(fake)).mean()
g_loss.backward()
out.update(g_loss=np.asarray(g_loss.item()))
grads(g, "g_step")

# path-length regulariser on half the batch
fake, latents = g([z1[:B // 2], z2[:B // 2]], return_latents=True, inject_index=3, noise=noise)
pn = path_noise / math.sqrt(SIZE * SIZE)
grad, = autograd.grad(outputs=(fake * pn).sum(), inputs=latents, create_graph=True)
path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
path_mean = 0 + 0.01 * (path_lengths.mean() - 0)
penalty = (path_lengths - path_mean).pow(2).mean()
weighted = 2 * 4 * penalty
weighted = weighted + 0 * fake[0, 0, 0, 0]
weighted.backward()
out.update(path_penalty=np.asarray(penalty.item()), path_mean=np.asarray(path_mean.item()),
           path_lengths=path_lengths.detach().numpy(), path_grad=grad.detach().numpy())
grads(g, "g_reg")

np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gan32.npz"), **out)
print({k: float(v) for k, v in out.items() if v.ndim == 0})
