"""Loss terms of StyleGAN2 training as plain functions (reference: methods of updater/stylegan_2_updater.py).

  logistic_discriminator_loss   softplus(-D(real)) + softplus(D(fake))                                   :82-86
  r1_penalty                    E_b || d sum(D(real)) / d real ||^2, graph kept for the second backward   :88-94
  nonsaturating_generator_loss  softplus(-D(G(z)))                                                        :96-99
  path_length_penalty           || J_w^T y || / sqrt(n_latent-mean) against its running mean              :105-120
"""
import math

import torch
from torch.nn.functional import softplus


def logistic_discriminator_loss(real_pred, fake_pred):
    return softplus(-real_pred).mean() + softplus(fake_pred).mean()


def nonsaturating_generator_loss(fake_pred):
    return softplus(-fake_pred).mean()


def r1_penalty(real_pred, real_img):
    (grad_real,) = torch.autograd.grad(real_pred.sum(), real_img, create_graph=True)
    return grad_real.square().flatten(1).sum(1).mean()


def path_length_penalty(fake_img, latents, mean_path_length, decay=0.01, noise=None):
    """-> (penalty, new running mean (detached), per-sample path lengths).  ``noise`` (a build-side addition for the
    parity tests): the N(0,1) image the reference draws here with ``randn_like``."""
    if noise is None:
        noise = torch.randn_like(fake_img)
    pixels = fake_img.shape[2] * fake_img.shape[3]
    (jac,) = torch.autograd.grad((fake_img * (noise / math.sqrt(pixels))).sum(), latents, create_graph=True)
    lengths = jac.square().sum(2).mean(1).sqrt()
    running = mean_path_length + decay * (lengths.mean() - mean_path_length)
    return (lengths - running).square().mean(), running.detach(), lengths
