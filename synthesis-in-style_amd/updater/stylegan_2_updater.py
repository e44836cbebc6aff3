"""One StyleGAN2 training iteration.

Mirrors /root/reference/stylegan_code_finder/updater/stylegan_2_updater.py:19-223: same class name, constructor
keywords (``latent_size``, ``style_mixing_prob``, ``regularization_options`` with ``d_reg_interval`` /
``g_reg_interval`` / ``r1_weight`` / ``path_reg_weight``, ``g_ema``, ``freeze_stochastic_noise_layers``), the same
method names and the same order inside ``update_core``:

  D step (logistic loss on fake / real)  ->  every d_reg_interval: lazy R1 penalty on real images
  G step (non-saturating loss)           ->  every g_reg_interval: path-length regulariser on half a batch
  g_ema <- decay * g_ema + (1 - decay) * G,   decay = 0.5 ** (32 / 10000)

MI355X specifics: both networks run their stride-1 3x3 convolutions on the Winograd MFMA kernels -- first-order
passes and the gradient-of-gradient terms of both regularisers (networks/hip_conv.py ``_Conv3x3Backward``); the
generator's modulated convolutions use one shared-weight convolution per layer instead of B grouped ones
(networks/stylegan2/model.py ``ModulatedConv2d._forward_autograd``); the weight average is two ``_foreach`` launches
instead of a Python loop over 135 tensors.  Gradients cross ranks through DistributedDataParallel (RCCL) when the
networks are wrapped, exactly as in the reference; the only explicit collective is ``reduce_sum`` of the mean path
length (:203-205).
"""
import math
import random
from collections.abc import Iterable
from typing import Tuple

import torch
import torch.nn.functional as F
from torch import autograd, nn
from torch.nn.parallel import DistributedDataParallel

from training.loop import GradientApplier, UpdateDisabler, Updater, get_current_reporter, get_world_size, reduce_sum


class Stylegan2Updater(Updater):
    def __init__(self, *args, latent_size: int = 512, style_mixing_prob: float = 0.9, regularization_options: dict = None,
                 g_ema=None, freeze_stochastic_noise_layers=False, **kwargs):
        super().__init__(*args, **kwargs)
        options = regularization_options or {}
        assert g_ema is not None, "For Training of Stylegan2 we need an accumulation generator!"
        self.g_ema = g_ema
        self.latent_size = latent_size
        self.style_mixing_prob = style_mixing_prob
        self.g_reg_batch_size_shrink_factor = 2
        self.mean_path_length = 0
        self.mean_path_length_avg = 0
        self.accumulation_decay = 0.5 ** (32 / (10 * 1000))
        if isinstance(freeze_stochastic_noise_layers, Iterable):
            self.stochastic_noise_layers_to_freeze = freeze_stochastic_noise_layers
        elif freeze_stochastic_noise_layers:  # freeze all
            self.stochastic_noise_layers_to_freeze = [int(name.split('_')[1]) for name, _ in g_ema.noises.named_buffers()]
        else:
            self.stochastic_noise_layers_to_freeze = []
        self.d_reg_interval = int(options.get('d_reg_interval', 16))
        self.g_reg_interval = int(options.get('g_reg_interval', 4))
        self.r1_weight = float(options.get('r1_weight', 10))
        self.path_reg_weight = float(options.get('path_reg_weight', 2))

    # ---- weight average ---------------------------------------------------------------------
    def accumulate(self, trained_model, decay=0.999):
        if isinstance(trained_model, DistributedDataParallel):
            trained_model = trained_model.module
        source = dict(trained_model.named_parameters())
        names = [k for k, _ in self.g_ema.named_parameters()]
        with torch.no_grad():
            averaged = [p.data for _, p in self.g_ema.named_parameters()]
            torch._foreach_mul_(averaged, decay)
            torch._foreach_add_(averaged, [source[k].data for k in names], alpha=1 - decay)

    # ---- latents and per-layer noise --------------------------------------------------------
    def make_noise(self, batch_size: int, n_noise: int) -> Tuple[torch.Tensor, ...]:
        if n_noise == 1:
            return (torch.randn(batch_size, self.latent_size, device=self.device),)
        return torch.randn(n_noise, batch_size, self.latent_size, device=self.device).unbind(0)

    def make_stochastic_noise(self):
        return [buffer if layer in self.stochastic_noise_layers_to_freeze else None
                for layer, (_, buffer) in enumerate(self.g_ema.noises.named_buffers())]

    def mixing_styles(self, batch_size: int) -> Tuple[torch.Tensor, ...]:
        mix = self.style_mixing_prob > 0 and random.random() < self.style_mixing_prob
        return self.make_noise(batch_size, 2 if mix else 1)

    # ---- losses -----------------------------------------------------------------------------
    def d_logistic_loss(self, real_pred: torch.Tensor, fake_pred: torch.Tensor) -> torch.Tensor:
        return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()

    def d_r1_loss(self, real_pred: torch.Tensor, real_img: torch.Tensor) -> torch.Tensor:
        grad_real, = autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
        return grad_real.pow(2).view(grad_real.shape[0], -1).sum(1).mean()

    def g_nonsaturating_loss(self, fake_pred: torch.Tensor) -> torch.Tensor:
        return F.softplus(-fake_pred).mean()

    def requires_grad(self, network: nn.Module, flag: bool):
        for parameter in network.parameters():
            parameter.requires_grad = flag

    def g_path_regularize(self, fake_img: torch.Tensor, latents: torch.Tensor, mean_path_length, decay: float = 0.01,
                          noise: torch.Tensor = None):
        """``noise`` (optional, a build-side addition for the parity tests): the N(0,1) image the reference draws here."""
        if noise is None:
            noise = torch.randn_like(fake_img)
        noise = noise / math.sqrt(fake_img.shape[2] * fake_img.shape[3])
        grad, = autograd.grad(outputs=(fake_img * noise).sum(), inputs=latents, create_graph=True)
        path_lengths = torch.sqrt(grad.pow(2).sum(2).mean(1))
        path_mean = mean_path_length + decay * (path_lengths.mean() - mean_path_length)
        path_penalty = (path_lengths - path_mean).pow(2).mean()
        return path_penalty, path_mean.detach(), path_lengths

    # ---- the four sub-steps -----------------------------------------------------------------
    def update_discriminator(self, images: torch.Tensor) -> dict:
        generator, discriminator = self.networks['generator'], self.networks['discriminator']
        with UpdateDisabler(generator), GradientApplier([discriminator], [self.optimizers['discriminator']]):
            generated_image, _ = generator(self.mixing_styles(len(images)), noise=self.make_stochastic_noise())
            fake_prediction = discriminator(generated_image)
            real_prediction = discriminator(images)
            d_loss = self.d_logistic_loss(real_prediction, fake_prediction)
            d_loss.backward()
        return {"discriminator_loss": d_loss.detach(), "real_score": real_prediction.mean().detach(),
                "fake_score": fake_prediction.mean().detach()}

    def regularize_discriminator(self, images: torch.Tensor) -> dict:
        discriminator = self.networks['discriminator']
        with GradientApplier([discriminator], [self.optimizers['discriminator']]):
            images.requires_grad = True
            real_pred = discriminator(images)
            r1_loss = self.d_r1_loss(real_pred, images)
            (self.r1_weight / 2 * r1_loss * self.d_reg_interval + 0 * real_pred[0]).backward()
        return {"r1_loss": r1_loss.detach()}

    def update_generator(self, images: torch.Tensor) -> dict:
        generator, discriminator = self.networks['generator'], self.networks['discriminator']
        with UpdateDisabler(discriminator), GradientApplier([generator], [self.optimizers['generator']]):
            fake_images, _ = generator(self.mixing_styles(len(images)), noise=self.make_stochastic_noise())
            g_loss = self.g_nonsaturating_loss(discriminator(fake_images))
            g_loss.backward()
        return {"generator_loss": g_loss.detach()}

    def regularize_generator(self, images: torch.Tensor) -> dict:
        generator = self.networks['generator']
        with GradientApplier([generator], [self.optimizers['generator']]):
            path_batch_size = max(1, len(images) // self.g_reg_batch_size_shrink_factor)
            fake_images, latents = generator(self.mixing_styles(path_batch_size), return_latents=True,
                                             noise=self.make_stochastic_noise())
            path_loss, self.mean_path_length, path_lengths = self.g_path_regularize(fake_images, latents,
                                                                                    self.mean_path_length)
            weighted_path_loss = self.path_reg_weight * self.g_reg_interval * path_loss
            if self.g_reg_batch_size_shrink_factor:
                weighted_path_loss += 0 * fake_images[0, 0, 0, 0]
            weighted_path_loss.backward()
            self.mean_path_length_avg = reduce_sum(self.mean_path_length).item() / get_world_size()
        return {"perceputal_path_loss": path_loss.detach(), "perceptual_path_lengths": path_lengths.mean().detach()}

    def update_core(self):
        batch = next(self.iterators['images'])
        batch = {key: value.to(self.device) for key, value in batch.items()}
        reporter = get_current_reporter()
        reporter.add_observation(self.update_discriminator(batch['image']), 'discriminator')
        if self.iteration % self.d_reg_interval == 0:
            reporter.add_observation(self.regularize_discriminator(batch['image']), 'discriminator')
        reporter.add_observation(self.update_generator(batch['image']), 'generator')
        if self.iteration % self.g_reg_interval == 0:
            reporter.add_observation(self.regularize_generator(batch['image']), 'generator')
        self.accumulate(self.networks['generator'], self.accumulation_decay)
