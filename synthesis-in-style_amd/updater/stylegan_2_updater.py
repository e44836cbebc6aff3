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
import contextlib
import random
from collections.abc import Iterable
from typing import Tuple

import torch
from torch import nn
from torch.nn.parallel import DistributedDataParallel

from updater import gan_losses
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

    # ---- losses (updater/gan_losses.py; kept as methods because callers and subclasses use them as such) -------------
    def d_logistic_loss(self, real_pred: torch.Tensor, fake_pred: torch.Tensor) -> torch.Tensor:
        return gan_losses.logistic_discriminator_loss(real_pred, fake_pred)

    def d_r1_loss(self, real_pred: torch.Tensor, real_img: torch.Tensor) -> torch.Tensor:
        return gan_losses.r1_penalty(real_pred, real_img)

    def g_nonsaturating_loss(self, fake_pred: torch.Tensor) -> torch.Tensor:
        return gan_losses.nonsaturating_generator_loss(fake_pred)

    def requires_grad(self, network: nn.Module, flag: bool):
        for parameter in network.parameters():
            parameter.requires_grad = flag

    def g_path_regularize(self, fake_img: torch.Tensor, latents: torch.Tensor, mean_path_length, decay: float = 0.01,
                          noise: torch.Tensor = None):
        return gan_losses.path_length_penalty(fake_img, latents, mean_path_length, decay, noise)

    # ---- the four sub-steps -----------------------------------------------------------------
    def _fakes(self, count: int, **generator_kwargs):
        """G(mixing styles) with the frozen / fresh per-layer noise selection of this run."""
        return self.networks['generator'](self.mixing_styles(count), noise=self.make_stochastic_noise(), **generator_kwargs)

    def _training(self, trained: str, frozen: str = None):
        """Context of one sub-step: zero_grad / step of ``trained``'s optimizer around it, ``frozen``'s parameters
        switched off inside it (for the D step that sends the generator down its fused inference path)."""
        stack = contextlib.ExitStack()
        if frozen is not None:
            stack.enter_context(UpdateDisabler(self.networks[frozen]))
        stack.enter_context(GradientApplier([self.networks[trained]], [self.optimizers[trained]]))
        return stack

    def update_discriminator(self, images: torch.Tensor) -> dict:
        critic = self.networks['discriminator']
        with self._training('discriminator', frozen='generator'):
            fake_score = critic(self._fakes(len(images))[0])
            real_score = critic(images)
            loss = self.d_logistic_loss(real_score, fake_score)
            loss.backward()
        return {"discriminator_loss": loss.detach(), "real_score": real_score.mean().detach(),
                "fake_score": fake_score.mean().detach()}

    def regularize_discriminator(self, images: torch.Tensor) -> dict:
        with self._training('discriminator'):
            images.requires_grad = True
            real_score = self.networks['discriminator'](images)
            penalty = self.d_r1_loss(real_score, images)
            # lazy regularisation: the weight is scaled by the interval; "+ 0 * score" keeps every output in the graph (DDP)
            (self.r1_weight / 2 * penalty * self.d_reg_interval + 0 * real_score[0]).backward()
        return {"r1_loss": penalty.detach()}

    def update_generator(self, images: torch.Tensor) -> dict:
        with self._training('generator', frozen='discriminator'):
            loss = self.g_nonsaturating_loss(self.networks['discriminator'](self._fakes(len(images))[0]))
            loss.backward()
        return {"generator_loss": loss.detach()}

    def regularize_generator(self, images: torch.Tensor) -> dict:
        with self._training('generator'):
            count = max(1, len(images) // self.g_reg_batch_size_shrink_factor)
            fakes, latents = self._fakes(count, return_latents=True)
            penalty, self.mean_path_length, lengths = self.g_path_regularize(fakes, latents, self.mean_path_length)
            weighted = self.path_reg_weight * self.g_reg_interval * penalty
            if self.g_reg_batch_size_shrink_factor:
                weighted = weighted + 0 * fakes[0, 0, 0, 0]
            weighted.backward()
            self.mean_path_length_avg = reduce_sum(self.mean_path_length).item() / get_world_size()
        return {"perceputal_path_loss": penalty.detach(), "perceptual_path_lengths": lengths.mean().detach()}  # (sic) reference keys

    def update_core(self):
        batch = {key: value.to(self.device) for key, value in next(self.iterators['images']).items()}
        images, report = batch['image'], get_current_reporter().add_observation
        report(self.update_discriminator(images), 'discriminator')
        if self.iteration % self.d_reg_interval == 0:
            report(self.regularize_discriminator(images), 'discriminator')
        report(self.update_generator(images), 'generator')
        if self.iteration % self.g_reg_interval == 0:
            report(self.regularize_generator(images), 'generator')
        self.accumulate(self.networks['generator'], self.accumulation_decay)
