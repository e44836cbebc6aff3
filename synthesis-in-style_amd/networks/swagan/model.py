"""SWAGAN generator: StyleGAN2 synthesis in the Haar-wavelet domain (reference: networks/swagan/model.py:14-285;
selected by ``stylegan_variant: 'swagan'``, networks/__init__.py:356-362,396-403).

Same constructor, attributes, state_dict keys (incl. the Haar filter buffers ``*.iwt.ll`` ... ``*.dwt.hh``) and
``forward`` signature as the reference.  The layers are this package's StyleGAN2 modules, so in inference every
modulated convolution runs on the MI355X kernels of libsis_hip.so (stride-1 3x3 -> Winograd, upsampling 3x3 ->
4-phase transposed kernel + blur, 1x1 ToRGB -> direct MFMA kernel); the wavelet transforms are four polyphase
``upfirdn2d`` passes each (K2 with the +/- 1/2 Haar taps: modes up=2 / down=2 with pad (1, 0) / (0, 0)).
"""
import math

import torch
from torch import nn

from networks.stylegan2.model import (ConstantInput, EqualLinear, ModulatedConv2d, PixelNorm, StyledConv, Upsample,
                                      resolve_latents)
from .op import upfirdn2d


def get_haar_wavelet(in_channels):
    """The four 2x2 Haar analysis filters (model.py:14-24; ``in_channels`` is unused there as well)."""
    low = 1 / (2 ** 0.5) * torch.ones(1, 2)
    high = 1 / (2 ** 0.5) * torch.ones(1, 2)
    high[0, 0] = -1 * high[0, 0]
    return low.T * low, high.T * low, low.T * high, high.T * high


class HaarTransform(nn.Module):
    """[B,C,H,W] -> [B,4C,H/2,W/2]: LL, LH, HL, HH sub-bands stacked on the channel axis (model.py:27-46)."""

    def __init__(self, in_channels):
        super().__init__()
        ll, lh, hl, hh = get_haar_wavelet(in_channels)
        self.register_buffer('ll', ll)
        self.register_buffer('lh', lh)
        self.register_buffer('hl', hl)
        self.register_buffer('hh', hh)

    def forward(self, input):
        return torch.cat([upfirdn2d(input, k, down=2) for k in (self.ll, self.lh, self.hl, self.hh)], 1)


class InverseHaarTransform(nn.Module):
    """[B,4C,H,W] -> [B,C,2H,2W] (model.py:48-67; LH / HL synthesis filters are negated)."""

    def __init__(self, in_channels):
        super().__init__()
        ll, lh, hl, hh = get_haar_wavelet(in_channels)
        self.register_buffer('ll', ll)
        self.register_buffer('lh', -lh)
        self.register_buffer('hl', -hl)
        self.register_buffer('hh', hh)

    def forward(self, input):
        bands = input.chunk(4, 1)
        out = None
        for band, k in zip(bands, (self.ll, self.lh, self.hl, self.hh)):
            up = upfirdn2d(band, k, up=2, pad=(1, 0, 1, 0))
            out = up if out is None else out + up
        return out


class ToRGB(nn.Module):
    """1x1 modulated conv to the 12 wavelet coefficients of an RGB image + bias + the previous resolution's
    coefficients taken through IWT -> Upsample -> DWT (model.py:70-95)."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.iwt = InverseHaarTransform(3)
            self.upsample = Upsample(blur_kernel)
            self.dwt = HaarTransform(3)
        self.conv = ModulatedConv2d(in_channel, 3 * 4, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3 * 4, 1, 1))

    def forward(self, input, style, skip=None):
        out = self.conv(input, style) + self.bias
        if skip is not None:
            out = out + self.dwt(self.upsample(self.iwt(skip)))
        return out


class Generator(nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        self.style = nn.Sequential(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp,
                                                              activation="fused_lrelu") for _ in range(n_mlp)])
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
                         256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)

        self.log_size = int(math.log(size, 2)) - 1  # the last doubling is the inverse wavelet transform
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel
        self.iwt = InverseHaarTransform(3)
        self.n_latent = self.log_size * 2 - 2

    def make_noise(self):
        device = self.input.input.device
        sizes = [4] + [2 ** i for i in range(3, self.log_size + 1) for _ in range(2)]
        return [torch.randn(1, 1, n, n, device=device) for n in sizes]

    def mean_latent(self, n_latent):
        return self.style(torch.randn(n_latent, self.style_dim, device=self.input.input.device)).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, return_intermediate_activations=False):
        latent, noise = resolve_latents(self, styles, inject_index, truncation, truncation_latent, input_is_latent,
                                        noise, randomize_noise)

        acts = {} if return_intermediate_activations else None

        def tap(idx, t):
            if acts is not None:
                acts[idx] = t.detach()  # fresh tensors: no clone needed to keep them

        out = self.input(latent)
        tap(0, out)
        out = self.conv1(out, latent[:, 0], noise=noise[0])
        tap(1, out)
        skip = self.to_rgb1(out, latent[:, 1])
        i = 1
        for conv1, conv2, noise1, noise2, to_rgb in zip(self.convs[::2], self.convs[1::2], noise[1::2], noise[2::2],
                                                        self.to_rgbs):
            out = conv1(out, latent[:, i], noise=noise1)
            tap(i + 1, out)
            out = conv2(out, latent[:, i + 1], noise=noise2)
            tap(i + 2, out)
            skip = to_rgb(out, latent[:, i + 2], skip)
            i += 2
        image = self.iwt(skip)
        if return_latents:
            return image, latent
        if return_intermediate_activations:
            return image, acts
        return image, None


from .discriminator import ConvBlock, Discriminator, FromRGB  # noqa: E402,F401
