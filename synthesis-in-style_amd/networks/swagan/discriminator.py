"""SWAGAN discriminator (GAN training): a wavelet image pyramid feeding a plain convolution stack.

Module surface and state_dict schema of the reference (networks/swagan/model.py: ConvBlock :288-299, FromRGB :302-326,
Discriminator :329-392; golden run in tests/golden/swagan_d32.npz).  Layers are this package's StyleGAN2 discriminator
layers (``networks/stylegan2/discriminator.py``: stride-1 3x3 convolutions on the Winograd kernels, blur / Haar filters
through the HIP ``upfirdn2d``, bias + leaky-ReLU through ``fused_leaky_relu``).
"""
import math

from torch import nn

from networks.stylegan2.discriminator import ConvLayer, Downsample, minibatch_stddev
from networks.stylegan2.model import EqualLinear
from .model import HaarTransform, InverseHaarTransform


class ConvBlock(nn.Module):
    """3x3 conv -> blur + stride-2 3x3 conv, both with bias / leaky-ReLU (no skip: the wavelet pyramid is the skip path)."""

    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)

    def forward(self, input):
        return self.conv2(self.conv1(input))


class FromRGB(nn.Module):
    """Wavelet pyramid step: (inverse Haar -> FIR decimation -> Haar) of the 12-band image, then a 1x1 conv of the bands
    added to the feature path.  Returns (bands at this level, features)."""

    def __init__(self, out_channel, downsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.downsample = downsample
        if downsample:
            self.iwt = InverseHaarTransform(3)
            self.downsample = Downsample(blur_kernel)
            self.dwt = HaarTransform(3)
        self.conv = ConvLayer(3 * 4, out_channel, 1)

    def forward(self, input, skip=None):
        if self.downsample:
            input = self.dwt(self.downsample(self.iwt(input)))
        out = self.conv(input)
        return input, out if skip is None else out + skip


class Discriminator(nn.Module):
    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        channels = {4: 512, 8: 512, 16: 512, 32: 512}
        channels.update({2 ** (6 + i): (256 >> i) * channel_multiplier for i in range(5)})
        self.dwt = HaarTransform(3)
        self.from_rgbs = nn.ModuleList()
        self.convs = nn.ModuleList()
        log_size = int(math.log(size, 2)) - 1
        width = channels[size]
        for i in range(log_size, 2, -1):
            self.from_rgbs.append(FromRGB(width, downsample=i != log_size))
            self.convs.append(ConvBlock(width, channels[2 ** (i - 1)], blur_kernel))
            width = channels[2 ** (i - 1)]
        self.from_rgbs.append(FromRGB(channels[4]))
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(width + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu'),
                                          EqualLinear(channels[4], 1))

    def forward(self, input):
        bands, out = self.dwt(input), None
        for from_rgb, conv in zip(self.from_rgbs, self.convs):
            bands, out = from_rgb(bands, out)
            out = conv(out)
        _, out = self.from_rgbs[-1](bands, out)
        out = self.final_conv(minibatch_stddev(out, self.stddev_group, self.stddev_feat))
        return self.final_linear(out.view(out.shape[0], -1))
