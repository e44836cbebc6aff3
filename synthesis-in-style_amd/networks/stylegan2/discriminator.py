"""StyleGAN2 discriminator (GAN-training half of networks/stylegan2/model.py).

Same module surface and ``state_dict`` schema as the reference (/root/reference/stylegan_code_finder/networks/
stylegan2/model.py: ScaledLeakyReLU :186-196, EqualConv2d :94-131, ConvLayer :564-609, ResBlock :612-631,
Discriminator :634-692), so a checkpoint's ``ckpt['d']`` loads with ``strict=True``:

  convs.0            1x1 from-RGB conv + bias/leaky-ReLU
  convs.1..n         ResBlock: 3x3 conv -> (blur, 3x3 stride-2 conv) with a (blur, 1x1 stride-2) skip, sum / sqrt(2)
  final_conv         3x3 conv on [features, minibatch-stddev map]
  final_linear       EqualLinear(C*16 -> C, fused leaky-ReLU) -> EqualLinear(C -> 1)

How a layer executes on MI355X:
* stride-1 3x3 convolutions run on the Winograd MFMA kernel (``networks.hip_conv.conv3x3``: forward, data gradient,
  Winograd-domain weight gradient, and the second-order terms the R1 penalty needs -- ``_Conv3x3Backward``);
* blur = ``upfirdn2d`` and bias + leaky-ReLU * sqrt(2) = ``fused_leaky_relu`` on the HIP kernels of ``networks.
  stylegan2.op`` through their (twice differentiable) autograd Functions;
* stride-2 and 1x1 convolutions stay on the library (MIOpen / hipBLASLt), which autograd can differentiate twice.
"""
import math

import torch
from torch import nn
from torch.nn import functional as F

import sis_hip
from networks.hip_conv import conv3x3, gan_winograd_enabled
from .op import FusedLeakyReLU


class ScaledLeakyReLU(nn.Module):
    def __init__(self, negative_slope=0.2):
        super().__init__()
        self.negative_slope = negative_slope

    def forward(self, input):
        return F.leaky_relu(input, negative_slope=self.negative_slope) * math.sqrt(2)


class Downsample(nn.Module):
    """x1/2 FIR decimation (model.py:55-73): normalised taps, pad ((p+1)//2, p//2) with p = taps - factor."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        from .model import make_kernel
        from .op import upfirdn2d
        self._op = upfirdn2d
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2, p // 2)

    def forward(self, input):
        return self._op(input, self.kernel, up=1, down=self.factor, pad=self.pad)


class EqualConv2d(nn.Module):
    """Convolution with equalised learning rate: unit-variance weights, scaled by 1/sqrt(fan_in) at run time."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride = stride
        self.padding = padding
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None

    def _winograd(self, input, weight):
        return (self.stride == 1 and self.padding == 1 and self.bias is None and input.is_contiguous()
                and not torch.is_autocast_enabled() and gan_winograd_enabled() and sis_hip.conv3x3_supported(input, weight))

    def forward(self, input):
        weight = self.weight * self.scale
        if weight.shape[2] == 3 and input.is_cuda and self._winograd(input, weight):
            return conv3x3(input, weight)
        return F.conv2d(input, weight, bias=self.bias, stride=self.stride, padding=self.padding)

    def __repr__(self):
        o, i, k, _ = self.weight.shape
        return f'{self.__class__.__name__}({i}, {o}, {k}, stride={self.stride}, padding={self.padding})'


class ConvLayer(nn.Sequential):
    """[Blur ->] EqualConv2d [-> FusedLeakyReLU | ScaledLeakyReLU]; child indices as in the reference (they are the
    state_dict keys): a downsampling layer is blur (index 0) + stride-2 conv without padding (index 1)."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=[1, 3, 3, 1], bias=True,
                 activate=True):
        from .model import Blur
        layers = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            layers.append(Blur(blur_kernel, pad=((p + 1) // 2, p // 2)))
            stride, self.padding = 2, 0
        else:
            stride, self.padding = 1, kernel_size // 2
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=self.padding, stride=stride,
                                  bias=bias and not activate))
        if activate:
            layers.append(FusedLeakyReLU(out_channel) if bias else ScaledLeakyReLU(0.2))
        super().__init__(*layers)


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, activate=False, bias=False)

    def forward(self, input):
        return (self.conv2(self.conv1(input)) + self.skip(input)) / math.sqrt(2)


def minibatch_stddev(features, group_size, n_feat=1):
    """One extra channel per ``n_feat``: the standard deviation over groups of ``group_size`` samples, averaged over
    channels and pixels, broadcast back to every sample and pixel (model.py:675-683)."""
    b, c, h, w = features.shape
    group = min(b, group_size)
    grouped = features.view(group, -1, n_feat, c // n_feat, h, w)
    std = torch.sqrt(grouped.var(0, unbiased=False) + 1e-8)
    std = std.mean([2, 3, 4], keepdims=True).squeeze(2)
    return torch.cat([features, std.repeat(group, 1, h, w)], 1)


class Discriminator(nn.Module):
    def __init__(self, size, channel_multiplier=2, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        from .model import EqualLinear
        channels = {4: 512, 8: 512, 16: 512, 32: 512}
        channels.update({2 ** (6 + i): (256 >> i) * channel_multiplier for i in range(5)})
        log_size = int(math.log(size, 2))
        width = channels[size]
        convs = [ConvLayer(3, width, 1)]
        for i in range(log_size, 2, -1):
            convs.append(ResBlock(width, channels[2 ** (i - 1)], blur_kernel))
            width = channels[2 ** (i - 1)]
        self.convs = nn.Sequential(*convs)
        self.stddev_group = 4
        self.stddev_feat = 1
        self.final_conv = ConvLayer(width + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation='fused_lrelu'),
                                          EqualLinear(channels[4], 1))

    def forward(self, input):
        sis_hip.require_device(input, "input")
        out = minibatch_stddev(self.convs(input), self.stddev_group, self.stddev_feat)
        out = self.final_conv(out)
        return self.final_linear(out.view(out.shape[0], -1))
