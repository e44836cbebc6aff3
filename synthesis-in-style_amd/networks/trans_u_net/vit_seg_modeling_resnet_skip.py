"""ResNetV2 stem of the hybrid TransUNet encoder: weight-standardised convolutions + GroupNorm, pre-activation
bottlenecks, three skip features for the decoder.

Drop-in for /root/reference/stylegan_code_finder/networks/trans_u_net/vit_seg_modeling_resnet_skip.py
(StdConv2d :20-27, PreActBottleneck :40-111, ResNetV2 :114-162): same module / parameter names (so the
``.npz`` loader and checkpoints line up), same arithmetic -- weights standardised on every forward as
(w - mean) / sqrt(var + 1e-5) per output filter, GroupNorm(32, eps 1e-6), a projection GroupNorm with one
group per channel, 3x3 max-pool stride 2 without padding, skip maps zero-padded up to S/4, S/8 when the
un-padded pooling leaves them one or two pixels short.
"""
from collections import OrderedDict
from os.path import join as pjoin

import torch
import torch.nn as nn
import torch.nn.functional as F


def np2th(weights, conv=False):
    """numpy HWIO -> torch OIHW for convolution kernels."""
    if conv:
        weights = weights.transpose([3, 2, 0, 1])
    return torch.from_numpy(weights)


class StdConv2d(nn.Conv2d):
    def standardized_weight(self):
        w = self.weight
        var, mean = torch.var_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
        return (w - mean) / torch.sqrt(var + 1e-5)

    def forward(self, x):
        return F.conv2d(x, self.standardized_weight(), self.bias, self.stride, self.padding, self.dilation,
                        self.groups)


def conv3x3(cin, cout, stride=1, groups=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=bias, groups=groups)


def conv1x1(cin, cout, stride=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=1, stride=stride, padding=0, bias=bias)


class PreActBottleneck(nn.Module):
    def __init__(self, cin, cout=None, cmid=None, stride=1):
        super().__init__()
        cout = cout or cin
        cmid = cmid or cout // 4
        self.gn1 = nn.GroupNorm(32, cmid, eps=1e-6)
        self.conv1 = conv1x1(cin, cmid, bias=False)
        self.gn2 = nn.GroupNorm(32, cmid, eps=1e-6)
        self.conv2 = conv3x3(cmid, cmid, stride, bias=False)
        self.gn3 = nn.GroupNorm(32, cout, eps=1e-6)
        self.conv3 = conv1x1(cmid, cout, bias=False)
        self.relu = nn.ReLU(inplace=True)
        if stride != 1 or cin != cout:
            self.downsample = conv1x1(cin, cout, stride, bias=False)
            self.gn_proj = nn.GroupNorm(cout, cout)

    def forward(self, x):
        residual = self.gn_proj(self.downsample(x)) if hasattr(self, 'downsample') else x
        y = self.relu(self.gn1(self.conv1(x)))
        y = self.relu(self.gn2(self.conv2(y)))
        y = self.gn3(self.conv3(y))
        return self.relu(residual + y)

    def load_from(self, weights, n_block, n_unit):
        def w(name, conv=False):
            return np2th(weights[pjoin(n_block, n_unit, name)], conv=conv)

        with torch.no_grad():
            for i in (1, 2, 3):
                getattr(self, f'conv{i}').weight.copy_(w(f'conv{i}/kernel', conv=True))
                getattr(self, f'gn{i}').weight.copy_(w(f'gn{i}/scale').view(-1))
                getattr(self, f'gn{i}').bias.copy_(w(f'gn{i}/bias').view(-1))
            if hasattr(self, 'downsample'):
                self.downsample.weight.copy_(w('conv_proj/kernel', conv=True))
                self.gn_proj.weight.copy_(w('gn_proj/scale').view(-1))
                self.gn_proj.bias.copy_(w('gn_proj/bias').view(-1))


class ResNetV2(nn.Module):
    def __init__(self, block_units, width_factor):
        super().__init__()
        width = int(64 * width_factor)
        self.width = width
        self.root = nn.Sequential(OrderedDict([
            ('conv', StdConv2d(3, width, kernel_size=7, stride=2, bias=False, padding=3)),
            ('gn', nn.GroupNorm(32, width, eps=1e-6)),
            ('relu', nn.ReLU(inplace=True)),
        ]))

        def stage(cin, cout, cmid, n, stride):
            units = [('unit1', PreActBottleneck(cin=cin, cout=cout, cmid=cmid, stride=stride))]
            units += [(f'unit{i:d}', PreActBottleneck(cin=cout, cout=cout, cmid=cmid)) for i in range(2, n + 1)]
            return nn.Sequential(OrderedDict(units))

        self.body = nn.Sequential(OrderedDict([
            ('block1', stage(width, width * 4, width, block_units[0], 1)),
            ('block2', stage(width * 4, width * 8, width * 2, block_units[1], 2)),
            ('block3', stage(width * 8, width * 16, width * 4, block_units[2], 2)),
        ]))

    def forward(self, x):
        b, _, in_size, _ = x.size()
        x = self.root(x)
        features = [x]
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=0)
        for i in range(len(self.body) - 1):
            x = self.body[i](x)
            right_size = int(in_size / 4 / (i + 1))
            if x.size(2) != right_size:
                pad = right_size - x.size(2)
                assert 0 < pad < 3, f"x {x.size()} should {right_size}"
                feat = F.pad(x, (0, pad, 0, pad))
            else:
                feat = x
            features.append(feat)
        x = self.body[-1](x)
        return x, features[::-1]
