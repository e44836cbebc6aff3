"""Cascaded-upsampling CNN decoder ("CUP") and segmentation head of TransUNet
(reference: networks/trans_u_net/vit_seg_modeling.py:265-373): tokens -> [B, hidden, h, w] -> 3x3 conv to 512 ->
four (bilinear x2, concat skip, 2 x conv-BN-ReLU) stages -> 3x3 head."""
import contextlib
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

import sis_hip

from networks.hip_conv import HipConv2d
from networks.hip_upsample import HipUpsamplingBilinear2d, upsample2x_cat

_FUSE_UP_CAT = os.environ.get('SIS_FUSE_UP_CAT', '1') != '0'  # 0: upsampling and torch.cat as two steps (A/B runs)
_DECODER_BANK = os.environ.get('SIS_DECODER_BANK', '1') != '0'  # 0: one pack launch per decoder convolution, one counter add per norm


class _BatchNormAct(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, relu):
        y, mean, rstd = sis_hip.batch_norm_train_fwd(x, weight, bias, running_mean, running_var, eps, momentum, relu)
        ctx.save_for_backward(x, mean, rstd, weight, bias)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, grad):
        x, mean, rstd, weight, bias = ctx.saved_tensors
        dx, dgamma, dbeta = sis_hip.batch_norm_train_bwd(grad, x, mean, rstd, weight, bias, ctx.relu)
        return dx, dgamma, dbeta, None, None, None, None, None


class HipBatchNorm2d(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` whose training-mode forward can apply the following ReLU and reads / writes the 16-bit
    tensors of the neighbouring convolutions directly under autocast (csrc/group_norm.hip, batch-norm mode)."""

    _counted_in_bulk = False   # this forward's num_batches_tracked += 1 already happened (decoder_step_state)

    def forward(self, x, relu=False):
        if (self.training and x.is_cuda and self.affine and self.track_running_stats and self.momentum is not None
                and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and x.dim() == 4):
            if self.num_batches_tracked is not None and not self._counted_in_bulk:
                self.num_batches_tracked.add_(1)
            return _BatchNormAct.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.eps,
                                       self.momentum, relu)
        y = super().forward(x)
        return F.relu(y) if relu else y


class Conv2dReLU(nn.Sequential):
    """conv -> batch norm -> ReLU with the reference's child indices (0, 1, 2); norm and ReLU run as one kernel."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0, stride=1, use_batchnorm=True):
        super().__init__(HipConv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   bias=not use_batchnorm),
                         HipBatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def forward(self, x):
        return self[1](self[0](x), relu=True)


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, skip_channels=0, use_batchnorm=True):
        super().__init__()
        self.conv1 = Conv2dReLU(in_channels + skip_channels, out_channels, kernel_size=3, padding=1,
                                use_batchnorm=use_batchnorm)
        self.conv2 = Conv2dReLU(out_channels, out_channels, kernel_size=3, padding=1, use_batchnorm=use_batchnorm)
        self.up = HipUpsamplingBilinear2d(scale_factor=2)

    def forward(self, x, skip=None):
        fused = upsample2x_cat(x, skip) if (skip is not None and _FUSE_UP_CAT and self.up.scale_factor == 2) else None
        if fused is not None:
            x = fused   # upsampling written straight into the concatenated tensor
        else:
            x = self.up(x)
            if skip is not None:
                x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class SegmentationHead(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, upsampling=1):
        super().__init__(HipConv2d(in_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2),
                         HipUpsamplingBilinear2d(scale_factor=upsampling) if upsampling > 1 else nn.Identity())


class DecoderCup(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        head_channels = 512
        self.conv_more = Conv2dReLU(config.hidden_size, head_channels, kernel_size=3, padding=1, use_batchnorm=True)
        decoder_channels = config.decoder_channels
        in_channels = [head_channels] + list(decoder_channels[:-1])
        if self.config.n_skip != 0:
            skip_channels = self.config.skip_channels
            for i in range(4 - self.config.n_skip):  # unused skips contribute no channels
                skip_channels[3 - i] = 0
        else:
            skip_channels = [0, 0, 0, 0]
        self.blocks = nn.ModuleList([DecoderBlock(i, o, s) for i, o, s in zip(in_channels, decoder_channels,
                                                                              skip_channels)])

    def forward(self, hidden_states, features=None):
        b, n_patch, hidden = hidden_states.size()
        h = w = int(np.sqrt(n_patch))
        if hidden_states.is_cuda and hidden_states.is_contiguous() and hidden_states.element_size() in (2, 4):
            tokens_last = sis_hip.swap_last2(hidden_states)   # [B, hidden, n_patch]: one tiled transpose per direction
        else:
            tokens_last = hidden_states.permute(0, 2, 1).contiguous()
        x = self.conv_more(tokens_last.view(b, hidden, h, w))
        for i, block in enumerate(self.blocks):
            skip = features[i] if (features is not None and i < self.config.n_skip) else None
            x = block(x, skip=skip)
        return x


def _bankable(m):
    """A decoder / head convolution whose packed bf16 images the bank can write (what ``HipConv2d._bf16`` would accept)."""
    return (isinstance(m, HipConv2d) and m.weight.is_cuda and m.groups == 1 and m.padding_mode == 'zeros' and m.dilation == (1, 1)
            and m.stride[0] == m.stride[1] and m.kernel_size[0] == m.kernel_size[1]
            and m.padding == (m.kernel_size[0] // 2, m.kernel_size[0] // 2) and sis_hip.WeightStdPackBank.supported(m.weight, m.stride[0]))


@contextlib.contextmanager
def decoder_step_state(model):
    """Per-forward state of the decoder + head that does not depend on the activations, set up by TWO launches instead of one
    per layer (12 ``conv_pack`` + 10 counter increments per TransUNet step before): the packed forward / adjoint bf16 images of
    every convolution weight (``sis_hip.WeightStdPackBank`` in its plain mode) and ``num_batches_tracked += 1`` of every
    training-mode batch norm (``torch._foreach_add_``).  Only under bf16 autocast on the GPU; otherwise a no-op."""
    from networks.hip_conv import _BF16_CONV
    convs, norms = [], []
    first = next(model.decoder.parameters())
    if (_DECODER_BANK and _BF16_CONV and first.is_cuda and torch.is_autocast_enabled()
            and torch.get_autocast_dtype('cuda') == torch.bfloat16):
        layers = getattr(model, '_decoder_bank_layers', None)
        if layers is None:   # (the module tree does not change after construction)
            mods = list(model.decoder.modules()) + list(model.segmentation_head.modules())
            layers = model._decoder_bank_layers = ([m for m in mods if _bankable(m)], [m for m in mods if isinstance(m, HipBatchNorm2d)])
        convs = layers[0]
        if convs:
            bank = getattr(model, '_decoder_bank', None)
            if bank is None or len(bank.weights) != len(convs) or any(a is not m.weight for a, m in zip(bank.weights, convs)) \
                    or not bank.current():
                bank = model._decoder_bank = sis_hip.WeightStdPackBank([m.weight for m in convs], [m.stride[0] for m in convs], 0.0,
                                                                         standardize=False)
            bank.refresh()
            for m, packed, adjoint in zip(convs, bank.packed, bank.adjoint):
                m._banked = (packed, adjoint)
        norms = [m for m in layers[1] if m.training and m.track_running_stats and m.momentum is not None
                 and m.num_batches_tracked is not None and m.num_batches_tracked.is_cuda]
        if norms:
            torch._foreach_add_([m.num_batches_tracked for m in norms], 1)
            for m in norms:
                m._counted_in_bulk = True
    try:
        yield
    finally:
        for m in convs:
            m._banked = None
        for m in norms:
            m._counted_in_bulk = False
