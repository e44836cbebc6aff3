"""``nn.MaxPool2d`` / ``F.max_pool2d`` on the hand-written HIP kernels (csrc/pool_ops.hip).

The two call sites of the segmentation backbones: EMANet's stem ``nn.MaxPool2d(3, 2, 1)`` (reference
networks/ema_net/network.py:66) and TransUNet's root ``MaxPool2d(3, 2, padding=0)``
(vit_seg_modeling_resnet_skip.py:146).  Bit-equal to ATen in both directions (first maximum of the window, gather
backward); floor mode, dilation 1, f32 / f16 / bf16 on a HIP device -- anything else goes to ``F.max_pool2d``.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn import functional as F

import sis_hip


class _MaxPool2d(Function):
    @staticmethod
    def forward(ctx, input, kernel, stride, padding):
        out, arg = sis_hip.max_pool2d(input, kernel, stride, padding)
        ctx.save_for_backward(arg)
        ctx.geom = (input.shape[2], input.shape[3], kernel, stride, padding)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (arg,) = ctx.saved_tensors
        h, w, kernel, stride, padding = ctx.geom
        return sis_hip.max_pool2d_backward(grad_output, arg, h, w, kernel, stride, padding), None, None, None


def _single(v):
    return v if isinstance(v, int) else (v[0] if v[0] == v[1] else None)


def max_pool2d(input, kernel_size, stride=None, padding=0):
    """Drop-in for ``F.max_pool2d(input, kernel_size, stride, padding)`` (square window, floor mode, dilation 1)."""
    k, s, p = _single(kernel_size), _single(kernel_size if stride is None else stride), _single(padding)
    if (input.is_cuda and input.dim() == 4 and input.dtype in (torch.float32, torch.float16, torch.bfloat16)
            and None not in (k, s, p) and 1 <= k <= 15 and 2 * p <= k and input.shape[2] + 2 * p >= k and input.shape[3] + 2 * p >= k):
        return _MaxPool2d.apply(input, k, s, p)
    sis_hip.library_call("hip_pool.max_pool2d")
    return F.max_pool2d(input, kernel_size, stride, padding)


class HipMaxPool2d(nn.MaxPool2d):
    def forward(self, input):
        if self.dilation in (1, (1, 1)) and not self.ceil_mode and not self.return_indices:
            return max_pool2d(input, self.kernel_size, self.stride, self.padding)
        return super().forward(input)
