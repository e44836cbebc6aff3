"""3x3 convolutions of the segmentation networks on the hand-written Winograd MFMA kernel.

``HipConv2d`` is an ``nn.Conv2d`` (same constructor, parameters and state_dict keys) whose forward takes the MI355X
kernel when the layer is a plain 3x3, stride 1, padding 1, dilation 1, bias-free convolution on a float32 NCHW tensor
with aligned shapes, and ATen otherwise.  Measured on EMANet-50's shapes (B = 16, tools/bench_conv_shapes.py) the
kernel is 1.5-1.9x faster than the library's fp32 path (2048->512 @32^2: 1.70 vs 2.48 ms; 64->128 @128^2: 0.22 vs
0.37 ms); 1x1 and dilated convolutions stay on hipBLASLt / MIOpen, which are faster there than this library's
direct kernels.

Autograd: the data gradient is the same kernel with adjoint weights (``sis_conv3x3_prepack(adjoint=1)``: channel
axes swapped, taps rotated by 180 degrees); the weight gradient (a reduction over pixels, a different kernel
shape) is ATen's ``convolution_backward``.
"""
import torch
from torch import nn
from torch.autograd import Function

import sis_hip


class _Conv3x3Function(Function):
    @staticmethod
    def forward(ctx, input, weight):
        ctx.save_for_backward(input, weight)
        return sis_hip.conv3x3(input, sis_hip.conv3x3_prepack(weight))

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        grad_input = grad_weight = None
        grad_output = grad_output.contiguous()
        if ctx.needs_input_grad[0]:
            grad_input = sis_hip.conv3x3(grad_output, sis_hip.conv3x3_prepack(weight, adjoint=True))
        if ctx.needs_input_grad[1]:
            grad_weight = torch.ops.aten.convolution_backward(
                grad_output, input, weight, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, (False, True, False))[1]
        return grad_input, grad_weight


def conv3x3(input, weight):
    """Differentiable stride-1, padding-1 3x3 convolution on the Winograd kernel (caller checks eligibility)."""
    return _Conv3x3Function.apply(input, weight)


class HipConv2d(nn.Conv2d):
    def _eligible(self, input):
        return (self.kernel_size == (3, 3) and self.stride == (1, 1) and self.padding == (1, 1)
                and self.dilation == (1, 1) and self.groups == 1 and self.bias is None
                and self.padding_mode == 'zeros' and not torch.is_autocast_enabled()
                and input.is_contiguous() and sis_hip.conv3x3_supported(input, self.weight))

    def forward(self, input):
        if self._eligible(input):
            return conv3x3(input, self.weight)
        return super().forward(input)
