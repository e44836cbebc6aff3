"""``fused_leaky_relu`` / ``FusedLeakyReLU`` backed by the MI355X kernel ``sis_fused_bias_act``.

Public surface = the reference module networks/stylegan2/op/fused_act.py (same four names, same
call signatures, same state_dict key ``bias``):

  fused_leaky_relu(input, bias, negative_slope=0.2, scale=2**0.5)   fused_act.py:85-86
  FusedLeakyReLU(channel, negative_slope, scale)                    fused_act.py:73-82
  FusedLeakyReLUFunction / FusedLeakyReLUFunctionBackward           fused_act.py:51-70 / :19-48

Math (fused_bias_act_kernel.cu:25-47):  y = lrelu_a(x + b_c) * g.   dL/dx = dL/dy * g * (1 or a),
gated on the sign of the saved OUTPUT y (sign(y) == sign(x + b) because g > 0), dL/db = dL/dx summed
over every axis but the channel axis 1.  The gradient map is linear in dL/dy with the same gate, so
the second-order rule is the same kernel again (mode act=3, grad=1).

Where the reference JIT-compiles a CUDA extension at import (fused_act.py:9-16) this module calls
the ahead-of-time built C-ABI library through ``sis_hip``; CPU tensors raise ``RuntimeError``.
"""
import torch
from torch import nn
from torch.autograd import Function

import sis_hip

_ACT_LRELU = 3


def _gated(t, saved_out, bias, slope, gain):
    """(t + bias) passed through the leaky-ReLU *gradient* gate of ``saved_out``, times gain."""
    absent = t.new_empty(0)
    return sis_hip.fused_bias_act(t, absent if bias is None else bias, saved_out, _ACT_LRELU, 1, slope, gain)


def _reduce_to_channel(t):
    axes = [a for a in range(t.ndim) if a != 1]
    return t.sum(axes)


class FusedLeakyReLUFunctionBackward(Function):
    """(grad_output, out) -> (grad_input, grad_bias); differentiable once more."""

    @staticmethod
    def forward(ctx, grad_output, out, negative_slope, scale):
        ctx.slope, ctx.gain = negative_slope, scale
        ctx.save_for_backward(out)
        grad_input = _gated(grad_output, out, None, negative_slope, scale)
        return grad_input, _reduce_to_channel(grad_input).detach()

    @staticmethod
    def backward(ctx, gradgrad_input, gradgrad_bias):
        (out,) = ctx.saved_tensors
        # d(grad_input)/d(grad_output) is the same diagonal gate; the bias cotangent broadcasts back in.
        return _gated(gradgrad_input, out, gradgrad_bias, ctx.slope, ctx.gain), None, None, None


class FusedLeakyReLUFunction(Function):
    @staticmethod
    def forward(ctx, input, bias, negative_slope, scale):
        out = sis_hip.fused_bias_act(input, bias, input.new_empty(0), _ACT_LRELU, 0, negative_slope, scale)
        ctx.slope, ctx.gain = negative_slope, scale
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        (out,) = ctx.saved_tensors
        grad_input, grad_bias = FusedLeakyReLUFunctionBackward.apply(grad_output, out, ctx.slope, ctx.gain)
        return grad_input, grad_bias, None, None


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return FusedLeakyReLUFunction.apply(input, bias, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    """Owns the per-channel bias (checkpoint key ``<prefix>.bias``)."""

    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.negative_slope = negative_slope
        self.scale = scale
        self.bias = nn.Parameter(torch.zeros(channel))

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)

    def extra_repr(self):
        return f"{self.bias.numel()}, negative_slope={self.negative_slope}, scale={self.scale:.4f}"
