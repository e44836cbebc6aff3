"""ORACLE (test infrastructure, not product): torch-CPU restatement of the
reference's two native ops.

* ``upfirdn2d``      follows the pure-PyTorch statement ``upfirdn2d_native`` in
  /root/reference/stylegan_code_finder/networks/stylegan2/op/upfirdn2d.py:152-186
  (zero-insert upsample -> pad / crop -> correlation with the flipped taps ->
  keep every ``down``-th sample), with the NCHW wrapper of ``upfirdn2d.py:144-149,98,121``.
* ``fused_bias_act`` follows the element formula of
  .../op/fused_bias_act_kernel.cu:25-47 (modes ``act*10+grad``) and the launcher's
  bias indexing ``(i / step_b) % size_b`` (``:62-71``).
* ``fused_leaky_relu`` / ``FusedLeakyReLU`` follow ``fused_act.py:51-86``
  (forward act=3 grad=0; backward act=3 grad=1 gated on the saved *output*,
  grad-bias = sum over all dims but 1, ``fused_act.py:28-37``).

Everything here is differentiable through stock autograd, which is what the
tests use as the gradient oracle for the HIP backward kernels.
"""

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F


def upfirdn2d_nhwc(x, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """Op-level layout ``[major, in_h, in_w, minor]`` (upfirdn2d.cpp:12-23)."""
    major, in_h, in_w, minor = x.shape
    kh, kw = kernel.shape
    # zero-insert upsample: sample i lands on i*up, the up-1 zeros follow it
    t = x.reshape(major, in_h, 1, in_w, 1, minor)
    t = F.pad(t, [0, 0, 0, up_x - 1, 0, 0, 0, up_y - 1])
    t = t.reshape(major, in_h * up_y, in_w * up_x, minor)
    # positive pads add zeros, negative pads crop
    t = F.pad(t, [0, 0, max(pad_x0, 0), max(pad_x1, 0), max(pad_y0, 0), max(pad_y1, 0)])
    t = t[:, max(-pad_y0, 0): t.shape[1] - max(-pad_y1, 0), max(-pad_x0, 0): t.shape[2] - max(-pad_x1, 0), :]
    ph, pw = t.shape[1], t.shape[2]
    t = t.permute(0, 3, 1, 2).reshape(major * minor, 1, ph, pw)
    # true convolution == correlation with the flipped taps
    w = torch.flip(kernel, [0, 1]).reshape(1, 1, kh, kw).to(t.dtype)
    t = F.conv2d(t, w)
    t = t.reshape(major, minor, ph - kh + 1, pw - kw + 1).permute(0, 2, 3, 1)
    return t[:, ::down_y, ::down_x, :]


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """NCHW wrapper with the reference's public signature (upfirdn2d.py:144-149)."""
    n, c, h, w = input.shape
    out = upfirdn2d_nhwc(input.reshape(n * c, h, w, 1), kernel, up, up, down, down,
                         pad[0], pad[1], pad[0], pad[1])
    return out.reshape(n, c, out.shape[1], out.shape[2])


def upfirdn2d_out_size(in_size, up, down, pad0, pad1, k):
    """upfirdn2d.py:102-103 == upfirdn2d_kernel.cu:167-168."""
    return (in_size * up + pad0 + pad1 - k) // down + 1


def _as_c_float(v):
    """The pybind entry takes ``float alpha, float scale`` (fused_bias_act.cpp:11-12): Python doubles are
    rounded to binary32 before the kernel widens them to scalar_t, also for float64 tensors."""
    return float(np.float32(v))


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """Element formula of fused_bias_act_kernel.cu:25-47; empty tensors mean absent."""
    alpha, scale = _as_c_float(alpha), _as_c_float(scale)
    y = x
    if bias is not None and bias.numel() > 0:
        shape = [1, -1] + [1] * (x.dim() - 2)
        y = y + bias.reshape(shape)
    mode = act * 10 + grad
    if mode == 30:
        y = torch.where(y > 0, y, y * alpha)
    elif mode == 31:
        y = torch.where(ref > 0, y, y * alpha)
    elif mode in (12, 32):
        y = torch.zeros_like(y)
    # 10, 11 and every unknown mode: linear (the kernel's `default:` label)
    return y * scale


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    shape = [1, -1] + [1] * (input.dim() - 2)
    return F.leaky_relu(input + bias.reshape(shape), _as_c_float(negative_slope)) * _as_c_float(scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)


def make_kernel(k):
    """model.py:23-31."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()
