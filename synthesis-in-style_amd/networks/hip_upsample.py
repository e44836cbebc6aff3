"""``nn.UpsamplingBilinear2d`` (align_corners=True) on the hand-written HIP kernels (csrc/upsample_ops.hip).

Same constructor and semantics as the torch module the reference uses in the TransUNet decoder
(networks/trans_u_net/vit_seg_modeling.py:299,328); no parameters, so state_dict keys are unchanged.  Forward is one
HBM pass with 16-byte stores, backward a deterministic gather (ATen scatters with float atomics).  f32 / f16 / bf16 on a
HIP device; anything else goes to ``F.interpolate``.
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn import functional as F

import sis_hip


class _UpsampleBilinear(Function):
    @staticmethod
    def forward(ctx, input, out_h, out_w):
        ctx.in_shape = input.shape
        ctx.meta = input.new_empty(0)  # dtype / device carrier
        return sis_hip.upsample_bilinear(input, out_h, out_w)

    @staticmethod
    def backward(ctx, grad_output):
        like = ctx.meta.new_empty(ctx.in_shape)  # shape / dtype / device of the input (never read)
        return sis_hip.upsample_bilinear(like, grad_output.shape[2], grad_output.shape[3],
                                         grad_output=grad_output.to(like.dtype)), None, None


class _UpsampleCat(Function):
    """torch.cat([up2x(x), skip], dim=1) in one upsampling pass + one copy of ``skip``: the upsampled tensor is written straight
    into the leading channels of the result, and the backward reads its share of the gradient from there (the reference's
    DecoderBlock.forward, networks/trans_u_net/vit_seg_modeling.py:300-303; torch.cat copies BOTH parts, and its backward hands
    the upsampling a non-contiguous slice that had to be copied once more)."""

    @staticmethod
    def forward(ctx, x, skip):
        b, c, h, w = x.shape
        out = torch.empty((b, c + skip.shape[1], 2 * h, 2 * w), dtype=x.dtype, device=x.device)
        sis_hip.upsample2x_into(out, x)
        out[:, c:].copy_(skip)
        ctx.channels = c
        return out

    @staticmethod
    def backward(ctx, grad):
        grad = grad.contiguous()
        c = ctx.channels
        gx = sis_hip.upsample2x_grad_from(grad, c) if ctx.needs_input_grad[0] else None
        return gx, (grad[:, c:] if ctx.needs_input_grad[1] else None)


def upsample2x_cat(x, skip):
    """``torch.cat([UpsamplingBilinear2d(2)(x), skip], 1)`` fused (``_UpsampleCat``) when the tensors allow it, None otherwise."""
    if (x.is_cuda and skip.is_cuda and x.dim() == 4 and skip.dim() == 4 and x.dtype == skip.dtype
            and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and skip.shape[0] == x.shape[0]
            and skip.shape[2:] == (2 * x.shape[2], 2 * x.shape[3]) and x.shape[2] >= 2 and x.shape[3] >= 2
            and (x.shape[3] * 2) % 8 == 0 and ((x.shape[1] + skip.shape[1]) * 4 * x.shape[2] * x.shape[3]) % 8 == 0):
        return _UpsampleCat.apply(x, skip)
    return None


class HipUpsamplingBilinear2d(nn.Module):
    def __init__(self, size=None, scale_factor=None):
        super().__init__()
        self.size, self.scale_factor = size, scale_factor

    def _out_size(self, input):
        if self.size is not None:
            return (self.size, self.size) if isinstance(self.size, int) else tuple(self.size)
        sf = self.scale_factor if isinstance(self.scale_factor, (tuple, list)) else (self.scale_factor,) * 2
        return int(input.shape[2] * sf[0]), int(input.shape[3] * sf[1])

    def forward(self, input):
        oh, ow = self._out_size(input)
        if input.is_cuda and input.dim() == 4 and input.dtype in (torch.float32, torch.float16, torch.bfloat16):
            return _UpsampleBilinear.apply(input, oh, ow)
        sis_hip.library_call("hip_upsample.HipUpsamplingBilinear2d")
        return F.interpolate(input, size=(oh, ow), mode='bilinear', align_corners=True)

    def extra_repr(self):
        return f"size={self.size}, scale_factor={self.scale_factor}, mode=bilinear, align_corners=True"
