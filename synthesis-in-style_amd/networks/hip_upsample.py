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
        return F.interpolate(input, size=(oh, ow), mode='bilinear', align_corners=True)

    def extra_repr(self):
        return f"size={self.size}, scale_factor={self.scale_factor}, mode=bilinear, align_corners=True"
