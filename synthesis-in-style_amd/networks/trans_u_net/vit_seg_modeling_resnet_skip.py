"""ResNetV2 stem of the hybrid TransUNet encoder.

Module path and public names of /root/reference/stylegan_code_finder/networks/trans_u_net/
vit_seg_modeling_resnet_skip.py (np2th, StdConv2d :20-27, conv3x3 / conv1x1, PreActBottleneck :40-75,
ResNetV2 :114-162), same parameter names and arithmetic:

* every convolution standardises its weights on the fly, per output filter: (w - mean) / sqrt(var_biased + 1e-5);
* GroupNorm(32 groups, eps 1e-6) after each convolution; the projection shortcut gets GroupNorm(cout groups, cout)
  with the default eps;
* root = 7x7 stride-2 conv -> GN -> ReLU, then a 3x3 stride-2 max-pool WITHOUT padding, three stages of
  bottleneck units (the first unit of stages 2 and 3 has stride 2);
* skip features for the decoder = root output and the outputs of stages 1 and 2, zero-padded on the bottom / right up
  to S/2, S/4, S/8 when the un-padded pooling left them 1-2 pixels short; returned deepest first.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function

import sis_hip
from networks.hip_conv import _BF16_CONV, _Pointwise, conv_bf16, conv_bf16_applicable
from networks.hip_pool import max_pool2d


def np2th(weights, conv=False):
    """numpy (HWIO for convolution kernels) -> torch (OIHW)."""
    return torch.from_numpy(weights.transpose([3, 2, 0, 1]) if conv else weights)


class _WeightStandardize(Function):
    """w_hat = (w - mean) / sqrt(var + eps) per output channel on one HIP launch (csrc/weight_std.hip), written in the
    autocast dtype when autocast is on so that the convolution needs no separate cast of its weight."""

    @staticmethod
    def forward(ctx, weight, eps, out_dtype):
        w_hat, invstd = sis_hip.weight_std_fwd(weight, eps, out_dtype)
        ctx.save_for_backward(weight, invstd)
        ctx.eps = eps
        return w_hat

    @staticmethod
    def backward(ctx, grad):
        weight, invstd = ctx.saved_tensors
        return sis_hip.weight_std_bwd(grad, weight, invstd, ctx.eps), None, None


_SAMPLE_POINTWISE = os.environ.get('SIS_SAMPLE_POINTWISE_S2', '1') != '0'   # 0: 1x1 stride-2 layers on the strided kernel (A/B runs)


def _sampled_pointwise(m):
    return _SAMPLE_POINTWISE and m.kernel_size == (1, 1) and m.stride == (2, 2) and m.padding == (0, 0)


_STEM_OWN = os.environ.get('SIS_STEM_OWN', '1') != '0'   # 0: the root convolution on the library (A/B runs)


class _StemConv(Function):
    """ResNetV2's root convolution (7x7, stride 2, padding 3, image -> 64 channels) on the matrix cores; the image gets no gradient."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return sis_hip.stem_conv_fwd(x, weight)

    @staticmethod
    def backward(ctx, grad):
        x, weight = ctx.saved_tensors
        return None, sis_hip.stem_conv_wgrad(x, grad, weight.dtype, for_param=weight.data_ptr())


class _BankStandardize(Function):
    """All StdConv2d weights of the trunk at once: ``forward(bank, *weights) -> w_hat per layer`` runs ONE launch that also
    writes every layer's packed images (``sis_hip.WeightStdPackBank``: 52 weight_std + 55 conv_pack launches per step before);
    the backward likewise is one launch for all layers (``sis_weight_std_bwd_multi``)."""

    @staticmethod
    def forward(ctx, bank, *weights):
        bank.refresh()
        ctx.bank = bank
        ctx.save_for_backward(*weights)
        return tuple(t.view_as(t) for t in bank.w_hat)   # fresh views of the persistent buffers

    @staticmethod
    def backward(ctx, *grads):
        sis_hip.flush_deferred()   # the layers' weight gradients were queued (conv_bf16(..., defer_wgrad=True)): complete them first
        return (None,) + tuple(ctx.bank.backward(grads))


_GN_GATE_BITS = os.environ.get('SIS_GN_GATE_BITS', '1') != '0'   # 0: the residual norms' backward gates on the saved fp32 output (A/B runs)
_WS_BANK = os.environ.get('SIS_WS_BANK', '1') != '0'   # 0: every StdConv2d standardises and packs its own weight (A/B runs)
_FUSE_RESIDUAL = os.environ.get('SIS_GN_RES', '1') != '0'
_DUAL_STREAM = os.environ.get('SIS_GN_DUAL', '1') != '0'  # bottlenecks hand (fp32 residual stream, 16-bit copy) to the next one


class _GroupNormAct(Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, groups, eps, relu, out_dtype, dual):
        ctx.has_residual = residual is not None
        # residual form: the backward gates on [y > 0] -- one bit per element written by the forward instead of the fp32 output
        want_gate = ctx.has_residual and relu and _GN_GATE_BITS
        out = sis_hip.group_norm_fwd(x, weight, bias, groups, eps, relu, out_dtype, residual, low_precision_copy=dual, want_gate=want_gate)
        y, mean, rstd = out[:3]
        gate = out[-1] if want_gate else None
        ctx.save_for_backward(x, mean, rstd, weight, bias, y if (ctx.has_residual and gate is None) else None, gate)
        ctx.groups, ctx.relu = groups, relu
        ctx.set_materialize_grads(False)  # an unused output of the pair arrives as None, not as a tensor of zeros
        return (y, out[3]) if dual else y

    @staticmethod
    def backward(ctx, grad, grad_lp=None):
        x, mean, rstd, weight, bias, y, gate = ctx.saved_tensors
        if grad is None:  # only the 16-bit copy was used downstream
            grad, grad_lp = grad_lp, None
        if ctx.has_residual:
            dx, dgamma, dbeta, dres = sis_hip.group_norm_bwd(grad, x, mean, rstd, weight, bias, ctx.groups, ctx.relu, y_mask=y,
                                                             want_residual_grad=True, grad_y_lp=grad_lp, gate=gate)
        else:
            (dx, dgamma, dbeta), dres = sis_hip.group_norm_bwd(grad, x, mean, rstd, weight, bias, ctx.groups, ctx.relu,
                                                               grad_y_lp=grad_lp), None
        return dx, dres, dgamma, dbeta, None, None, None, None, None


class HipGroupNorm(nn.GroupNorm):
    """``nn.GroupNorm`` whose forward can also apply the ReLU that follows it and, under autocast, reads and writes the
    16-bit tensors of the neighbouring convolutions directly (csrc/group_norm.hip: one launch per direction instead of
    cast -> moments -> normalise -> relu -> cast).  ``keep_fp32=True`` returns float32 (the residual sum of a bottleneck
    stays in fp32, as autocast would have it)."""

    def forward(self, x, relu=False, keep_fp32=False, residual=None, dual=False):
        """``residual`` (float32): y = relu?(norm(x) + residual) in one pass -- the tail of a bottleneck.  ``dual`` (with a
        16-bit x and a float32 result): returns ``(y, y rounded to x's dtype)`` -- the fp32 residual stream plus the tensor
        the next convolutions read, so that autocast has nothing left to cast; the two gradients are summed in the backward
        kernel's loads."""
        if x.is_cuda and self.affine and x.dtype in (torch.float32, torch.float16, torch.bfloat16) and x.dim() >= 3 \
                and (residual is None or (residual.dtype == torch.float32 and residual.shape == x.shape)):
            out_dtype = torch.float32 if (keep_fp32 or residual is not None) else x.dtype
            dual = bool(dual) and out_dtype == torch.float32 and x.dtype != torch.float32
            return _GroupNormAct.apply(x, residual, self.weight, self.bias, self.num_groups, self.eps, relu, out_dtype, dual)
        if x.is_cuda:
            sis_hip.library_call("vit_seg_modeling_resnet_skip.HipGroupNorm")
        y = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y


class StdConv2d(nn.Conv2d):
    EPS = 1e-5

    def standardized_weight(self):
        w = self.weight
        if w.is_cuda and w.dtype == torch.float32:
            out_dtype = torch.get_autocast_dtype('cuda') if torch.is_autocast_enabled() else torch.float32
            if out_dtype in (torch.float32, torch.float16, torch.bfloat16):
                return _WeightStandardize.apply(w, self.EPS, out_dtype)
        var, mean = torch.var_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
        return (w - mean) / torch.sqrt(var + self.EPS)

    _banked = None   # (w_hat, packed, adjoint) of this forward when the trunk's weight bank produced them

    def forward(self, x):
        if self._banked is not None:
            w, packed, adjoint = self._banked
            xb = x if x.dtype == torch.bfloat16 else x.bfloat16()
            stride = self.stride
            if _sampled_pointwise(self):
                # 1x1 stride 2 (the projection shortcuts of block2 / block3) = the dense 1x1 layer on the even pixels: the strided
                # kernel took 102 us for 8.6 GF at 127 x 127, the sampling copy + the pointwise kernel take 35; the bank packed the
                # weight for stride 1 (with its adjoint image: no pack launch in the backward), autograd scatters dL/dx back
                xb, stride = xb[:, :, ::2, ::2].contiguous(), (1, 1)
            if conv_bf16_applicable(xb, w, stride, self.padding, self.dilation, self.groups):
                return conv_bf16(xb, w, self.bias, stride[0], prepacked=(packed, adjoint), defer_wgrad=True)
        else:
            w = self.standardized_weight()
        if _BF16_CONV and w.dtype == torch.bfloat16 and self.padding_mode == 'zeros' and x.is_cuda and x.dim() == 4:
            xb = x if x.dtype == torch.bfloat16 else x.bfloat16()  # (the root convolution's fp32 image)
            if conv_bf16_applicable(xb, w, self.stride, self.padding, self.dilation, self.groups):
                return conv_bf16(xb, w, self.bias, self.stride[0])  # MI355X bf16 MFMA kernels, NCHW in and out
        if (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) and self.groups == 1
                and x.is_cuda and x.dim() == 4 and x.is_contiguous() and x.dtype == w.dtype):
            return _Pointwise.apply(x, w, self.bias)  # weight gradient as a batched GEMM on the NCHW tensors
        if (_STEM_OWN and _BF16_CONV and x.is_cuda and w.dtype == torch.bfloat16 and self.bias is None and self.groups == 1
                and self.padding_mode == 'zeros' and self.dilation == (1, 1) and self.stride[0] == self.stride[1]
                and self.padding[0] == self.padding[1] and not x.requires_grad
                and sis_hip.stem_conv_supported(x, w, self.stride[0], self.padding[0])):
            return _StemConv.apply(x, w)   # the 7x7 stride-2 root on the 3-channel image: csrc/stem_conv.hip (round 5; the library before)
        if x.is_cuda:   # (anything else of this kind is a library layer: counted)
            sis_hip.library_call("vit_seg_modeling_resnet_skip.StdConv2d.forward", intended=(self.in_channels <= 4))
        return F.conv2d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)


def conv3x3(cin, cout, stride=1, groups=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=bias, groups=groups)


def conv1x1(cin, cout, stride=1, bias=False):
    return StdConv2d(cin, cout, kernel_size=1, stride=stride, padding=0, bias=bias)


class PreActBottleneck(nn.Module):
    """1x1 -> 3x3 (carries the stride) -> 1x1, GroupNorm after each, ReLU after the first two and after the sum."""

    def __init__(self, cin, cout=None, cmid=None, stride=1):
        super().__init__()
        cout = cout or cin
        cmid = cmid or cout // 4
        plan = ((conv1x1, cin, cmid, 1), (conv3x3, cmid, cmid, stride), (conv1x1, cmid, cout, 1))
        for i, (make, a, b, s) in enumerate(plan, start=1):  # registration order gn_i, conv_i as in the reference
            self.add_module(f'gn{i}', HipGroupNorm(32, b, eps=1e-6))
            self.add_module(f'conv{i}', make(a, b, s) if make is conv3x3 else make(a, b))
        self.relu = nn.ReLU(inplace=True)
        if stride != 1 or cin != cout:
            self.downsample = conv1x1(cin, cout, stride)
            self.gn_proj = HipGroupNorm(cout, cout)

    def forward(self, x):
        """``x``: a tensor, or the pair (fp32 residual stream, its 16-bit copy) the previous bottleneck produced under
        autocast -- the convolutions read the copy, the shortcut stays fp32; returns the same kind of pair when it can."""
        x_res, x_conv = x if isinstance(x, tuple) else (x, x)
        shortcut = self.gn_proj(self.downsample(x_conv), keep_fp32=True) if hasattr(self, 'downsample') else x_res
        y = self.gn1(self.conv1(x_conv), relu=True)   # GroupNorm and the ReLU after it in one kernel
        y = self.gn2(self.conv2(y), relu=True)
        if shortcut.dtype == torch.float32 and _FUSE_RESIDUAL:  # relu(shortcut + gn3(conv3(y))) in one kernel, fp32 residual stream
            return self.gn3(self.conv3(y), relu=True, residual=shortcut.contiguous(), dual=_DUAL_STREAM)
        return self.relu(shortcut + self.gn3(self.conv3(y), keep_fp32=True))

    def load_from(self, weights, n_block, n_unit):
        from .npz_import import load_bottleneck
        load_bottleneck(self, weights, n_block, n_unit)


class _Root(nn.Sequential):
    """conv -> gn -> relu with the reference's child names; GroupNorm and ReLU run as one kernel."""

    def forward(self, x):
        return self.gn(self.conv(x), relu=True)


class ResNetV2(nn.Module):
    def __init__(self, block_units, width_factor):
        super().__init__()
        width = self.width = int(64 * width_factor)
        self.root = _Root(OrderedDict(conv=StdConv2d(3, width, kernel_size=7, stride=2, bias=False, padding=3),
                                      gn=HipGroupNorm(32, width, eps=1e-6), relu=nn.ReLU(inplace=True)))
        stages, cin = OrderedDict(), width
        for si, n_units in enumerate(block_units):
            cout, cmid = width * 4 * 2 ** si, width * 2 ** si
            units = OrderedDict()
            for u in range(1, n_units + 1):
                units[f'unit{u:d}'] = PreActBottleneck(cin=cin if u == 1 else cout, cout=cout, cmid=cmid,
                                                       stride=2 if (u == 1 and si > 0) else 1)
            stages[f'block{si + 1}'] = nn.Sequential(units)
            cin = cout
        self.body = nn.Sequential(stages)

    _bank = None
    _bank_list = None

    def _bank_layers(self):
        """The StdConv2d layers whose weights the bank can serve: float32 HIP weights under bf16 autocast with a packing plan
        (all 1x1 / 3x3 layers of the trunk; the 7x7 root keeps its own launch)."""
        return [m for m in self.modules() if isinstance(m, StdConv2d) and m.weight.is_cuda and m.groups == 1 and m.bias is None
                and m.padding_mode == 'zeros' and m.dilation == (1, 1) and m.stride[0] == m.stride[1]
                and m.kernel_size[0] == m.kernel_size[1] and m.padding == (m.kernel_size[0] // 2, m.kernel_size[0] // 2)
                and sis_hip.WeightStdPackBank.supported(m.weight, m.stride[0])]

    def _standardize_all(self):
        if not (_WS_BANK and _BF16_CONV and torch.is_autocast_enabled() and torch.get_autocast_dtype('cuda') == torch.bfloat16):
            return []
        layers = self._bank_list
        if layers is None:
            layers = self._bank_list = self._bank_layers()   # (the module tree does not change after construction)
        if not layers:
            return []
        bank = self._bank
        if bank is None or len(bank.weights) != len(layers) or any(a is not m.weight for a, m in zip(bank.weights, layers)) \
                or not bank.current():
            bank = self._bank = sis_hip.WeightStdPackBank([m.weight for m in layers], [1 if _sampled_pointwise(m) else m.stride[0] for m in layers],
                                                          StdConv2d.EPS)
        w_hats = _BankStandardize.apply(bank, *bank.weights)
        for m, w_hat, packed, adjoint in zip(layers, w_hats, bank.packed, bank.adjoint):
            m._banked = (w_hat, packed, adjoint)
        return layers

    def forward(self, x):
        banked = self._standardize_all() if x.is_cuda else []
        try:
            return self._forward(x)
        finally:
            for m in banked:
                m._banked = None

    def _forward(self, x):
        in_size = x.size(2)
        x = self.root(x)
        skips = [x]
        x = max_pool2d(x, kernel_size=3, stride=2, padding=0)
        for i, stage in enumerate(self.body):
            x = stage(x)  # possibly (fp32, 16-bit) pairs between the units; everything outside the trunk reads the 16-bit copy
            feat = x[1] if isinstance(x, tuple) else x
            if i == len(self.body) - 1:
                x = feat
                break
            want = int(in_size / 4 / (i + 1))
            short = want - feat.size(2)
            assert 0 <= short < 3, f"x {feat.size()} should {want}"
            skips.append(F.pad(feat, (0, short, 0, short)) if short else feat)
        return x, skips[::-1]
