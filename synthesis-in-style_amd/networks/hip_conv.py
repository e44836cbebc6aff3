"""3x3 convolutions of the segmentation networks on the hand-written Winograd MFMA kernel.

``HipConv2d`` is an ``nn.Conv2d`` (same constructor, parameters and state_dict keys) whose forward takes the MI355X
kernel when the layer is a plain 3x3, stride 1, padding = dilation, bias-free convolution on a float32 NCHW tensor
with aligned shapes, and ATen otherwise.  Dilation d is run as the d*d ordinary convolutions of the stride-d
sub-images (EMANet's output-stride-8 trunk: d = 2 at 32^2 -> 16^2 sub-images; d = 8 / 16 leave 4^2 / 2^2
sub-images, which the tile plan rejects: ATen).  Measured on EMANet-50's shapes (B = 16, tools/bench_conv_shapes.py) the
kernel is 1.5-1.9x faster than the library's fp32 path (2048->512 @32^2: 1.70 vs 2.48 ms; 64->128 @128^2: 0.22 vs
0.37 ms).  fp32 1x1 stride-1 convolutions run forward and data gradient on ``sis_conv1x1_f32`` (csrc/conv1x1_f32.hip, exact
fp32 MFMA, the weight read K-major for the data gradient) and their weight gradient on ``sis_conv1x1_wgrad_f32``
(csrc/conv1x1_wgrad_f32.hip: a batched GEMM on the NCHW tensors, deterministic split-K); stride-2 3x3 layers take the dense
kernel and sample its output; only the 3-channel stem and layers the tile plans reject fall through to ATen.

Autograd: the data gradient is the same kernel with adjoint weights (``sis_conv3x3_prepack(adjoint=1)``: channel
axes swapped, taps rotated by 180 degrees); the weight gradient is ``sis_conv3x3_wgrad`` (Winograd-domain GEMM over
the tile axis, csrc/conv_wgrad_wino.hip) where its tile plan applies (channels % 64) and there is enough work, ATen's
``convolution_backward`` otherwise.
"""
import os

import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn import functional as F

import sis_hip


def _space_to_batch(t, d):
    """[B,C,H,W] -> [B*d*d,C,H/d,W/d]: the d*d stride-d sub-images of every sample (a dilated 3x3 convolution is an
    ordinary one on each of them)."""
    if d == 1:
        return t
    b, c, h, w = t.shape
    return t.view(b, c, h // d, d, w // d, d).permute(0, 3, 5, 1, 2, 4).reshape(b * d * d, c, h // d, w // d)


def _batch_to_space(t, d):
    if d == 1:
        return t
    bdd, c, h, w = t.shape
    return t.view(bdd // (d * d), d, d, c, h, w).permute(0, 3, 4, 1, 5, 2).reshape(bdd // (d * d), c, h * d, w * d)


def _conv3x3_fwd(input, weight, d):
    return _batch_to_space(sis_hip.conv3x3(_space_to_batch(input, d), sis_hip.conv3x3_prepack(weight)), d)


def _conv3x3_dgrad(grad_output, weight, d):
    return _batch_to_space(sis_hip.conv3x3(_space_to_batch(grad_output, d), sis_hip.conv3x3_prepack(weight, adjoint=True)), d)


def _conv3x3_wgrad(input, grad_output, weight_shape, d, input_s=None, grad_output_s=None, for_param=None, defer=False):
    """``input_s`` / ``grad_output_s``: the sub-image forms of the two tensors when the caller already has them (a dilated
    layer's forward keeps its permuted input, its backward permutes dL/dy once for both gradients: 4 permute copies per
    layer and step instead of 6)."""
    b, cin, h, w = input.shape
    if sis_hip.conv3x3_wgrad_supported(b * d * d, cin, weight_shape[0], h // d, w // d):
        return sis_hip.conv3x3_wgrad(_space_to_batch(input, d) if input_s is None else input_s,
                                     _space_to_batch(grad_output, d) if grad_output_s is None else grad_output_s, for_param=for_param,
                                     defer=defer)
    # narrow sub-images / channel counts below a 64 x 64 tile / too little work: the library's kernel
    return torch.ops.aten.convolution_backward(grad_output, input, input.new_empty(weight_shape), None, (1, 1), (d, d), (d, d),
                                               False, (0, 0), 1, (False, True, False))[1]


def _may_defer(flag, weight):
    """A layer's weight gradient may wait for ``sis_hip.flush_deferred`` (batched with the other layers of its shape) when the
    owner of the layer said so (``HipConv2d._defer_wgrad``: its weight is used ONCE per forward) and the weight is a parameter
    without a gradient in place -- autograd then takes the tensor as it is; with one in place it would add the unwritten result."""
    return bool(flag) and weight.is_leaf and weight.grad is None and not weight._backward_hooks   # (a tensor hook would read the unwritten gradient)


class _Conv3x3Function(Function):
    @staticmethod
    def forward(ctx, input, weight, dilation, prepacked=None, defer=False):
        """``prepacked``: (forward image, adjoint image) of ``weight`` when a ``sis_hip.WinogradPackBank`` wrote them for this step;
        ``defer``: see ``_may_defer``."""
        ctx.defer = defer
        input_s = _space_to_batch(input, dilation)
        if prepacked is not None:
            u, u_adjoint = prepacked
        elif ctx.needs_input_grad[0]:   # the data gradient of this step convolves with the adjoint image: both from one launch
            u, u_adjoint = sis_hip.conv3x3_prepack_both(weight)
        else:
            u, u_adjoint = sis_hip.conv3x3_prepack(weight), None
        ctx.save_for_backward(input, weight, input_s if dilation > 1 and ctx.needs_input_grad[1] else None, u_adjoint)
        ctx.dilation = dilation
        return _batch_to_space(sis_hip.conv3x3(input_s, u), dilation)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight, input_s, u_adjoint = ctx.saved_tensors
        grad_input, grad_weight = _Conv3x3Backward.apply(grad_output.contiguous(), input, weight, ctx.dilation,
                                                         ctx.needs_input_grad[0], ctx.needs_input_grad[1], input_s, u_adjoint,
                                                         _may_defer(ctx.defer, weight) and not torch.is_grad_enabled())
        return grad_input, grad_weight, None, None, None


class _Conv3x3Backward(Function):
    """(dy, x, W) -> (dx, dW), itself differentiable once more (R1 / path-length regularisers of GAN training take
    a gradient of a gradient): with incoming (ggx, ggW),

        d dy = conv(ggx, W) + conv(x, ggW)      d x = dgrad(dy, ggW)      d W = wgrad(ggx, dy)

    all three on the same Winograd kernels as the first-order pass."""

    @staticmethod
    def forward(ctx, grad_output, input, weight, dilation, want_input, want_weight, input_s=None, u_adjoint=None, defer=False):
        ctx.save_for_backward(grad_output, input, weight)
        ctx.dilation = dilation
        grad_output_s = _space_to_batch(grad_output, dilation)  # once, for both gradients
        grad_input = grad_weight = None
        if want_input:
            if u_adjoint is None:
                u_adjoint = sis_hip.conv3x3_prepack(weight, adjoint=True)
            grad_input = _batch_to_space(sis_hip.conv3x3(grad_output_s, u_adjoint), dilation)
        if want_weight:
            # (for_param: under the data-parallel wrap the kernel writes into the parameter's bucket slice, sis_hip.grad_out)
            grad_weight = _conv3x3_wgrad(input, grad_output, weight.shape, dilation, input_s, grad_output_s, for_param=weight.data_ptr(),
                                         defer=defer)
        return grad_input, grad_weight

    @staticmethod
    @once_differentiable
    def backward(ctx, gg_input, gg_weight):
        grad_output, input, weight = ctx.saved_tensors
        d = ctx.dilation
        need_gy, need_x, need_w = ctx.needs_input_grad[:3]
        d_gy = d_x = d_w = None
        if need_gy:
            if gg_input is not None:
                d_gy = _conv3x3_fwd(gg_input.contiguous(), weight, d)
            if gg_weight is not None:
                t = _conv3x3_fwd(input, gg_weight.contiguous(), d)
                d_gy = t if d_gy is None else d_gy + t
        if need_x and gg_weight is not None:
            d_x = _conv3x3_dgrad(grad_output, gg_weight.contiguous(), d)
        if need_w and gg_input is not None:
            d_w = _conv3x3_wgrad(gg_input.contiguous(), grad_output, weight.shape, d)
        return d_gy, d_x, d_w, None, None, None, None, None, None


def gan_winograd_enabled():
    """SIS_GAN_WINOGRAD=0 sends the GAN-training convolutions (Discriminator, Generator under autograd) to the library
    instead of the Winograd kernels (A/B runs of ``bench.py --workload gan``)."""
    return os.environ.get("SIS_GAN_WINOGRAD", "1") != "0"


def conv3x3(input, weight, dilation=1, prepacked=None, defer_wgrad=False):
    """Differentiable stride-1 3x3 convolution with padding = dilation on the Winograd kernel (caller checks
    eligibility with ``sis_hip.conv3x3_supported``)."""
    return _Conv3x3Function.apply(input, weight, dilation, prepacked, defer_wgrad)


_F32_POINTWISE = os.environ.get('SIS_F32_POINTWISE', '1') != '0'  # 0: fp32 1x1 convolutions stay on the library (A/B runs)


class _Pointwise(Function):
    """1x1 stride-1 convolution whose weight gradient is issued as the batched GEMM it is on the NCHW tensors,
    dW = sum_b dy_b x_b^T (``bmm`` + sum over the batch): the library's weight-gradient path goes through NHWC kernels
    with layout transposes around them (EMANet-50, 37 layers: 5.0 -> 3.2 ms per step, tools/bench_conv1x1.py, plus the
    transposes).  Forward and data gradient stay on the library, which already runs them as plain GEMMs."""

    @staticmethod
    def _narrow(input, weight):
        """Output channels to append (zero filters) so that a layer with a handful of outputs -- EMANet's classifier fc2,
        256 -> num_classes -- runs on the MFMA kernels, whose forward wants Cout % 4 == 0 and whose data gradient contracts
        over Cout in chunks of 32, and whose weight gradient takes 64 output channels per tile (the bf16 path does the same for
        TransUNet's segmentation head); 0 when not applicable."""
        cout = weight.shape[0]
        pad = (-cout) % 64
        if not (_F32_POINTWISE and pad and input.is_cuda and input.dtype == torch.float32 and weight.dtype == torch.float32):
            return 0
        return pad if sis_hip.conv1x1_f32_supported(input, weight.new_empty((cout + pad,) + tuple(weight.shape[1:]))) else 0

    @staticmethod
    def forward(ctx, input, weight, bias, defer=False):
        ctx.save_for_backward(input, weight)
        ctx.has_bias = bias is not None
        ctx.defer = defer
        ctx.pad = 0
        if _F32_POINTWISE and sis_hip.conv1x1_f32_supported(input, weight):
            return sis_hip.conv1x1_f32(input, weight, bias)  # fp32 MFMA kernel, csrc/conv1x1_f32.hip
        ctx.pad = _Pointwise._narrow(input, weight)
        if ctx.pad:
            cout = weight.shape[0]
            w = torch.cat([weight, weight.new_zeros((ctx.pad,) + tuple(weight.shape[1:]))], 0)
            bpad = None if bias is None else torch.cat([bias, bias.new_zeros(ctx.pad)], 0)
            return sis_hip.conv1x1_f32(input, w, bpad)[:, :cout].contiguous()
        sis_hip.library_call("hip_conv._Pointwise.forward")
        return F.conv2d(input, weight, bias)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        b, cin, h, w = input.shape
        cout = weight.shape[0]
        grad_output = grad_output.contiguous()
        grad_input = grad_weight = grad_bias = None
        if ctx.pad:   # narrow layer: zero channels appended to dL/dy and zero filters to W; dW / db sliced back
            gy = torch.cat([grad_output, grad_output.new_zeros(b, ctx.pad, h, w)], 1)
            wp = torch.cat([weight, weight.new_zeros((ctx.pad,) + tuple(weight.shape[1:]))], 0)
            if ctx.needs_input_grad[0]:
                grad_input = sis_hip.conv1x1_f32(gy, wp, data_gradient=True)
            if ctx.needs_input_grad[1]:
                if sis_hip.conv1x1_wgrad_f32_supported(gy, input):
                    grad_weight = sis_hip.conv1x1_wgrad_f32(gy, input)[:cout].contiguous()
                else:
                    sis_hip.library_call("hip_conv._Pointwise.wgrad")
                    grad_weight = torch.bmm(grad_output.view(b, cout, h * w), input.view(b, cin, h * w).transpose(1, 2)).sum(0).view(cout, cin, 1, 1)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                grad_bias = grad_output.sum((0, 2, 3))
            return grad_input, grad_weight, grad_bias, None
        if ctx.needs_input_grad[0]:
            if _F32_POINTWISE and sis_hip.conv1x1_f32_supported(grad_output, weight):
                grad_input = sis_hip.conv1x1_f32(grad_output, weight, data_gradient=True)
            else:
                sis_hip.library_call("hip_conv._Pointwise.dgrad")
                grad_input = torch.ops.aten.convolution_backward(grad_output, input, weight, None, (1, 1), (0, 0), (1, 1), False,
                                                                 (0, 0), 1, (True, False, False))[0]
        g = grad_output.view(b, cout, h * w)
        if ctx.needs_input_grad[1]:
            if _F32_POINTWISE and sis_hip.conv1x1_wgrad_f32_supported(grad_output, input):
                grad_weight = sis_hip.conv1x1_wgrad_f32(grad_output, input, for_param=weight.data_ptr(),   # csrc/conv1x1_wgrad_f32.hip
                                                        defer=_may_defer(ctx.defer, weight) and not torch.is_grad_enabled())
            else:
                sis_hip.library_call("hip_conv._Pointwise.wgrad")
                grad_weight = torch.bmm(g, input.view(b, cin, h * w).transpose(1, 2)).sum(0).view(cout, cin, 1, 1)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = g.sum((0, 2))
        return grad_input, grad_weight, grad_bias, None


class _PointwiseWithSkip(Function):
    """A bottleneck's first 1x1 convolution together with the identity shortcut that leaves from the same tensor:
    forward(x, w) -> (conv1x1(x, w), x).  The shortcut output goes into the block's residual add, so the backward receives BOTH
    gradients of x's two uses and the data-gradient kernel adds the shortcut's in its epilogue (``sis_conv1x1_f32_dgrad_add``)
    -- the separate element-wise sum autograd would launch for a tensor with two consumers disappears (13 of EMANet-50's 16
    bottlenecks; reference networks/ema_net/network.py:37-56)."""

    @staticmethod
    def forward(ctx, input, weight, defer=False):
        ctx.save_for_backward(input, weight)
        ctx.set_materialize_grads(False)
        ctx.defer = defer
        return sis_hip.conv1x1_f32(input, weight), input.view_as(input)

    @staticmethod
    def backward(ctx, grad_output, grad_skip):
        input, weight = ctx.saved_tensors
        grad_input = grad_weight = None
        if grad_output is None:
            return grad_skip, None, None
        grad_output = grad_output.contiguous()
        if ctx.needs_input_grad[0]:
            if grad_skip is not None:
                grad_input = sis_hip.conv1x1_f32_dgrad_add(grad_output, weight, grad_skip.contiguous())
            else:
                grad_input = sis_hip.conv1x1_f32(grad_output, weight, data_gradient=True)
        if ctx.needs_input_grad[1]:
            if sis_hip.conv1x1_wgrad_f32_supported(grad_output, input):
                grad_weight = sis_hip.conv1x1_wgrad_f32(grad_output, input, for_param=weight.data_ptr(),
                                                        defer=_may_defer(ctx.defer, weight) and not torch.is_grad_enabled())
            else:
                sis_hip.library_call("hip_conv._PointwiseWithSkip.wgrad")
                b, cin, h, w = input.shape
                cout = weight.shape[0]
                grad_weight = torch.bmm(grad_output.view(b, cout, h * w), input.view(b, cin, h * w).transpose(1, 2)).sum(0).view(cout, cin, 1, 1)
        return grad_input, grad_weight, None


def pointwise_with_skip(conv, input):
    """(conv(input), input-as-shortcut) through ``_PointwiseWithSkip`` when ``conv`` is a plain fp32 1x1 layer the MFMA kernel
    takes; None otherwise (the caller then runs conv and shortcut separately)."""
    if (_F32_POINTWISE and _FUSE_SKIP_GRAD and isinstance(conv, HipConv2d) and conv._pointwise(input) and conv.bias is None
            and not torch.is_autocast_enabled() and input.dtype == torch.float32 and torch.is_grad_enabled() and input.requires_grad
            and sis_hip.conv1x1_f32_supported(input, conv.weight)):
        return _PointwiseWithSkip.apply(input, conv.weight, conv._defer_wgrad)   # (the support check covers forward and data gradient)
    return None


_FUSE_SKIP_GRAD = os.environ.get('SIS_FUSE_SKIP_GRAD', '1') != '0'


_TAP_SELECT = {}  # device -> [3, 4] 0/1 matrix: row k selects the (u, u') pairs with u' - u + 1 == k


def _tap_select(device):
    sel = _TAP_SELECT.get(device)
    if sel is None:
        sel = torch.zeros(3, 4, device=device)
        for u in (0, 1):
            for u2 in (0, 1):
                sel[u2 - u + 1, 2 * u + u2] = 1.0
        _TAP_SELECT[device] = sel
    return sel


_HALF_DIL_OWN = os.environ.get('SIS_HALF_DIL_OWN', '1') != '0'   # 0: the half-image-dilation layer as library GEMMs (A/B runs)


class _HalfDilationTaps(Function):
    @staticmethod
    def forward(ctx, weight):
        ctx.shape = (weight.shape[0], weight.shape[1])
        ctx.key = weight.data_ptr()   # the parameter's gradient-arena slice (sis_hip.grad_out)
        return sis_hip.half_dilation_taps(weight)

    @staticmethod
    def backward(ctx, grad_taps):
        return sis_hip.half_dilation_taps_bwd(grad_taps.contiguous(), *ctx.shape, for_param=ctx.key)


def conv3x3_half_image_dilation(input, weight):
    """3x3 convolution whose dilation is half the image side (padding = dilation): every output pixel (u*d + p,
    v*d + q) only sees the 2 x 2 pixels {(u'*d + p, v'*d + q)} -- EMANet's last bottleneck (dilation 16 on 32 x 32).
    That is one dense [4 Cin] -> [4 Cout] linear map per (p, q), a single GEMM with 4/9 of the multiplies a 9-tap
    convolution spends (5 of its taps fall into the zero padding); autograd differentiates the composite.

    The [4 Cout, 4 Cin] matrix holds tap (u' - u + 1, v' - v + 1) of the kernel in block ((u, v), (u', v')); it is gathered
    with two multiplications by a 0/1 selection matrix (exact: every sum has one non-zero term), whose autograd backward
    is the matching scatter-add -- 3 launches per direction where slicing and stacking the 16 blocks took 47 (16 zero
    fills and 15 accumulations of the whole weight gradient among them: 0.5 ms of EMANet-50's step)."""
    b, cin, h, w = input.shape
    d = h // 2
    cout = weight.shape[0]
    if _HALF_DIL_OWN and input.is_cuda and input.dtype == torch.float32 and weight.dtype == torch.float32 and not torch.is_autocast_enabled():
        # own kernels end to end: the [4 Cout, 4 Cin] matrix by a gather kernel (csrc/dilation_taps.hip), the product as a
        # pointwise convolution of the [b, (u', v', ci), d x d] arrangement of the input on the fp32 MFMA kernels
        # (csrc/conv1x1_f32.hip: forward, data gradient and weight gradient), no library GEMM left in the layer
        xs = input.view(b, cin, 2, d, 2, d).permute(0, 2, 4, 1, 3, 5).reshape(b, 4 * cin, d, d)
        taps = _HalfDilationTaps.apply(weight)
        if sis_hip.conv1x1_f32_supported(xs, taps.view(4 * cout, 4 * cin, 1, 1)):
            y = _Pointwise.apply(xs, taps.view(4 * cout, 4 * cin, 1, 1), None)                 # [b, (u, v, co), d, d]
            return y.view(b, 2, 2, cout, d, d).permute(0, 3, 1, 4, 2, 5).reshape(b, cout, h, w)
    if input.is_cuda:
        sis_hip.library_call("hip_conv.conv3x3_half_image_dilation (library GEMMs)")
    x = input.view(b, cin, 2, d, 2, d).permute(0, 3, 5, 2, 4, 1).reshape(b * d * d, 4 * cin)  # rows (b,p,q), cols (u',v',ci)
    sel = _tap_select(weight.device)
    # (two plain 2-D products with a transpose copy between them: a batched product of 262 144 3 x 4 matrices costs the
    # library 2 ms per direction)
    t = weight.reshape(cout * cin * 3, 3) @ sel                                     # [(co,ci,ky), (v,v')]
    t = t.view(cout * cin, 3, 4).transpose(1, 2).reshape(cout * cin * 4, 3) @ sel   # [(co,ci,v,v'), (u,u')]
    taps = t.view(cout, cin, 2, 2, 2, 2).permute(4, 2, 0, 5, 3, 1).reshape(4 * cout, 4 * cin)  # rows (u,v,co), cols (u',v',ci)
    y = x @ taps.t()
    return y.view(b, d, d, 2, 2, cout).permute(0, 5, 3, 1, 4, 2).reshape(b, cout, h, w)


def conv_bf16_applicable(input, weight, stride, padding, dilation, groups):
    """The bf16 matrix-core kernels (csrc/conv_bf16.hip) take this layer: bf16 NCHW input on a HIP device (what the
    16-bit norm kernels hand over under autocast), square 1x1 / 3x3 kernel with padding k // 2, stride 1 or 2."""
    if not (input.is_cuda and input.dim() == 4 and input.dtype == torch.bfloat16 and weight.dim() == 4 and groups == 1
            and weight.dtype in (torch.float32, torch.bfloat16) and weight.shape[1] == input.shape[1]):
        return False
    k = weight.shape[2]
    if weight.shape[3] != k or tuple(stride) not in ((1, 1), (2, 2)) or tuple(padding) != (k // 2, k // 2) or tuple(dilation) != (1, 1):
        return False
    return sis_hip.conv_bf16_supported(weight.shape[1], weight.shape[0], input.shape[2], input.shape[3], k, stride[0])


class _ConvBf16Function(Function):
    """bf16 convolution on the hand-written MFMA kernels, NCHW in and out (no layout transposes, no weight cast: the pack
    kernel reads the fp32 master weight or the bf16 standardised weight directly).

    forward        sis_conv_bf16 on the packed weight
    dL/dx          the same kernel on the adjoint packing; 3x3 stride 2: on dL/dy zero-stuffed to the input's size
    dL/dw          3x3: sis_conv_bf16_wgrad where its tile plan applies (stride 2: on the zero-stuffed dL/dy), the library
                   otherwise; 1x1 stride 1: sis_conv1x1_bf16_wgrad (pixels are the contiguous reduction axis of both
                   NCHW operands)
    dL/dbias       fp32 sum of dL/dy
    """

    @staticmethod
    def forward(ctx, input, weight, bias, stride, packed=None, adjoint=None, defer_wgrad=False):
        """``packed`` / ``adjoint``: the images of ``weight`` when the caller already has them (the trunk's weight bank writes
        them for all layers in one launch).  ``defer_wgrad``: the weight gradient may be completed as late as
        ``sis_hip.flush_deferred`` (batched with the other layers of its shape) -- only a caller whose consumer of that gradient
        flushes first may say so (the trunk: ``_BankStandardize.backward``)."""
        input = input.contiguous()
        cout, cin, k, _ = weight.shape
        h, w = input.shape[2], input.shape[3]
        if packed is not None:
            pass
        elif stride == 1 and ctx.needs_input_grad[0] and sis_hip.conv_bf16_supported(cout, cin, h, w, k, 1):
            packed, adjoint = sis_hip.conv_bf16_pack_both(weight, h, w)  # one launch packs for forward AND data gradient
        else:
            packed = sis_hip.conv_bf16_pack(weight, h, w, stride)
        ctx.save_for_backward(input, weight, adjoint)
        ctx.stride, ctx.has_bias = stride, bias is not None
        ctx.defer_wgrad = bool(defer_wgrad)
        return sis_hip.conv_bf16(input, packed, cout, k, stride, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, weight, adjoint = ctx.saved_tensors
        b, cin, h, w = input.shape
        cout, _, k, _ = weight.shape
        s = ctx.stride
        gy = grad_output.contiguous()
        if gy.dtype != torch.bfloat16:
            gy = gy.bfloat16()
        grad_input = grad_weight = grad_bias = None
        lib_weight = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            # on dL/dy as it arrived: the zero-filled (stride 2) and zero-padded (narrow layer) forms below add nothing to the sums
            # and are up to 5x the bytes (the segmentation head: 3 channels padded to 16)
            grad_bias = gy.sum((0, 2, 3), dtype=torch.float32)
        if s == 2 and k == 3 and _STRIDE2_OWN and sis_hip.conv_bf16_supported(cout, cin, h, w, k, 1):
            # A stride-2 convolution is the stride-1 convolution sampled at the even pixels, so its gradients are the
            # stride-1 gradients of dL/dy written into a zero map of the input's size: 4x the multiplies of a dedicated
            # stride-2 backward, on kernels that run several times the rate of the library's (three layers of the net).
            g_full = gy.new_zeros((b, cout, h, w))
            g_full[:, :, ::2, ::2] = gy
            gy, s = g_full, 1
        pad_out = 0
        if s == 1 and k == 3 and cout % 16 and not sis_hip.conv_bf16_supported(cout, cin, h, w, k, 1) \
                and sis_hip.conv_bf16_supported(-(-cout // 16) * 16, cin, h, w, k, 1):
            # a layer with a handful of output channels (the segmentation head: Cout = classes): its gradients contract over
            # Cout, which the kernels take in chunks of 16 -- zero channels are appended to dL/dy and zero filters to W
            pad_out = -(-cout // 16) * 16 - cout
            gy = F.pad(gy, (0, 0, 0, 0, 0, pad_out))
            weight = F.pad(weight, (0, 0, 0, 0, 0, 0, 0, pad_out))
            cout, adjoint = cout + pad_out, None
        scatter = None
        if s == 2 and k == 1 and _STRIDE2_OWN and sis_hip.conv_bf16_supported(cout, cin, gy.shape[2], gy.shape[3], 1, 1):
            # 1x1 stride 2 (the bottlenecks' projection shortcuts) = the dense 1x1 layer on the even pixels of its input: both
            # gradients on the sampled tensors, dL/dx scattered back into a zero map.
            scatter = (h, w)
            input = input[:, :, ::2, ::2].contiguous()
            h, w, s, adjoint = input.shape[2], input.shape[3], 1, None
        if ctx.needs_input_grad[0]:
            if s == 1 and sis_hip.conv_bf16_supported(cout, cin, h, w, k, 1):
                if adjoint is None:
                    adjoint = sis_hip.conv_bf16_pack(weight, h, w, 1, adjoint=True)
                grad_input = sis_hip.conv_bf16(gy, adjoint, cin, k, 1)
            else:
                sis_hip.library_call("hip_conv._ConvBf16Function.dgrad", intended=(cin <= 4))
                lib_weight = weight if weight.dtype == torch.bfloat16 else weight.bfloat16()
                grad_input = torch.ops.aten.convolution_backward(gy, input, lib_weight, None, (s, s), (k // 2, k // 2), (1, 1), False,
                                                                 (0, 0), 1, (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            key = None if pad_out else weight.data_ptr()   # the parameter's arena slice (sis_hip.grad_out); a padded weight is a copy
            # (deferred: same-shape layers of the trunk in one launch at the end of its backward; plain stride-1 layers only --
            # the zero-stuffed / sampled / padded operands above are temporaries of this call)
            later = ctx.defer_wgrad and ctx.stride == 1 and not pad_out and scatter is None and grad_weight is None
            if k == 3 and s == 1 and sis_hip.conv_bf16_wgrad_supported(b, cin, cout, h, w):
                grad_weight = sis_hip.conv_bf16_wgrad(input, gy, weight.dtype, for_param=key, defer=later)
            elif k == 1 and s == 1 and _PW_WGRAD_OWN and sis_hip.conv1x1_bf16_wgrad_supported(b, cin, cout, h * w):
                grad_weight = sis_hip.conv1x1_bf16_wgrad(input, gy, weight.dtype, for_param=key, defer=later)   # csrc/conv_bf16_wgrad.hip, any plane size
            elif k == 1 and s == 1:
                sis_hip.library_call("hip_conv._ConvBf16Function.wgrad_1x1")
                grad_weight = torch.bmm(gy.view(b, cout, h * w), input.view(b, cin, h * w).transpose(1, 2)).sum(0, dtype=torch.float32)
                grad_weight = grad_weight.view(cout, cin, 1, 1)
            else:
                sis_hip.library_call("hip_conv._ConvBf16Function.wgrad", intended=(cin <= 4))
                if lib_weight is None:
                    lib_weight = weight if weight.dtype == torch.bfloat16 else weight.bfloat16()
                grad_weight = torch.ops.aten.convolution_backward(gy, input, lib_weight, None, (s, s), (k // 2, k // 2), (1, 1), False,
                                                                  (0, 0), 1, (False, True, False))[1]
            if grad_weight.dtype != weight.dtype:
                grad_weight = grad_weight.to(weight.dtype)
        if pad_out:
            grad_weight = None if grad_weight is None else grad_weight[:cout - pad_out]
        if scatter is not None and grad_input is not None:
            full = grad_input.new_zeros((b, cin) + scatter)
            full[:, :, ::2, ::2] = grad_input
            grad_input = full
        return grad_input, grad_weight, grad_bias, None, None, None, None


def conv_bf16(input, weight, bias=None, stride=1, prepacked=None, defer_wgrad=False):
    """Differentiable bf16 convolution (padding k // 2) on the matrix-core kernels; the caller checks
    ``conv_bf16_applicable``.  ``prepacked``: (forward image, adjoint image or None) of ``weight``; ``defer_wgrad``: see
    ``_ConvBf16Function.forward``."""
    if prepacked is not None:
        return _ConvBf16Function.apply(input, weight, bias, stride, prepacked[0], prepacked[1], defer_wgrad)
    return _ConvBf16Function.apply(input, weight, bias, stride, None, None, defer_wgrad)


_BF16_CONV = os.environ.get('SIS_BF16_CONV', '1') != '0'  # 0: bf16 convolutions stay on the library (A/B runs)
_PW_WGRAD_OWN = os.environ.get('SIS_PW_WGRAD_OWN', '1') != '0'  # 0: the bf16 1x1 weight gradient as a library bmm + sum over the batch
_STRIDE2_OWN = os.environ.get('SIS_STRIDE2_OWN', '1') != '0'  # 0: stride-2 layers (their backward under bf16) stay on the library


class HipConv2d(nn.Conv2d):
    _banked = None   # (forward image, adjoint image or None) of the weight when a pack bank wrote them for this forward
    _wino_banked = None   # the same for the fp32 Winograd path (sis_hip.WinogradPackBank, networks/ema_net)
    _takes_wino = False   # a forward of this layer went through conv3x3 (the bank packs these layers only)
    _defer_wgrad = False  # the owning network uses this layer's weight once per forward: its fp32 weight gradient may be queued and
                          # batched with the other layers of its shape (sis_hip.flush_deferred; networks/ema_net sets it)

    def _bf16(self, input):
        """Under bf16 autocast: the input is already bf16 (norm kernels write it) or is cast here, as autocast would."""
        if not (_BF16_CONV and input.is_cuda and self.padding_mode == 'zeros' and torch.is_autocast_enabled()
                and torch.get_autocast_dtype('cuda') == torch.bfloat16 and input.dim() == 4):
            return None
        x = input if input.dtype == torch.bfloat16 else input.bfloat16()
        return x if conv_bf16_applicable(x, self.weight, self.stride, self.padding, self.dilation, self.groups) else None

    def _eligible(self, input):
        return (self.kernel_size == (3, 3) and self.stride == (1, 1) and self.dilation[0] == self.dilation[1]
                and self.padding == self.dilation and self.groups == 1 and self.bias is None
                and self.padding_mode == 'zeros' and not torch.is_autocast_enabled()
                and input.is_contiguous() and sis_hip.conv3x3_supported(input, self.weight, self.dilation[0]))

    def _half_image_dilation(self, input):
        d = self.dilation[0]
        return (self.kernel_size == (3, 3) and self.stride == (1, 1) and self.dilation == (d, d) and self.padding == (d, d)
                and self.groups == 1 and self.bias is None and self.padding_mode == 'zeros' and input.dim() == 4
                and input.shape[2] == 2 * d and input.shape[3] == 2 * d and input.is_cuda and input.is_contiguous())

    def _pointwise(self, input):
        # (same dtype on both sides: under autocast with the bf16 kernels switched off the layer is the plain module, whose
        # backward autograd derives -- _Pointwise.backward runs outside the autocast region and would see bf16 x fp32)
        return (self.kernel_size == (1, 1) and self.stride == (1, 1) and self.padding == (0, 0) and self.groups == 1
                and input.is_cuda and input.dim() == 4 and input.is_contiguous() and input.dtype == self.weight.dtype
                and not torch.is_autocast_enabled())

    def _stride2(self, input):
        """fp32 stride-2 layers (EMANet: layer2's 3x3 and its 1x1 shortcut) on the stride-1 kernels: a strided convolution
        is the dense one sampled at the even pixels (3x3, padding 1: 4x the multiplies on a kernel several times faster than
        the library's; autograd's slice backward zero-fills the gradient) or the dense 1x1 on the sampled input."""
        if not (_STRIDE2_OWN and self.stride == (2, 2) and self.dilation == (1, 1) and self.groups == 1 and self.bias is None
                and self.padding_mode == 'zeros' and not torch.is_autocast_enabled() and input.is_cuda and input.dim() == 4
                and input.dtype == torch.float32 and input.is_contiguous()):
            return None
        if self.kernel_size == (3, 3) and self.padding == (1, 1) and sis_hip.conv3x3_supported(input, self.weight, 1):
            self._takes_wino = True
            return conv3x3(input, self.weight, 1, self._wino_banked, self._defer_wgrad)[:, :, ::2, ::2].contiguous()
        if self.kernel_size == (1, 1) and self.padding == (0, 0):
            return _Pointwise.apply(input[:, :, ::2, ::2].contiguous(), self.weight, None, self._defer_wgrad)
        return None

    def forward(self, input):
        x = self._bf16(input)
        if x is not None:
            return conv_bf16(x, self.weight, self.bias, self.stride[0], prepacked=self._banked)
        y = self._stride2(input)
        if y is not None:
            return y
        if self._pointwise(input):
            return _Pointwise.apply(input, self.weight, self.bias, self._defer_wgrad)
        if self._half_image_dilation(input):
            return conv3x3_half_image_dilation(input, self.weight)
        if self._eligible(input):
            self._takes_wino = True
            return conv3x3(input, self.weight, self.dilation[0], self._wino_banked, self._defer_wgrad)
        if input.is_cuda:   # the 3-channel stems are the documented library layers; anything else is a fallback
            sis_hip.library_call("hip_conv.HipConv2d.forward", intended=(self.in_channels <= 4))
        return super().forward(input)
