"""StyleGAN2 generator for MI355X: the reference's module surface on hand-written HIP kernels.

Drop-in for /root/reference/stylegan_code_finder/networks/stylegan2/model.py (generator half):
same class names, constructor signatures, attribute names and -- checked by the test-suite --
the same 135-key ``state_dict`` schema, so ``load_state_dict(torch.load(ckpt)['g_ema'], strict=True)``
(networks/__init__.py:22-29,422) works unchanged, as do the callers in utils/dataset_creation.py.

What is different is how a layer executes.  The reference materialises
per-sample weights ``[B,Cout,Cin,k,k]`` and runs a grouped cuDNN convolution (model.py:237-278),
followed by separate noise, bias/activation and clone kernels.  Here, under ``torch.no_grad()`` on
a HIP device, every StyledConv is one or two launches of ``libsis_hip.so``:

  conv (stride 1)   sis_modconv2d      MFMA fp32; style applied to the input tile on load; demod, noise,
                                       bias, leaky-ReLU*sqrt2 in the epilogue
  conv (upsample)   sis_modconv2d_up   MFMA fp32 transposed conv in 4-phase gather form -> (2H+1)^2
                    sis_blur_noise_act 4x4 FIR + noise + bias + leaky-ReLU*sqrt2
  ToRGB             sis_to_rgb         1x1 modulated conv + bias + polyphase-upsampled skip, one pass
  mapping network   sis_pixel_norm, sis_equal_linear (bias + leaky-ReLU fused), sis_truncate

Weights are shared by the whole batch and prepacked once per checkpoint ([Cin][tap][Cout] plus the
per-(Cout,Cin) squared sums that give the demodulation coefficients from a [B,Cin]x[Cin,Cout]
product).  Requested intermediate activations are the layer outputs themselves (never written
in place afterwards), not extra ``detach().clone()`` copies (model.py:532-549).

With autograd enabled (GAN training / latent projection) the layers run as
differentiable device code: a modulated convolution is channel-scale -> ONE shared-weight convolution
-> channel-scale (``ModulatedConv2d._forward_autograd``; stride-1 3x3 layers on the Winograd kernels,
twice differentiable for the path-length regulariser), with ``upfirdn2d`` / ``fused_leaky_relu`` on
the HIP kernels through their autograd Functions.  The discriminator half of the reference file lives
in ``discriminator.py`` and is re-exported here.  CPU tensors are rejected with ``RuntimeError`` like the reference's extension does.
"""
import math
import os
import random
import struct
import typing

import torch
from torch import nn
from torch.nn import functional as F

import sis_hip
from networks import hip_conv
from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d

_NOISE_ONE_LAUNCH = os.environ.get('SIS_NOISE_ONE_LAUNCH', '1') != '0'   # make_noise(): all maps from one randn launch (device only)
_RGB_STREAMS = {}  # device -> side stream of the ToRGB chain (process-wide: streams are not copyable module state)


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def make_kernel(k):
    """Normalised 2-D FIR taps from a 1-D (outer product) or 2-D list (model.py:23-31)."""
    taps = torch.tensor(k, dtype=torch.float32)
    if taps.ndim == 1:
        taps = torch.outer(taps, taps)
    return taps / taps.sum()


class PixelNorm(nn.Module):
    def forward(self, input):
        if _needs_grad(input) or input.dim() != 2:
            return input * torch.rsqrt(input.pow(2).mean(dim=1, keepdim=True) + 1e-8)
        return sis_hip.pixel_norm(input)


class Upsample(nn.Module):
    """x2 upsampling FIR of the RGB skip branch: taps * factor^2, pad ((p+1)//2 + f - 1, p//2)."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer('kernel', make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, input):
        return upfirdn2d(input, self.kernel, up=self.factor, down=1, pad=self.pad)


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        taps = make_kernel(kernel)
        if upsample_factor > 1:
            taps = taps * (upsample_factor ** 2)
        self.register_buffer('kernel', taps)
        self.pad = pad

    def forward(self, input):
        return upfirdn2d(input, self.kernel, pad=self.pad)


class EqualLinear(nn.Module):
    """Linear layer with equalised learning rate: weights stored / lr_mul, scaled by lr_mul/sqrt(in)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.full((out_dim,), float(bias_init))) if bias else None
        self.activation = activation
        self.lr_mul = lr_mul
        self.scale = lr_mul / math.sqrt(in_dim)

    def forward(self, input):
        if _needs_grad(input, self.weight, self.bias) or input.dim() != 2:
            bias = None if self.bias is None else self.bias * self.lr_mul
            if self.activation:
                return fused_leaky_relu(F.linear(input, self.weight * self.scale), bias)
            return F.linear(input, self.weight * self.scale, bias=bias)
        if input.is_contiguous():
            return sis_hip.equal_linear(input, self.weight, self.bias, self.scale, self.lr_mul, self.activation)
        # a latent[:, i] view of [B, n_latent, D]: rows are a fixed stride apart, read it in place
        if input.stride(1) != 1:
            input = input.contiguous()
            return sis_hip.equal_linear(input, self.weight, self.bias, self.scale, self.lr_mul, self.activation)
        return sis_hip.equal_linear(input, self.weight, self.bias, self.scale, self.lr_mul, self.activation,
                                    row_stride=input.stride(0), batch=input.shape[0])

    def __repr__(self):
        return f'{self.__class__.__name__}({self.weight.shape[1]}, {self.weight.shape[0]})'


class ModulatedConv2d(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        self.demodulate = demodulate
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        if downsample:
            factor = 2
            p = (len(blur_kernel) - factor) + (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2, p // 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self._pack = None
        self._pack_key = None

    def __repr__(self):
        return (f'{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, '
                f'upsample={self.upsample}, downsample={self.downsample})')

    # ---- MI355X path -----------------------------------------------------------------------
    def packed_weights(self):
        """(wpk [Cin, k*k, Cout], wsq [Cout, Cin]); rebuilt when the parameter changes."""
        w = self.weight
        key = (w.data_ptr(), w._version, w.device)
        if self._pack_key != key:
            self._pack = sis_hip.modconv_prepack(w.detach())
            # stride-1 3x3 layers also keep the Winograd F(2x2,3x3) transform of the weights (2.25x fewer MFMA
            # FLOPs); SIS_WINOGRAD=0 selects the direct kernel everywhere (bisecting / A-B runs)
            self._wino = None
            if (self.kernel_size == 3 and not self.upsample and not self.downsample and self.in_channel % 8 == 0
                    and self.out_channel % 4 == 0 and os.environ.get("SIS_WINOGRAD", "1") != "0"):
                self._wino = sis_hip.modconv_prepack_wino(w.detach())
            # up-convolutions keep the 16 transformed planes of the fast-FIR kernel (25 instead of 36 multiplies per 2 x 2
            # input positions, csrc/modconv_upfir.hip); the launch falls back to the 4-phase kernel below 32 x 32
            self._fir = None
            if self.kernel_size == 3 and self.upsample and self.in_channel % 8 == 0 and self.out_channel % 64 == 0:
                self._fir = sis_hip.modconv_prepack_up_fir(w.detach())
            self._pack_key = key
        return self._pack

    def wino_weights(self):
        self.packed_weights()
        return self._wino

    def fir_weights(self):
        self.packed_weights()
        return self._fir

    def hip_supported(self):
        return (not self.downsample) and (self.kernel_size == 3 or (self.kernel_size == 1 and not self.upsample))

    def modulate(self, style):
        """s [B, Cin] and the epilogue factor scale * demod [B, Cout]."""
        wpk, wsq = self.packed_weights()
        s = self.modulation(style)
        return wpk, s, sis_hip.modconv_demod(s, wsq, self.scale, self.demodulate)

    def forward(self, input, style):
        if _needs_grad(input, style, self.weight, self.modulation.weight) or not self.hip_supported():
            return self._forward_autograd(input, style)
        sis_hip.require_device(input, "input")
        wpk, s, dscale = self.modulate(style)
        if self.upsample:
            return self.blur(sis_hip.modconv2d_up(input, wpk, s, dscale))
        return sis_hip.modconv2d(input, wpk, s, dscale, self.kernel_size, wino_u=self.wino_weights())

    # ---- differentiable path (GAN training / latent projection) ---------------------------------
    def _forward_autograd(self, input, style):
        """y[b] = dcoef[b] (.) conv(W, s[b] (.) x[b]): the per-sample weights w[b] = scale * W * s[b] * demod[b] of the
        reference (model.py:237-278) factor into a channel scaling of the input, ONE convolution with the shared
        weights over the whole batch, and a channel scaling of the output with
        dcoef[b,co] = scale * rsqrt(scale^2 * sum_ci s[b,ci]^2 * sum_taps W[co,ci]^2 + 1e-8) -- no [B,Cout,Cin,k,k]
        tensor, no B-group convolution; stride-1 3x3 layers run on the Winograd kernels (first and second order).
        SIS_MODCONV_GROUPED=1 selects the reference's grouped formulation (A/B runs; downsampling layers always)."""
        if self.downsample or os.environ.get("SIS_MODCONV_GROUPED", "0") == "1":
            return self._forward_grouped(input, style)
        b, cin, h, w = input.shape
        k, cout = self.kernel_size, self.out_channel
        s = self.modulation(style)
        weight = self.weight[0]
        if self.demodulate:
            wsq = weight.pow(2).sum((2, 3))
            dcoef = self.scale * torch.rsqrt((self.scale * self.scale) * (s.pow(2) @ wsq.t()) + 1e-8)
        x = input * s.view(b, cin, 1, 1)
        if self.upsample:
            out = F.conv_transpose2d(x, weight.transpose(0, 1), stride=2)
        elif (k == 3 and x.is_cuda and x.is_contiguous() and hip_conv.gan_winograd_enabled()
              and sis_hip.conv3x3_supported(x, weight)):
            out = hip_conv.conv3x3(x, weight)
        else:
            out = F.conv2d(x, weight, padding=self.padding)
        out = out * dcoef.view(b, cout, 1, 1) if self.demodulate else out * self.scale
        return self.blur(out) if self.upsample else out

    def _forward_grouped(self, input, style):
        b, cin, h, w = input.shape
        k, cout = self.kernel_size, self.out_channel
        mod = self.modulation(style).view(b, 1, cin, 1, 1)
        wt = self.scale * self.weight * mod
        if self.demodulate:
            wt = wt * torch.rsqrt(wt.pow(2).sum([2, 3, 4], keepdim=True) + 1e-8)
        x = input.reshape(1, b * cin, h, w)
        if self.upsample:
            out = F.conv_transpose2d(x, wt.transpose(1, 2).reshape(b * cin, cout, k, k), stride=2, groups=b)
            return self.blur(out.view(b, cout, out.shape[2], out.shape[3]))
        if self.downsample:
            x = self.blur(input)
            x = x.reshape(1, b * cin, x.shape[2], x.shape[3])
            out = F.conv2d(x, wt.view(b * cout, cin, k, k), stride=2, groups=b)
        else:
            out = F.conv2d(x, wt.view(b * cout, cin, k, k), padding=self.padding, groups=b)
        return out.view(b, cout, out.shape[2], out.shape[3])


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    @staticmethod
    def fresh(image):
        b, _, h, w = image.shape
        return image.new_empty(b, 1, h, w).normal_()

    def forward(self, image, noise=None):
        if noise is None:
            noise = self.fresh(image)
        return image + self.weight * noise


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, input):
        return self.input.repeat(input.shape[0], 1, 1, 1)


class StyledConv(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)

    def forward(self, input, style, noise=None):
        conv, act = self.conv, self.activate
        fused_ok = (conv.hip_supported() and act.negative_slope == 0.2 and abs(act.scale - 2 ** 0.5) < 1e-12
                    and not _needs_grad(input, style, conv.weight, conv.modulation.weight, self.noise.weight, act.bias))
        if not fused_ok:
            return act(self.noise(conv(input, style), noise=noise))
        sis_hip.require_device(input, "input")
        wpk, s, dscale = conv.modulate(style)
        return self.forward_s(input, wpk, s, dscale, noise)

    def forward_s(self, input, wpk, s, dscale, noise=None, out=None):
        """Fused layer given the already computed style vector ``s`` [B,Cin] and epilogue factors [B,Cout].  ``out`` (stride-1
        layers): result tensor to write into."""
        conv, act = self.conv, self.activate
        b, _, h, w = input.shape
        if conv.upsample:
            taps, pad = conv.blur.kernel, conv.blur.pad
            padded = tuple(taps.shape) == (4, 4) and pad[0] == 1  # row-streaming blur wants 16-byte aligned rows
            t = sis_hip.modconv2d_up(input, wpk, s, dscale, padded_rows=padded, fir_u=conv.fir_weights())
            in_w = 2 * w + 1
            oh = t.shape[2] + pad[0] + pad[1] - taps.shape[0] + 1
            ow = in_w + pad[0] + pad[1] - taps.shape[1] + 1
            if noise is None:
                noise = input.new_empty(b, 1, oh, ow).normal_()
            return sis_hip.blur_noise_act(t, taps, pad, noise, self.noise.weight, act.bias, fuse_act=True, in_w=in_w)
        if noise is None:
            noise = input.new_empty(b, 1, h, w).normal_()
        return sis_hip.modconv2d(input, wpk, s, dscale, conv.kernel_size, noise, self.noise.weight, act.bias,
                                 fuse_act=True, wino_u=conv.wino_weights(), out=out)


class ToRGB(nn.Module):
    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, input, style, skip=None):
        conv = self.conv
        if _needs_grad(input, style, skip, conv.weight, conv.modulation.weight, self.bias):
            out = conv(input, style) + self.bias
            return out if skip is None else out + self.upsample(skip)
        sis_hip.require_device(input, "input")
        return self.forward_s(input, conv.modulation(style), skip)

    def forward_s(self, input, s, skip=None, out=None):
        """``out``: result tensor to write into (a batch slice of the image: Generator.forward's tail)."""
        conv = self.conv
        if skip is None:
            return sis_hip.to_rgb(input, conv.weight, s, self.bias, conv.scale, out=out)
        up = self.upsample
        if up.factor != 2:
            return sis_hip.to_rgb(input, conv.weight, s, self.bias, conv.scale) + up(skip)
        return sis_hip.to_rgb(input, conv.weight, s, self.bias, conv.scale, skip, up.kernel, up.pad, out=out)


def truncate_styles(style, truncation, truncation_latent):
    """truncation_latent + truncation * (style - truncation_latent) (model.py:509-516), one HIP launch in inference."""
    if _needs_grad(style, truncation_latent) or not style.is_cuda or truncation_latent.numel() != style.shape[-1]:
        return truncation_latent + truncation * (style - truncation_latent)
    return sis_hip.truncate(style, truncation_latent, truncation)


def resolve_latents(generator, styles, inject_index, truncation, truncation_latent, input_is_latent, noise,
                    randomize_noise):
    """The argument handling every generator variant of the reference opens ``forward`` with (stylegan2
    model.py:491-531, swagan model.py:214-250): mapping network unless ``input_is_latent``, stored / fresh noise
    selection, truncation towards ``truncation_latent``, and the [B, n_latent, D] latent with optional style mixing
    at ``inject_index``.  Returns (latent, noise list)."""
    if not input_is_latent:
        styles = [generator.style(s) for s in styles]
    if noise is None:
        noise = [None] * generator.num_layers if randomize_noise else [
            getattr(generator.noises, f'noise_{i}') for i in range(generator.num_layers)]
    if truncation < 1:
        styles = [truncate_styles(s, truncation, truncation_latent) for s in styles]
    n_latent = generator.n_latent
    if len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, n_latent, 1) if styles[0].ndim < 3 else styles[0]
    else:
        if inject_index is None:
            inject_index = random.randint(1, n_latent - 1)
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)
    return latent, noise


class Generator(nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size = size
        self.style_dim = style_dim
        self.style = nn.Sequential(PixelNorm(), *[
            EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation='fused_lrelu') for _ in range(n_mlp)])
        self.channels = self.get_channels(channel_multiplier)
        self.input = ConstantInput(self.channels[4])
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2

        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        for layer_idx in range(self.num_layers):
            res = 2 ** ((layer_idx + 5) // 2)
            self.noises.register_buffer(f'noise_{layer_idx}', torch.randn(1, 1, res, res))
        in_channel = self.channels[4]
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel))
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel

    @staticmethod
    def get_channels(channel_multiplier=2) -> typing.Dict[int, int]:
        cm = channel_multiplier
        return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm,
                1024: 16 * cm}

    def make_noise(self) -> typing.List[torch.Tensor]:
        device = self.input.input.device
        sizes = [4] + [2 ** i for i in range(3, self.log_size + 1) for _ in range(2)]
        if device.type == 'cuda' and _NOISE_ONE_LAUNCH:
            # one generator launch for all maps (13 launches of 5-12 us each otherwise, on the critical path of the dataset loop):
            # the maps are consecutive slices of one buffer (every offset a multiple of 16 floats)
            flat = torch.randn(sum(s * s for s in sizes), device=device)
            out, o = [], 0
            for s in sizes:
                out.append(flat[o:o + s * s].view(1, 1, s, s))
                o += s * s
            return out
        return [torch.randn(1, 1, s, s, device=device) for s in sizes]

    def mean_latent(self, n_latent):
        latent_in = torch.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, input):
        return self.style(input)

    # ---- one-launch modulation / demodulation for the whole forward (MI355X fast path) ----------------
    def _layer_sequence(self):
        """(module, latent index) in execution order (model.py:534-552 of the reference)."""
        seq = [(self.conv1, 0), (self.to_rgb1, 1)]
        for r in range(self.log_size - 2):
            i = 1 + 2 * r
            seq += [(self.convs[2 * r], i), (self.convs[2 * r + 1], i + 1), (self.to_rgbs[r], i + 2)]
        return seq

    def _modulation_plan(self, batch, device):
        seq = self._layer_sequence()
        key = (batch, device) + tuple((m.conv.weight.data_ptr(), m.conv.weight._version, m.conv.modulation.weight.data_ptr(),
                                       m.conv.modulation.bias.data_ptr()) for m, _ in seq)
        if getattr(self, '_plan_key', None) == key:
            return self._plan
        mod_rows, dem_rows, s_off, d_off, mblocks, dblocks = [], [], 0, 0, 0, 0
        tile = sis_hip.head_gemm_tile()
        s_slices, d_slices = [], []
        for m, lat in seq:
            conv = m.conv
            cin, cout = conv.in_channel, conv.out_channel
            lin = conv.modulation
            mod_rows.append([lin.weight.data_ptr(), lin.bias.data_ptr(), s_off, lat, cin, mblocks, 0, 0])
            s_slices.append((s_off, cin))
            mblocks += (cin + tile - 1) // tile
            if isinstance(m, StyledConv):
                _, wsq = conv.packed_weights()
                bits = struct.unpack('<i', struct.pack('<f', conv.scale))[0]
                dem_rows.append([wsq.data_ptr(), bits, s_off, d_off, cout, dblocks, cin, int(conv.demodulate)])
                d_slices.append((d_off, cout))
                dblocks += (cout + tile - 1) // tile
                d_off += batch * cout
            else:
                d_slices.append(None)
            s_off += batch * cin
        self._plan = dict(mod=torch.tensor(mod_rows, dtype=torch.int64).to(device),
                          dem=torch.tensor(dem_rows, dtype=torch.int64).to(device), n_mod=len(mod_rows),
                          n_dem=len(dem_rows), mblocks=mblocks, dblocks=dblocks, s_total=s_off, d_total=d_off,
                          s_slices=s_slices, d_slices=d_slices, lin_scale=seq[0][0].conv.modulation.scale)
        self._plan_key = key
        return self._plan

    def _fast_path(self, latent):
        if _needs_grad(latent) or any(p.requires_grad for p in self.parameters()) and torch.is_grad_enabled():
            return False
        if not latent.is_cuda or latent.dtype != torch.float32 or latent.dim() != 3 or latent.shape[-1] > 1024:
            return False
        return all(m.conv.hip_supported() and m.conv.modulation.lr_mul == 1 and m.conv.modulation.activation is None
                   for m, _ in self._layer_sequence())

    def _rgb_stream(self, device):
        """Side stream of the ToRGB chain (one per device, created on first use; SIS_RGB_STREAM=0 disables it)."""
        if os.environ.get("SIS_RGB_STREAM", "1") == "0":
            return None
        if device not in _RGB_STREAMS:
            _RGB_STREAMS[device] = sis_hip.side_stream(device)   # (one that really runs beside the current stream)
        return _RGB_STREAMS[device]

    @staticmethod
    def _tail_parts(x, rgb, skip):
        """In how many batch parts the last resolution's convolution + ToRGB run (``_tail``): SIS_RGB_TAIL_SPLIT when the batch
        divides and holds at least 4 samples per part, else 1.  Default 1 (off): measured on one box at B = 32, two alternating
        rounds, 15.01-15.02 ms unsplit against 15.02-15.04 (2 parts) and 15.06-15.09 ms (4 parts) -- what the hidden ToRGB
        saves, the convolution loses to the extra launch tails and to sharing HBM with it (profiles/r05_syn_tail_split.txt)."""
        parts = int(os.environ.get("SIS_RGB_TAIL_SPLIT", "1"))
        b = x.shape[0]
        if parts < 2 or b % parts or b // parts < 4 or skip is None or rgb.upsample.factor != 2:
            return 1
        return parts

    def _tail(self, x, conv, rgb, s_conv, d_conv, s_rgb, noise, skip, main, side):
        """The last StyledConv and its ToRGB by batch parts.  The final ToRGB has no later convolution to hide behind (196 us of
        a 14.9 ms step at B = 32: an HBM-bound pass over the 1 GB activation with the matrix cores idle), so the batch is cut
        in parts: while the convolution of part k + 1 runs, the side stream converts part k; only the last part's ToRGB is
        exposed.  Results are written into batch slices of ONE activation tensor and ONE image: same values, same layout."""
        parts = self._tail_parts(x, rgb, skip)
        b, _, h, w = x.shape
        n = b // parts
        cout = conv.conv.out_channel
        act = x.new_empty((b, cout, h, w))
        image = x.new_empty((b, 3, h, w))
        wpk = conv.conv.packed_weights()[0]
        per_sample_noise = noise is not None and noise.numel() == b * h * w
        if noise is None:
            noise = x.new_empty(b, 1, h, w).normal_()
            per_sample_noise = True
        for k in range(parts):
            sl = slice(k * n, (k + 1) * n)
            conv.forward_s(x[sl], wpk, s_conv[sl], d_conv[sl], noise[sl] if per_sample_noise else noise, out=act[sl])
            if k + 1 < parts:
                ready = main.record_event()
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    rgb.forward_s(act[sl], s_rgb[sl], skip[sl], out=image[sl])
            else:   # behind the side chain, on the main stream (as rgb_branch's last=True)
                main.wait_event(side.record_event())
                rgb.forward_s(act[sl], s_rgb[sl], skip[sl], out=image[sl])
        for t in (act, image):
            t.record_stream(side)
        skip.record_stream(main)
        return act, image

    def _modulate_all(self, latent):
        """Every layer's s [B,Cin] (and scale*demod [B,Cout] for the styled convs): two launches in total."""
        latent = latent.contiguous()
        b = latent.shape[0]
        plan = self._modulation_plan(b, latent.device)
        s_flat = torch.empty(plan['s_total'], dtype=torch.float32, device=latent.device)
        d_flat = torch.empty(max(plan['d_total'], 1), dtype=torch.float32, device=latent.device)
        sis_hip.modulation_batch(s_flat, latent, plan['mod'], plan['n_mod'], plan['mblocks'], plan['lin_scale'])
        sis_hip.demod_batch(d_flat, s_flat, plan['dem'], plan['n_dem'], plan['dblocks'], b)
        s_list = [s_flat[o:o + b * c].view(b, c) for o, c in plan['s_slices']]
        d_list = [None if sl is None else d_flat[sl[0]:sl[0] + b * sl[1]].view(b, sl[1]) for sl in plan['d_slices']]
        return s_list, d_list

    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, return_intermediate_activations=False):
        latent, noise = resolve_latents(self, styles, inject_index, truncation, truncation_latent, input_is_latent,
                                        noise, randomize_noise)

        acts = {} if return_intermediate_activations else None

        def tap(idx, t):
            if acts is not None:
                acts[idx] = t.detach()

        out = self.input(latent)
        tap(0, out)
        if self._fast_path(latent):
            s, d = self._modulate_all(latent)
            # The ToRGB / skip chain only reads each resolution's activation: it runs on a side stream so that its
            # HBM-bound passes overlap the MFMA-bound convolutions of the next resolution.
            main, side = torch.cuda.current_stream(latent.device), self._rgb_stream(latent.device)

            def rgb_branch(layer, x, style, prev, last=False):
                if side is None:
                    return layer.forward_s(x, style, prev)
                if last:  # nothing left to overlap it with: on the main stream, behind the side chain (no fork / join gap)
                    main.wait_event(side.record_event())
                    if prev is not None:
                        prev.record_stream(main)  # allocated on the side stream, read here
                    return layer.forward_s(x, style, prev)
                ready = main.record_event()
                x.record_stream(side)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    return layer.forward_s(x, style, prev)

            out = self.conv1.forward_s(out, self.conv1.conv.packed_weights()[0], s[0], d[0], noise[0])
            tap(1, out)
            skip = rgb_branch(self.to_rgb1, out, s[1], None)
            for r in range(self.log_size - 2):
                i, j = 1 + 2 * r, 2 + 3 * r
                up, conv, rgb = self.convs[2 * r], self.convs[2 * r + 1], self.to_rgbs[r]
                out = up.forward_s(out, up.conv.packed_weights()[0], s[j], d[j], noise[1 + 2 * r])
                tap(i + 1, out)
                last = r == self.log_size - 3
                if last and side is not None and self._tail_parts(out, rgb, skip) > 1:
                    out, skip = self._tail(out, conv, rgb, s[j + 1], d[j + 1], s[j + 2], noise[2 + 2 * r], skip, main, side)
                    tap(i + 2, out)
                    continue
                out = conv.forward_s(out, conv.conv.packed_weights()[0], s[j + 1], d[j + 1], noise[2 + 2 * r])
                tap(i + 2, out)
                skip = rgb_branch(rgb, out, s[j + 2], skip, last=last)
            if side is not None and self.log_size == 2:  # 4 x 4 generator: to_rgb1 is the only (side-stream) ToRGB
                main.wait_event(side.record_event())
                skip.record_stream(main)
            if return_latents:
                return skip, latent
            return (skip, acts) if return_intermediate_activations else (skip, None)
        out = self.conv1(out, latent[:, 0], noise=noise[0])
        tap(1, out)
        skip = self.to_rgb1(out, latent[:, 1])
        for r in range(self.log_size - 2):
            i = 1 + 2 * r
            out = self.convs[2 * r](out, latent[:, i], noise=noise[1 + 2 * r])
            tap(i + 1, out)
            out = self.convs[2 * r + 1](out, latent[:, i + 1], noise=noise[2 + 2 * r])
            tap(i + 2, out)
            skip = self.to_rgbs[r](out, latent[:, i + 2], skip)
        image = skip
        if return_latents:
            return image, latent
        if return_intermediate_activations:
            return image, acts
        return image, None


from .discriminator import ConvLayer, Discriminator, Downsample, EqualConv2d, ResBlock, ScaledLeakyReLU  # noqa: E402,F401
