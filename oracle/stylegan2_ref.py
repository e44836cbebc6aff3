"""ORACLE (test infrastructure, not product): functional CPU restatement of the
reference StyleGAN2 generator forward pass, over the ``g_ema`` state_dict.

Follows /root/reference/stylegan_code_finder/networks/stylegan2/model.py:
  PixelNorm            :19-20        EqualLinear          :152-162 (ctor scale :149)
  ModulatedConv2d      :237-278      (per-sample weights, grouped conv / grouped
                                      transposed conv stride 2 + Blur pad (1,1))
  NoiseInjection       :287-292      ConstantInput        :301-305
  StyledConv           :336-342      ToRGB                :355-364
  Upsample             :35-52 (pad (2,1), taps * 4)       Blur :77-92
  Generator.forward    :479-561      (latent indexing :534-552, truncation :502-510)

The formulation is deliberately the reference's own (materialise
``[B,Cout,Cin,k,k]`` weights, ``F.conv2d(groups=B)``), NOT the shared-weight
formulation the HIP kernels use, so that agreement between the two is evidence
and not tautology.  It is also the "reference's pure-PyTorch CPU path" that
``bench.py`` times as ``cpu_baseline`` (kind "port").
"""
import math

import numpy as np
import torch
from torch.nn import functional as F

from . import ops_ref

SQRT2 = 2 ** 0.5


def get_channels(channel_multiplier=2):
    """model.py:443-455."""
    return {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
            256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}


def state_dict_schema(size, style_dim=512, n_mlp=8, channel_multiplier=2, blur_kernel=(1, 3, 3, 1)):
    """Ordered (name, shape) list of ``Generator(size, style_dim, n_mlp, cm).state_dict()``
    (135 keys at size 256).  Order = module registration order of model.py:367-441."""
    ch = get_channels(channel_multiplier)
    log_size = int(math.log(size, 2))
    num_layers = (log_size - 2) * 2 + 1
    k = len(blur_kernel)
    out = []
    for i in range(n_mlp):
        out += [(f"style.{i + 1}.weight", (style_dim, style_dim)), (f"style.{i + 1}.bias", (style_dim,))]
    out += [("input.input", (1, ch[4], 4, 4))]

    def styled(prefix, cin, cout, up):
        r = [(f"{prefix}.conv.weight", (1, cout, cin, 3, 3))]
        if up:
            r += [(f"{prefix}.conv.blur.kernel", (k, k))]
        r += [(f"{prefix}.conv.modulation.weight", (cin, style_dim)), (f"{prefix}.conv.modulation.bias", (cin,)),
              (f"{prefix}.noise.weight", (1,)), (f"{prefix}.activate.bias", (cout,))]
        return r

    def torgb(prefix, cin, up):
        r = [(f"{prefix}.bias", (1, 3, 1, 1))]
        if up:
            r += [(f"{prefix}.upsample.kernel", (k, k))]
        r += [(f"{prefix}.conv.weight", (1, 3, cin, 1, 1)), (f"{prefix}.conv.modulation.weight", (cin, style_dim)),
              (f"{prefix}.conv.modulation.bias", (cin,))]
        return r

    out += styled("conv1", ch[4], ch[4], False)
    out += torgb("to_rgb1", ch[4], False)
    cin = ch[4]
    convs, rgbs = [], []
    for i in range(3, log_size + 1):
        cout = ch[2 ** i]
        convs += styled(f"convs.{2 * (i - 3)}", cin, cout, True)
        convs += styled(f"convs.{2 * (i - 3) + 1}", cout, cout, False)
        rgbs += torgb(f"to_rgbs.{i - 3}", cout, True)
        cin = cout
    out += convs + rgbs
    for li in range(num_layers):
        res = (li + 5) // 2
        out += [(f"noises.noise_{li}", (1, 1, 2 ** res, 2 ** res))]
    return out


def seeded_state_dict(size, style_dim=512, n_mlp=8, channel_multiplier=2, seed=0, noise_weight_std=0.1,
                      lr_mlp=0.01, dtype=torch.float32):
    """Platform-independent synthetic checkpoint: every tensor drawn from
    ``numpy.random.RandomState(seed)`` (frozen legacy stream) in schema order, with the
    reference's init *scales* (model.py:139 weights ~N(0,1)/lr_mul, :142 biases 0 but
    modulation bias 1, :285 noise weight 0 -> here N(0, noise_weight_std^2) so that the noise
    path is exercised; activation biases N(0, 0.1^2) for the same reason).
    Blur / upsample taps are the fixed make_kernel([1,3,3,1]) buffers (model.py:82-85, 39-40)."""
    rng = np.random.RandomState(seed)
    k2d = ops_ref.make_kernel([1, 3, 3, 1])
    sd = {}
    for name, shape in state_dict_schema(size, style_dim, n_mlp, channel_multiplier):
        if name.endswith("blur.kernel") or name.endswith("upsample.kernel"):
            sd[name] = (k2d * 4).to(dtype)
            continue
        t = torch.from_numpy(rng.standard_normal(shape)).to(dtype)
        if name.startswith("style.") and name.endswith(".weight"):
            t = t / lr_mlp
        elif name.startswith("style.") and name.endswith(".bias"):
            t = t * 0.1 / lr_mlp  # effective bias = bias * lr_mul ~ N(0, 0.1^2)
        elif name.endswith("modulation.bias"):
            t = 1 + 0.1 * t
        elif name.endswith("noise.weight"):
            t = t * noise_weight_std
        elif name.endswith("activate.bias"):
            t = t * 0.1
        elif name.startswith("to_rgb") and name.endswith(".bias") and ".conv." not in name:
            t = t * 0.1
        sd[name] = t.contiguous()
    return sd


def discriminator_schema(size, channel_multiplier=2):
    """(name, shape) of every ``Discriminator(size, channel_multiplier).state_dict()`` entry in registration order
    (model.py:634-668 with ConvLayer :564-609 -- a downsampling layer is Blur (child 0, buffer ``kernel``) + conv
    (child 1) [+ FusedLeakyReLU (child 2, ``bias``)] -- and ResBlock :612-620)."""
    ch = get_channels(channel_multiplier)
    out = []

    def conv_layer(prefix, cin, cout, k, down=False, bias=True, activate=True):
        i = 0
        if down:
            out.append((f"{prefix}.0.kernel", (4, 4)))
            i = 1
        out.append((f"{prefix}.{i}.weight", (cout, cin, k, k)))
        if bias and not activate:
            out.append((f"{prefix}.{i}.bias", (cout,)))
        if activate and bias:
            out.append((f"{prefix}.{i + 1}.bias", (cout,)))

    width = ch[size]
    conv_layer("convs.0", 3, width, 1)
    log_size = int(math.log(size, 2))
    for n, i in enumerate(range(log_size, 2, -1), start=1):
        nxt = ch[2 ** (i - 1)]
        conv_layer(f"convs.{n}.conv1", width, width, 3)
        conv_layer(f"convs.{n}.conv2", width, nxt, 3, down=True)
        conv_layer(f"convs.{n}.skip", width, nxt, 1, down=True, bias=False, activate=False)
        width = nxt
    conv_layer("final_conv", width + 1, ch[4], 3)
    out += [("final_linear.0.weight", (ch[4], ch[4] * 16)), ("final_linear.0.bias", (ch[4],)),
            ("final_linear.1.weight", (1, ch[4])), ("final_linear.1.bias", (1,))]
    return out


def seeded_discriminator_state_dict(size, channel_multiplier=2, seed=0, dtype=torch.float32):
    """Synthetic ``ckpt['d']``: N(0,1) weights (model.py:103,139), activation / linear biases N(0, 0.1^2) instead of the
    zeros of a fresh network so that every bias path is exercised; blur taps = make_kernel([1,3,3,1]) (:79-85)."""
    rng = np.random.RandomState(seed)
    k2d = ops_ref.make_kernel([1, 3, 3, 1])
    sd = {}
    for name, shape in discriminator_schema(size, channel_multiplier):
        if name.endswith(".kernel"):
            sd[name] = k2d.to(dtype)
            continue
        t = torch.from_numpy(rng.standard_normal(shape)).to(dtype)
        sd[name] = (t * 0.1 if name.endswith(".bias") else t).contiguous()
    return sd


def seeded_inputs(size, batch, style_dim=512, seed=1, dtype=torch.float32):
    """z [B, style_dim] and the explicit noise list (13 maps at 256) from a frozen numpy stream."""
    rng = np.random.RandomState(seed)
    z = torch.from_numpy(rng.standard_normal((batch, style_dim))).to(dtype)
    log_size = int(math.log(size, 2))
    noise = [torch.from_numpy(rng.standard_normal((1, 1, 4, 4))).to(dtype)]
    for i in range(3, log_size + 1):
        for _ in range(2):
            noise.append(torch.from_numpy(rng.standard_normal((1, 1, 2 ** i, 2 ** i))).to(dtype))
    return z, noise


# --------------------------------------------------------------------------------------
# functional layers


def pixel_norm(x):
    return x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    scale = (1 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        out = F.linear(x, weight * scale)
        return ops_ref.fused_leaky_relu(out, bias * lr_mul)
    return F.linear(x, weight * scale, bias=bias * lr_mul)


def modulated_conv2d(x, style_vec, weight, mod_w, mod_b, demodulate=True, upsample=False, blur_kernel=None):
    """model.py:237-278; ``weight`` is the [1,Cout,Cin,k,k] parameter."""
    batch, cin, h, w = x.shape
    _, cout, _, k, _ = weight.shape
    scale = 1 / math.sqrt(cin * k * k)
    s = equal_linear(style_vec, mod_w, mod_b).view(batch, 1, cin, 1, 1)
    wt = scale * weight * s
    if demodulate:
        d = torch.rsqrt(wt.pow(2).sum([2, 3, 4]) + 1e-8)
        wt = wt * d.view(batch, cout, 1, 1, 1)
    if upsample:
        xin = x.reshape(1, batch * cin, h, w)
        wt = wt.transpose(1, 2).reshape(batch * cin, cout, k, k)
        out = F.conv_transpose2d(xin, wt, padding=0, stride=2, groups=batch)
        out = out.view(batch, cout, out.shape[2], out.shape[3])
        # Blur(pad=(1,1), taps*4): model.py:203-209
        p = (len(blur_kernel) - 2) - (k - 1) if not torch.is_tensor(blur_kernel) else (blur_kernel.shape[0] - 2) - (k - 1)
        pad0, pad1 = (p + 1) // 2 + 1, p // 2 + 1
        taps = blur_kernel if torch.is_tensor(blur_kernel) else ops_ref.make_kernel(list(blur_kernel)) * 4
        return ops_ref.upfirdn2d(out, taps.to(out.dtype), pad=(pad0, pad1))
    xin = x.reshape(1, batch * cin, h, w)
    wt = wt.view(batch * cout, cin, k, k)
    out = F.conv2d(xin, wt, padding=k // 2, groups=batch)
    return out.view(batch, cout, out.shape[2], out.shape[3])


def styled_conv(sd, prefix, x, latent_vec, noise, upsample):
    out = modulated_conv2d(x, latent_vec, sd[f"{prefix}.conv.weight"], sd[f"{prefix}.conv.modulation.weight"],
                           sd[f"{prefix}.conv.modulation.bias"], True, upsample,
                           sd.get(f"{prefix}.conv.blur.kernel"))
    if noise is None:
        noise = torch.randn(out.shape[0], 1, out.shape[2], out.shape[3], dtype=out.dtype)
    out = out + sd[f"{prefix}.noise.weight"] * noise
    return ops_ref.fused_leaky_relu(out, sd[f"{prefix}.activate.bias"])


def to_rgb(sd, prefix, x, latent_vec, skip=None):
    out = modulated_conv2d(x, latent_vec, sd[f"{prefix}.conv.weight"], sd[f"{prefix}.conv.modulation.weight"],
                           sd[f"{prefix}.conv.modulation.bias"], demodulate=False)
    out = out + sd[f"{prefix}.bias"]
    if skip is not None:
        taps = sd[f"{prefix}.upsample.kernel"]
        p = taps.shape[0] - 2
        out = out + ops_ref.upfirdn2d(skip, taps.to(skip.dtype), up=2, down=1, pad=((p + 1) // 2 + 1, p // 2))
    return out


def mapping(sd, z, n_mlp=None, lr_mlp=0.01):
    if n_mlp is None:
        n_mlp = len([k for k in sd if k.startswith("style.") and k.endswith(".weight")])
    x = pixel_norm(z)
    for i in range(n_mlp):
        x = equal_linear(x, sd[f"style.{i + 1}.weight"], sd[f"style.{i + 1}.bias"], lr_mul=lr_mlp, activation=True)
    return x


def generator_forward(sd, styles, noise=None, input_is_latent=False, truncation=1.0, truncation_latent=None,
                      return_intermediate_activations=False, inject_index=None):
    """Restates Generator.forward (model.py:479-561).  ``styles`` is a list of 1 or 2 tensors."""
    size = sd[[k for k in sd if k.startswith("noises.")][-1]].shape[-1]
    log_size = int(math.log(size, 2))
    n_latent = log_size * 2 - 2
    num_layers = (log_size - 2) * 2 + 1
    if not input_is_latent:
        styles = [mapping(sd, s) for s in styles]
    if noise is None:
        noise = [sd[f"noises.noise_{i}"] for i in range(num_layers)]
    if truncation < 1:
        styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
    if len(styles) < 2:
        latent = styles[0].unsqueeze(1).repeat(1, n_latent, 1) if styles[0].ndim < 3 else styles[0]
    else:
        assert inject_index is not None
        latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                            styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)
    acts = {}
    out = sd["input.input"].repeat(latent.shape[0], 1, 1, 1)
    acts[0] = out
    out = styled_conv(sd, "conv1", out, latent[:, 0], noise[0], False)
    acts[1] = out
    skip = to_rgb(sd, "to_rgb1", out, latent[:, 1])
    i = 1
    for r in range(log_size - 2):
        out = styled_conv(sd, f"convs.{2 * r}", out, latent[:, i], noise[1 + 2 * r], True)
        acts[i + 1] = out
        out = styled_conv(sd, f"convs.{2 * r + 1}", out, latent[:, i + 1], noise[2 + 2 * r], False)
        acts[i + 2] = out
        skip = to_rgb(sd, f"to_rgbs.{r}", out, latent[:, i + 2], skip)
        i += 2
    if return_intermediate_activations:
        return skip, acts
    return skip, None
