"""ORACLE (test infrastructure, not product): functional CPU restatement of the SWAGAN generator forward over a
plain state_dict.

Follows /root/reference/stylegan_code_finder/networks/swagan/model.py:
  get_haar_wavelet :14-24, HaarTransform :27-46 (four K2 passes, down=2), InverseHaarTransform :48-67 (up=2,
  pad (1,0), LH/HL filters negated), ToRGB :70-95 (1x1 mod-conv to 12 coefficients, no demodulation, + bias +
  DWT(Upsample(IWT(skip)))), Generator.forward :202-285 (log_size = log2(size) - 1, final IWT).
The styled convolutions are those of oracle/stylegan2_ref.py (the reference imports them from stylegan2 too).
Pinned by tests/golden/swagan32.npz, made by the imported reference (tests/golden/make_golden_swagan.py).
"""
import math

import numpy as np
import torch

from oracle import ops_ref
from oracle import stylegan2_ref as S

CHANNELS = lambda cm: {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm,
                       1024: 16 * cm}  # noqa: E731  (model.py:120-130)


def haar_filters(dtype=torch.float32):
    low = (1 / (2 ** 0.5) * torch.ones(1, 2)).to(dtype)
    high = low.clone()
    high[0, 0] = -high[0, 0]
    return low.T * low, high.T * low, low.T * high, high.T * high


def dwt(x, filters):
    return torch.cat([ops_ref.upfirdn2d(x, k.to(x.dtype), down=2) for k in filters], 1)


def iwt(x, filters):
    bands = x.chunk(4, 1)
    return sum(ops_ref.upfirdn2d(b, k.to(x.dtype), up=2, pad=(1, 0)) for b, k in zip(bands, filters))


def to_rgb(sd, prefix, x, latent_vec, skip=None):
    out = S.modulated_conv2d(x, latent_vec, sd[f"{prefix}.conv.weight"], sd[f"{prefix}.conv.modulation.weight"],
                             sd[f"{prefix}.conv.modulation.bias"], demodulate=False)
    out = out + sd[f"{prefix}.bias"]
    if skip is not None:
        inv = [sd[f"{prefix}.iwt.{n}"] for n in ("ll", "lh", "hl", "hh")]
        fwd = [sd[f"{prefix}.dwt.{n}"] for n in ("ll", "lh", "hl", "hh")]
        taps = sd[f"{prefix}.upsample.kernel"]
        p = taps.shape[0] - 2
        up = ops_ref.upfirdn2d(iwt(skip, inv), taps.to(skip.dtype), up=2, down=1, pad=((p + 1) // 2 + 1, p // 2))
        out = out + dwt(up, fwd)
    return out


def state_dict_schema(size, style_dim=512, n_mlp=8, channel_multiplier=2):
    """Ordered (name, shape) list of networks.swagan.model.Generator(size, ...).state_dict()."""
    ch = CHANNELS(channel_multiplier)
    log_size = int(math.log(size, 2)) - 1
    out = []
    for i in range(n_mlp):
        out += [(f"style.{i + 1}.weight", (style_dim, style_dim)), (f"style.{i + 1}.bias", (style_dim,))]
    out.append(("input.input", (1, ch[4], 4, 4)))

    def styled(prefix, cin, cout, up):
        r = [(f"{prefix}.conv.weight", (1, cout, cin, 3, 3))]
        if up:
            r.append((f"{prefix}.conv.blur.kernel", (4, 4)))
        return r + [(f"{prefix}.conv.modulation.weight", (cin, style_dim)), (f"{prefix}.conv.modulation.bias", (cin,)),
                    (f"{prefix}.noise.weight", (1,)), (f"{prefix}.activate.bias", (cout,))]

    def torgb(prefix, cin, up):
        r = []
        if up:
            r += [(f"{prefix}.iwt.{n}", (2, 2)) for n in ("ll", "lh", "hl", "hh")]
            r += [(f"{prefix}.upsample.kernel", (4, 4))]
            r += [(f"{prefix}.dwt.{n}", (2, 2)) for n in ("ll", "lh", "hl", "hh")]
        return r + [(f"{prefix}.bias", (1, 12, 1, 1)), (f"{prefix}.conv.weight", (1, 12, cin, 1, 1)),
                    (f"{prefix}.conv.modulation.weight", (cin, style_dim)), (f"{prefix}.conv.modulation.bias", (cin,))]

    out += styled("conv1", ch[4], ch[4], False) + torgb("to_rgb1", ch[4], False)
    cin = ch[4]
    for r, i in enumerate(range(3, log_size + 1)):
        cout = ch[2 ** i]
        out += styled(f"convs.{2 * r}", cin, cout, True) + styled(f"convs.{2 * r + 1}", cout, cout, False)
        cin = cout
    cin = ch[4]
    for r, i in enumerate(range(3, log_size + 1)):
        out += torgb(f"to_rgbs.{r}", ch[2 ** i], True)
    for layer in range((log_size - 2) * 2 + 1):
        res = (layer + 5) // 2
        out.append((f"noises.noise_{layer}", (1, 1, 2 ** res, 2 ** res)))
    out += [(f"iwt.{n}", (2, 2)) for n in ("ll", "lh", "hl", "hh")]
    return out


def seeded_state_dict(size, style_dim=512, n_mlp=8, channel_multiplier=2, seed=0):
    """Synthetic checkpoint from a frozen numpy stream (schema order); fixed buffers (Haar filters, blur taps) take
    their defined values, noise strengths are non-zero so the noise path is exercised."""
    rng = np.random.RandomState(seed)
    ll, lh, hl, hh = haar_filters()
    fixed = {"ll": ll, "lh": lh, "hl": hl, "hh": hh}
    sd = {}
    for name, shape in state_dict_schema(size, style_dim, n_mlp, channel_multiplier):
        leaf = name.rsplit(".", 1)[-1]
        if ".iwt." in name or name.startswith("iwt."):
            sd[name] = fixed[leaf] * (-1.0 if leaf in ("lh", "hl") else 1.0)
        elif ".dwt." in name:
            sd[name] = fixed[leaf].clone()
        elif name.endswith("blur.kernel"):
            sd[name] = ops_ref.make_kernel([1, 3, 3, 1]) * 4
        elif name.endswith("upsample.kernel"):
            sd[name] = ops_ref.make_kernel([1, 3, 3, 1]) * 4
        elif name.endswith("modulation.bias"):
            sd[name] = torch.from_numpy(1.0 + 0.1 * rng.randn(*shape)).float()
        elif name.endswith("noise.weight"):
            sd[name] = torch.from_numpy(0.1 * rng.randn(*shape)).float()
        elif name.endswith(".bias"):
            sd[name] = torch.from_numpy(0.1 * rng.randn(*shape)).float()
        else:
            sd[name] = torch.from_numpy(rng.randn(*shape)).float()
    return sd


def generator_forward(sd, styles, noise=None, input_is_latent=False, truncation=1.0, truncation_latent=None,
                      return_intermediate_activations=False, inject_index=None):
    """Restates swagan Generator.forward (model.py:202-285)."""
    num_layers = len([k for k in sd if k.startswith("noises.")])
    log_size = (num_layers - 1) // 2 + 2
    n_latent = log_size * 2 - 2
    if not input_is_latent:
        styles = [S.mapping(sd, s) for s in styles]
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
    out = S.styled_conv(sd, "conv1", out, latent[:, 0], noise[0], False)
    acts[1] = out
    skip = to_rgb(sd, "to_rgb1", out, latent[:, 1])
    i = 1
    for r in range(log_size - 2):
        out = S.styled_conv(sd, f"convs.{2 * r}", out, latent[:, i], noise[1 + 2 * r], True)
        acts[i + 1] = out
        out = S.styled_conv(sd, f"convs.{2 * r + 1}", out, latent[:, i + 1], noise[2 + 2 * r], False)
        acts[i + 2] = out
        skip = to_rgb(sd, f"to_rgbs.{r}", out, latent[:, i + 2], skip)
        i += 2
    image = iwt(skip, [sd[f"iwt.{n}"] for n in ("ll", "lh", "hl", "hh")])
    return (image, acts) if return_intermediate_activations else (image, None)
