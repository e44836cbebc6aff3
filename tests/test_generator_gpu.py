"""GPU parity, part 2: the generator kernels and the whole Generator.forward against the oracle
(the reference's own per-sample-weight / grouped-conv formulation, run live on the CPU) and against
the golden fixtures captured from the unmodified reference.

Stated fp32 tolerance.  The HIP path re-associates the contraction (shared weights, style on the
input side, demodulation in the epilogue; f32 MFMA = k-ordered fmaf chain).  Per layer we require
|err| <= 2e-5 * max|ref|; for the image after 14 layers of 256^2 synthesis, 2e-4 * max|ref| (measured:
see DESIGN.md).  Integer side outputs (argmax label maps) are checked bit-exact in test_labels_gpu.py.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ops_ref
from oracle import stylegan2_ref as R

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.double().cpu() - b.double()).abs().max().item() / max(b.abs().max().item(), 1e-30)


def _mk(gen, *shape):
    return torch.randn(*shape, generator=gen)


@pytest.mark.parametrize("b,cin,cout,h,w", [(2, 16, 32, 8, 8), (3, 24, 136, 16, 16), (5, 64, 128, 4, 4),
                                             (2, 32, 64, 32, 32), (1, 40, 256, 40, 72), (17, 8, 8, 4, 4),
                                             (2, 512, 512, 8, 8), (3, 64, 64, 64, 8), (6, 24, 96, 8, 8), (1, 16, 64, 2, 128)])
@pytest.mark.parametrize("fuse", [False, True])
@pytest.mark.parametrize("wino", [False, True])
def test_modconv3x3_vs_oracle(device, b, cin, cout, h, w, fuse, wino):
    """Direct MFMA kernel and the Winograd F(2x2,3x3) kernel, with and without the fused layer tail."""
    import sis_hip
    gen = torch.Generator().manual_seed(b * 1000 + cin + cout + h)
    x, style = _mk(gen, b, cin, h, w), _mk(gen, b, 48)
    weight, mod_w, mod_b = _mk(gen, 1, cout, cin, 3, 3), _mk(gen, cin, 48), 1 + 0.1 * _mk(gen, cin)
    noise, nw, bias = _mk(gen, 1, 1, h, w), 0.3 * _mk(gen, 1), 0.2 * _mk(gen, cout)
    with torch.no_grad():
        ref = R.modulated_conv2d(x, style, weight, mod_w, mod_b, demodulate=True)
        if fuse:
            ref = ops_ref.fused_leaky_relu(ref + nw * noise, bias)
        d = lambda t: t.to(device)
        wpk, wsq = sis_hip.modconv_prepack(d(weight))
        assert torch.equal(wpk.cpu(), weight[0].permute(1, 2, 3, 0).reshape(cin, 9, cout))
        s = sis_hip.equal_linear(d(style), d(mod_w), d(mod_b), 1 / 48 ** 0.5, 1.0, False)
        assert _rel(s, R.equal_linear(style, mod_w, mod_b)) < 1e-5
        ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
        u = sis_hip.modconv_prepack_wino(d(weight)) if wino else None
        y = sis_hip.modconv2d(d(x), wpk, s, ds, 3, d(noise) if fuse else None, d(nw) if fuse else None,
                              d(bias) if fuse else None, fuse_act=fuse, wino_u=u)
    assert y.shape == ref.shape
    assert _rel(y, ref) < 2e-5, _rel(y, ref)


@pytest.mark.parametrize("b,cin,cout,h,w", [(2, 16, 32, 4, 4), (3, 24, 72, 8, 8), (2, 32, 64, 16, 16),
                                             (1, 16, 128, 32, 32), (2, 8, 64, 64, 64), (9, 8, 16, 4, 4),
                                             (1, 8, 8, 5, 12)])
def test_modconv_up_vs_oracle(device, b, cin, cout, h, w):
    """Transposed stride-2 conv (all four tile classes: interior, last row, last column, corner) and the
    fused blur + noise + bias + activation that follows it."""
    import sis_hip
    gen = torch.Generator().manual_seed(b * 77 + cin + cout + h)
    x, style = _mk(gen, b, cin, h, w), _mk(gen, b, 32)
    weight, mod_w, mod_b = _mk(gen, 1, cout, cin, 3, 3), _mk(gen, cin, 32), 1 + 0.1 * _mk(gen, cin)
    noise, nw, bias = _mk(gen, b, 1, 2 * h, 2 * w), 0.3 * _mk(gen, 1), 0.2 * _mk(gen, cout)
    taps = ops_ref.make_kernel([1, 3, 3, 1]) * 4
    with torch.no_grad():
        s_ref = R.equal_linear(style, mod_w, mod_b).view(b, 1, cin, 1, 1)
        wt = (1 / (cin * 9) ** 0.5) * weight * s_ref
        wt = wt * torch.rsqrt(wt.pow(2).sum([2, 3, 4]) + 1e-8).view(b, cout, 1, 1, 1)
        t_ref = torch.nn.functional.conv_transpose2d(x.reshape(1, b * cin, h, w),
                                                     wt.transpose(1, 2).reshape(b * cin, cout, 3, 3), stride=2,
                                                     groups=b).view(b, cout, 2 * h + 1, 2 * w + 1)
        ref = R.modulated_conv2d(x, style, weight, mod_w, mod_b, True, True, taps)
        ref_act = ops_ref.fused_leaky_relu(ref + nw * noise, bias)
        d = lambda t: t.to(device)
        wpk, wsq = sis_hip.modconv_prepack(d(weight))
        s = sis_hip.equal_linear(d(style), d(mod_w), d(mod_b), 1 / 32 ** 0.5, 1.0, False)
        ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
        t = sis_hip.modconv2d_up(d(x), wpk, s, ds)
        assert _rel(t, t_ref) < 2e-5, _rel(t, t_ref)
        y = sis_hip.blur_noise_act(t, d(taps), (1, 1))
        assert _rel(y, ref) < 2e-5
        ya = sis_hip.blur_noise_act(t, d(taps), (1, 1), d(noise), d(nw), d(bias), fuse_act=True)
        assert _rel(ya, ref_act) < 2e-5
        # padded-row layout (rows of 2W+4 floats): 8-byte phase-pair stores + the row-streaming blur
        tp = sis_hip.modconv2d_up(d(x), wpk, s, ds, padded_rows=True)
        assert tuple(tp.shape) == (b, cout, 2 * h + 1, 2 * w + 4)
        assert torch.equal(tp[..., :2 * w + 1], t)
        yp = sis_hip.blur_noise_act(tp, d(taps), (1, 1), d(noise), d(nw), d(bias), fuse_act=True, in_w=2 * w + 1)
        assert _rel(yp, ref_act) < 2e-5
        yp0 = sis_hip.blur_noise_act(tp, d(taps), (1, 1), in_w=2 * w + 1)
        assert _rel(yp0, ref) < 2e-5


@pytest.mark.parametrize("b,cin,cout,h,w", [(2, 16, 64, 32, 32), (3, 40, 128, 32, 32), (1, 8, 64, 64, 64), (2, 24, 64, 34, 40),
                                             (5, 8, 64, 32, 32), (1, 8, 128, 128, 128), (2, 16, 64, 32, 64), (2, 12, 64, 32, 32), (2, 16, 64, 16, 16), (3, 32, 128, 16, 16)])
def test_modconv_up_fir_vs_oracle(device, b, cin, cout, h, w):
    """The fast-FIR transposed convolution (csrc/modconv_upfir.hip: 25 products per 2 x 2 positions on
    v_mfma_f32_16x16x4_f32) against the reference's formulation (per-sample weights, conv_transpose2d with B groups) at the
    per-layer tolerance of every other generator kernel (2e-5 max|ref|), against the 4-phase kernel, and through the fused
    blur + noise + bias + activation.  Shapes: tiles that cross from one sample into the next (3 x 289 blocks / 64), an odd
    number of block rows, non-square maps, one to two workgroup columns."""
    import sis_hip
    if cin % 8 and os.environ.get("SIS_UPFIR_WAVES") == "8":
        pytest.skip("the one-workgroup-per-CU tile stages 8 input channels per chunk")
    gen = torch.Generator().manual_seed(b * 131 + cin + cout + h + w)
    x, style = _mk(gen, b, cin, h, w), _mk(gen, b, 32)
    weight, mod_w, mod_b = _mk(gen, 1, cout, cin, 3, 3), _mk(gen, cin, 32), 1 + 0.1 * _mk(gen, cin)
    noise, nw, bias = _mk(gen, b, 1, 2 * h, 2 * w), 0.3 * _mk(gen, 1), 0.2 * _mk(gen, cout)
    taps = ops_ref.make_kernel([1, 3, 3, 1]) * 4
    with torch.no_grad():
        s_ref = R.equal_linear(style, mod_w, mod_b).view(b, 1, cin, 1, 1)
        wt = (1 / (cin * 9) ** 0.5) * weight * s_ref
        wt = wt * torch.rsqrt(wt.pow(2).sum([2, 3, 4]) + 1e-8).view(b, cout, 1, 1, 1)
        t_ref = torch.nn.functional.conv_transpose2d(x.reshape(1, b * cin, h, w),
                                                     wt.transpose(1, 2).reshape(b * cin, cout, 3, 3), stride=2,
                                                     groups=b).view(b, cout, 2 * h + 1, 2 * w + 1)
        ref_act = ops_ref.fused_leaky_relu(R.modulated_conv2d(x, style, weight, mod_w, mod_b, True, True, taps) + nw * noise, bias)
        d = lambda t: t.to(device)
        wpk, wsq = sis_hip.modconv_prepack(d(weight))
        fir_u = sis_hip.modconv_prepack_up_fir(d(weight))
        assert tuple(fir_u.shape) == (cin, 8, cout, 2)
        s = sis_hip.equal_linear(d(style), d(mod_w), d(mod_b), 1 / 32 ** 0.5, 1.0, False)
        ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
        assert sis_hip.lib().sis_modconv_up_fir_supported(b, cin, cout, h, w, 2 * w + 4)
        records = []
        sis_hip.set_profiler(records)
        try:
            tp = sis_hip.modconv2d_up(d(x), wpk, s, ds, padded_rows=True, fir_u=fir_u)
        finally:
            sis_hip.set_profiler(None)
        assert [r[0] for r in records] == ["modconv_upfir_kernel"]
        assert tuple(tp.shape) == (b, cout, 2 * h + 1, 2 * w + 4)
        assert _rel(tp[..., :2 * w + 1], t_ref) < 2e-5, _rel(tp[..., :2 * w + 1], t_ref)
        t4 = sis_hip.modconv2d_up(d(x), wpk, s, ds, padded_rows=True)                     # the 4-phase gather kernel
        assert _rel(tp[..., :2 * w + 1], t4[..., :2 * w + 1].cpu()) < 2e-5
        assert torch.isfinite(tp).all()                                                   # padding columns: unspecified but finite
        yp = sis_hip.blur_noise_act(tp, d(taps), (1, 1), d(noise), d(nw), d(bias), fuse_act=True, in_w=2 * w + 1)
        assert _rel(yp, ref_act) < 2e-5
        again = sis_hip.modconv2d_up(d(x), wpk, s, ds, padded_rows=True, fir_u=fir_u)
        assert torch.equal(again[..., :2 * w + 1], tp[..., :2 * w + 1])


def test_modconv_up_fir_declines_what_it_does_not_serve(device):
    import sis_hip
    L = sis_hip.lib()
    assert not L.sis_modconv_up_fir_supported(2, 16, 64, 8, 8, 20)       # below 16 x 16: the 4-phase kernel (split-K there)
    assert not L.sis_modconv_up_fir_supported(2, 16, 48, 32, 32, 68)     # output channels not a multiple of 64
    assert not L.sis_modconv_up_fir_supported(2, 10, 64, 32, 32, 68)     # input channels not a multiple of the 4-channel chunk
    assert not L.sis_modconv_up_fir_supported(2, 16, 64, 32, 32, 65)     # un-padded rows
    assert L.sis_modconv_up_fir_supported(32, 512, 512, 32, 32, 68) and L.sis_modconv_up_fir_supported(32, 256, 128, 128, 128, 260)


@pytest.mark.parametrize("b,cin,h", [(2, 32, 4), (3, 64, 8), (2, 128, 32), (1, 16, 6)])
def test_to_rgb_vs_oracle(device, b, cin, h):
    import sis_hip
    gen = torch.Generator().manual_seed(cin + h)
    sd = {"p.conv.weight": _mk(gen, 1, 3, cin, 1, 1), "p.conv.modulation.weight": _mk(gen, cin, 32),
          "p.conv.modulation.bias": 1 + 0.1 * _mk(gen, cin), "p.bias": 0.1 * _mk(gen, 1, 3, 1, 1),
          "p.upsample.kernel": ops_ref.make_kernel([1, 3, 3, 1]) * 4}
    x, style, skip = _mk(gen, b, cin, h, h), _mk(gen, b, 32), _mk(gen, b, 3, h // 2, h // 2)
    d = lambda t: t.to(device)
    with torch.no_grad():
        s = sis_hip.equal_linear(d(style), d(sd["p.conv.modulation.weight"]), d(sd["p.conv.modulation.bias"]),
                                 1 / 32 ** 0.5, 1.0, False)
        y0 = sis_hip.to_rgb(d(x), d(sd["p.conv.weight"]), s, d(sd["p.bias"]), 1 / cin ** 0.5)
        assert _rel(y0, R.to_rgb(sd, "p", x, style)) < 1e-5
        y1 = sis_hip.to_rgb(d(x), d(sd["p.conv.weight"]), s, d(sd["p.bias"]), 1 / cin ** 0.5, d(skip),
                            d(sd["p.upsample.kernel"]), (2, 1))
        assert _rel(y1, R.to_rgb(sd, "p", x, style, skip)) < 1e-5


def _build(size, sdim, n_mlp, cm, wseed, device):
    from networks.stylegan2.model import Generator
    g = Generator(size, sdim, n_mlp, channel_multiplier=cm)
    sd = R.seeded_state_dict(size, sdim, n_mlp, cm, seed=wseed)
    g.load_state_dict(sd, strict=True)
    return g.to(device).eval(), sd


@pytest.mark.parametrize("name", ["gen16.npz", "gen32.npz"])
def test_generator_vs_golden_small(device, golden_dir, name):
    gold = np.load(os.path.join(golden_dir, name))
    size, sdim, n_mlp, cm, wseed, iseed, batch = gold["cfg"].tolist()
    g, sd = _build(size, sdim, n_mlp, cm, wseed, device)
    z, noise = R.seeded_inputs(size, batch, sdim, seed=iseed)
    zd, nd = z.to(device), [n.to(device) for n in noise]
    with torch.no_grad():
        img, acts = g([zd], noise=nd, return_intermediate_activations=True)
        assert _rel(img, torch.from_numpy(gold["image"])) < 1e-4
        assert sorted(acts) == list(range(g.n_latent))
        for k, v in acts.items():
            ref = gold[f"act{k}"]
            got = v if ref.shape == tuple(v.shape) else v[:, ::8]
            assert _rel(got, torch.from_numpy(ref)) < 5e-5, k
        if "image_trunc07" in gold.files:
            ml = torch.from_numpy(gold["mean_latent"]).to(device)
            a, none = g([zd], noise=nd, truncation=0.7, truncation_latent=ml)
            assert none is None and _rel(a, torch.from_numpy(gold["image_trunc07"])) < 1e-4
            z2, _ = R.seeded_inputs(size, batch, sdim, seed=24)
            a, _ = g([zd, z2.to(device)], noise=nd, inject_index=3)
            assert _rel(a, torch.from_numpy(gold["image_mix_inject3"])) < 1e-4
            a, _ = g([zd], randomize_noise=False)
            assert _rel(a, torch.from_numpy(gold["image_stored_noise"])) < 1e-4
            w = g.get_latent(zd)
            assert _rel(w, torch.from_numpy(gold["latent_w"])) < 2e-5
            a, lat = g([torch.from_numpy(gold["latent_w"]).to(device)], input_is_latent=True, noise=nd,
                       return_latents=True)
            assert _rel(a, torch.from_numpy(gold["image_from_w"])) < 1e-4
            assert tuple(lat.shape) == (batch, g.n_latent, sdim)


def test_generator_256_vs_golden_and_oracle(device, golden_dir):
    """BASELINE.json configs[0]/[1] model.  Golden: full image, activation slices and checksums from the
    reference; then every activation in full against the oracle run live on the host cores."""
    gold = np.load(os.path.join(golden_dir, "gen256.npz"))
    size, sdim, n_mlp, cm, wseed, iseed, batch = gold["cfg"].tolist()
    g, sd = _build(size, sdim, n_mlp, cm, wseed, device)
    z, noise = R.seeded_inputs(size, batch, sdim, seed=iseed)
    with torch.no_grad():
        img, acts = g([z.to(device)], noise=[n.to(device) for n in noise], return_intermediate_activations=True)
        torch.cuda.synchronize()
        err_img = _rel(img, torch.from_numpy(gold["image"]))
        assert err_img < 2e-4, err_img
        for k, v in acts.items():
            assert tuple(v.shape) == tuple(gold[f"act{k}_shape"])
            c, h = v.shape[1], v.shape[2]
            sl = v[:, ::max(1, c // 4), ::max(1, h // 32), ::max(1, h // 32)]
            assert _rel(sl, torch.from_numpy(gold[f"act{k}_slice"])) < 1e-4, k
            t = v.double().reshape(batch, -1)
            sums = torch.stack([t.abs().sum(1), (t ** 2).sum(1)], 1).cpu().numpy()
            np.testing.assert_allclose(sums, gold[f"act{k}_sums"][:, 1:], rtol=1e-5)
        img_o, acts_o = R.generator_forward(sd, [z], noise=noise, return_intermediate_activations=True)
        assert _rel(img, img_o) < 2e-4
        for k in acts_o:
            assert _rel(acts[k], acts_o[k]) < 1e-4, k


def test_generator_batch32_properties(device):
    """configs[1] size (B=32): results do not depend on how images are batched (each image only depends on
    its own z row), fresh-noise mode is deterministic under a seeded device RNG, shapes are the reference's."""
    g, _ = _build(256, 512, 8, 2, 0, device)
    z = torch.randn(32, 512, generator=torch.Generator().manual_seed(1)).to(device)
    noise = [n.to(device) for n in R.seeded_inputs(256, 1, 512, seed=1)[1]]
    with torch.no_grad():
        img, acts = g([z], noise=noise, return_intermediate_activations=True)
        assert tuple(img.shape) == (32, 3, 256, 256) and torch.isfinite(img).all()
        shapes = [tuple(acts[k].shape[1:]) for k in range(14)]
        assert shapes == [(512, 4, 4)] * 2 + [(512, 8, 8)] * 2 + [(512, 16, 16)] * 2 + [(512, 32, 32)] * 2 + \
            [(512, 64, 64)] * 2 + [(256, 128, 128)] * 2 + [(128, 256, 256)] * 2
        img4, acts4 = g([z[8:12]], noise=noise, return_intermediate_activations=True)
        assert _rel(img[8:12], img4.cpu()) < 1e-5
        assert _rel(acts[13][8:12], acts4[13].cpu()) < 1e-5
        torch.manual_seed(123)
        a, _ = g([z[:2]])
        torch.manual_seed(123)
        b, _ = g([z[:2]])
        assert torch.equal(a, b)


def test_generator_tail_split_is_the_unsplit_forward(device, monkeypatch):
    """SIS_RGB_TAIL_SPLIT: the last StyledConv + ToRGB run by batch parts so that the final ToRGB overlaps the convolution of
    the next part (Generator._tail).  Whatever the number of parts -- 1 (off, the default), 2, 4 -- image and activations are
    the same values in the same tensors (per-sample work, written into batch slices), with explicit shared noise, explicit
    per-sample noise and fresh noise under a seeded device RNG."""
    g, _ = _build(64, 512, 8, 2, 3, device)
    z = torch.randn(16, 512, generator=torch.Generator().manual_seed(9)).to(device)
    shared = [n.to(device) for n in R.seeded_inputs(64, 1, 512, seed=10)[1]]
    per_sample = [torch.randn(16, 1, n.shape[2], n.shape[3], generator=torch.Generator().manual_seed(11 + i)).to(device)
                  for i, n in enumerate(shared)]
    results = {}
    for parts in ("1", "2", "4"):
        monkeypatch.setenv("SIS_RGB_TAIL_SPLIT", parts)
        with torch.no_grad():
            out = []
            for noise in (shared, per_sample, None):
                torch.manual_seed(77)
                img, acts = g([z], noise=noise, return_intermediate_activations=True)
                out += [img] + [acts[k] for k in sorted(acts)]
            torch.cuda.synchronize()
        results[parts] = out
    for parts in ("2", "4"):
        # the same per-sample arithmetic (a smaller batch may pick another tile / split plan for the convolution, hence a
        # tolerance instead of bitwise equality); fresh noise: the tail draws the layer's noise for the whole batch in one call,
        # exactly as the unsplit layer does
        for a, b in zip(results["1"], results[parts]):
            assert a.shape == b.shape and _rel(a, b.cpu()) < 1e-6, parts


def test_generator_batch32_first_and_last_sample_vs_oracle(device):
    """BASELINE.json configs[1] at its own batch size: ONE Generator.forward over 32 latents -- the tile plans, persistent
    Winograd workgroups and multi-sample tiles the batch selects (a B = 4 forward takes other plans) -- with samples 0 and 31
    compared in full (image and all 14 activations) against the oracle run live on those two latents.  Each image depends on
    its own z row only (model.py:479-561), so the oracle needs just the two rows."""
    g, sd = _build(256, 512, 8, 2, 0, device)
    z = torch.randn(32, 512, generator=torch.Generator().manual_seed(5))
    noise = R.seeded_inputs(256, 1, 512, seed=6)[1]
    with torch.no_grad():
        img, acts = g([z.to(device)], noise=[n.to(device) for n in noise], return_intermediate_activations=True)
        torch.cuda.synchronize()
        pick = torch.tensor([0, 31])
        img_o, acts_o = R.generator_forward(sd, [z[pick]], noise=noise, return_intermediate_activations=True)
    assert _rel(img[pick.to(device)], img_o) < 2e-4
    for k in acts_o:
        assert _rel(acts[k][pick.to(device)], acts_o[k]) < 1e-4, k


def test_dataset_creation_driver(device):
    """utils/dataset_creation.py surface (reference :32-58): seeded latent stream + generate_images."""
    from latent_projecting import Latents
    from utils.dataset_creation import build_latent_and_noise_generator, generate_images

    class AE:  # the reference passes an autoencoder whose .decoder is the Generator
        pass
    ae = AE()
    ae.decoder, sd = _build(32, 512, 8, 2, 21, device)
    it = iter(build_latent_and_noise_generator(ae, {"batch_size": 3, "latent_size": 512}, seed=1))
    batch = next(it)
    assert isinstance(batch, Latents) and tuple(batch.latent.shape) == (3, 512) and len(batch.noise) == 7
    torch.random.manual_seed(1)
    assert torch.equal(batch.latent, torch.randn(3, 512))  # CPU RNG stream, dataset_creation.py:33-35
    z_cpu = batch.latent.clone()
    noise_cpu = [n.cpu() for n in batch.noise]
    acts, img = generate_images(batch, ae, device=device)
    with torch.no_grad():
        ref, acts_o = R.generator_forward(sd, [z_cpu], noise=noise_cpu, return_intermediate_activations=True)
    assert _rel(img, ref) < 1e-4 and _rel(acts[5], acts_o[5]) < 5e-5
    mean_latent = ae.decoder.mean_latent(64)
    acts, img = generate_images(next(it), ae, device=device, mean_latent=mean_latent)
    assert tuple(img.shape) == (3, 3, 32, 32)


@pytest.mark.parametrize("cin,cout,hw", [(512, 512, 64), (128, 128, 256)])
def test_full_size_winograd_agrees_with_direct_kernel(device, cin, cout, hw):
    """BASELINE configs[1] layer sizes (B = 32), too large for the CPU oracle: the Winograd kernel (persistent
    workgroups, 16 tiles each at these sizes) and the independently written direct MFMA kernel must agree, with the
    fused noise / bias / activation tail."""
    import sis_hip
    g = torch.Generator().manual_seed(cin + hw)
    b = 32
    x = torch.randn(b, cin, hw, hw, generator=g).to(device)
    weight = (torch.randn(1, cout, cin, 3, 3, generator=g)).to(device)
    s = (1 + 0.3 * torch.randn(b, cin, generator=g)).to(device)
    noise = torch.randn(1, 1, hw, hw, generator=g).to(device)
    nw = torch.tensor([0.1], device=device)
    bias = (0.1 * torch.randn(cout, generator=g)).to(device)
    wpk, wsq = sis_hip.modconv_prepack(weight)
    dscale = sis_hip.modconv_demod(s, wsq, 1.0 / (cin * 9) ** 0.5, True)
    u = sis_hip.modconv_prepack_wino(weight)
    direct = sis_hip.modconv2d(x, wpk, s, dscale, 3, noise=noise, noise_weight=nw, bias=bias, fuse_act=True)
    wino = sis_hip.modconv2d(x, wpk, s, dscale, 3, noise=noise, noise_weight=nw, bias=bias, fuse_act=True, wino_u=u)
    err = (direct - wino).abs().max().item() / direct.abs().max().item()
    assert err < 2e-5, err


@pytest.mark.parametrize("cin,cout,hw", [(512, 512, 64), (128, 128, 256)])
def test_full_size_layer_slice_vs_oracle(device, cin, cout, hw):
    """BASELINE configs[1] layer sizes at B = 32 against the ORACLE (not against the build's other kernel): the batch-32
    launch runs the persistent-workgroup / XCD-placement path; samples 0 and 31 of its output are compared with
    ``R.modulated_conv2d`` (reference model.py:237-278 restated) evaluated on those two samples alone -- every sample of a
    modulated convolution depends only on its own input and style, so the slice is exact, and it costs seconds of CPU."""
    import sis_hip
    g = torch.Generator().manual_seed(cin * 3 + hw)
    b = 32
    x = torch.randn(b, cin, hw, hw, generator=g)
    style = torch.randn(b, 64, generator=g)
    weight = torch.randn(1, cout, cin, 3, 3, generator=g)
    mod_w, mod_b = torch.randn(cin, 64, generator=g), 1 + 0.1 * torch.randn(cin, generator=g)
    noise, nw, bias = torch.randn(1, 1, hw, hw, generator=g), torch.tensor([0.1]), 0.1 * torch.randn(cout, generator=g)
    pick = [0, 31]
    with torch.no_grad():
        torch.set_num_threads(16)
        ref = R.modulated_conv2d(x[pick], style[pick], weight, mod_w, mod_b, demodulate=True)
        ref = ops_ref.fused_leaky_relu(ref + nw * noise, bias)
        d = lambda t: t.to(device)
        wpk, wsq = sis_hip.modconv_prepack(d(weight))
        s = sis_hip.equal_linear(d(style), d(mod_w), d(mod_b), 1 / 64 ** 0.5, 1.0, False)
        ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
        u = sis_hip.modconv_prepack_wino(d(weight))
        for wino_u in (u, None):   # the Winograd kernel (what the generator runs at these sizes) and the direct one
            y = sis_hip.modconv2d(d(x), wpk, s, ds, 3, d(noise), d(nw), d(bias), fuse_act=True, wino_u=wino_u)
            assert tuple(y.shape) == (b, cout, hw, hw)
            assert _rel(y[pick], ref) < 2e-5, (wino_u is not None, _rel(y[pick], ref))


@pytest.mark.parametrize("size,cm,batch", [(512, 2, 3), (1024, 1, 1), (128, 2, 5), (64, 1, 1)])
def test_generator_other_resolutions_vs_oracle(device, size, cm, batch):
    """Every size of ``get_channels`` (model.py:443-455) beyond the benchmarked 256: the narrow tails (64 / 32 / 16
    channels at 512 / 1024) and odd batches take other tile plans and the direct-kernel fallbacks; each activation is
    compared in full with the oracle run live on the host cores."""
    g, sd = _build(size, 512, 8, cm, 21, device)
    z, noise = R.seeded_inputs(size, batch, 512, seed=22)
    with torch.no_grad():
        img, acts = g([z.to(device)], noise=[n.to(device) for n in noise], return_intermediate_activations=True)
        img_o, acts_o = R.generator_forward(sd, [z], noise=noise, return_intermediate_activations=True)
    assert tuple(img.shape) == (batch, 3, size, size) and len(acts) == len(acts_o) == 2 * int(np.log2(size)) - 2
    assert _rel(img, img_o) < 2e-4
    for k in acts_o:
        assert _rel(acts[k], acts_o[k]) < 1e-4, k
