"""Generates the golden fixtures under tests/golden/ by running the UNMODIFIED reference
modules (imported by file path from /root/reference, see oracle/load_reference.py).

Run in the build container only (``python tests/golden/make_golden.py``); the reference never
travels, only these small data files do.  A fixture is data: seeds / inputs and the outputs the
reference produced for them.

Fixtures
--------
ops_known_answers.npz   G1  per-op known-answer vectors.  upfirdn2d: outputs of the reference's OWN
                            pure-PyTorch statement ``upfirdn2d_native`` (op/upfirdn2d.py:152-186, lifted
                            with ``ast``: oracle/load_reference.py), asserted equal to 1e-12 (fp64) to
                            the two independent restatements (oracle/ops_ref.py, oracle/ops_c.c) first.
                            fused_bias_act: the reference holds only the CUDA kernel, so
                            these come from the two restatements, asserted equal to each other.
gen16.npz               G2  reference Generator(16, 64, 3, cm=1): image + all activations, B=2.
gen32.npz               G2/G4 reference Generator(32, 512, 8, cm=2): image, channel-strided
                            activations, truncation 0.7 image, style-mixing image, B=2.
gen256.npz              G3  reference Generator(256, 512, 8, cm=2), B=2: full image, per-activation
                            fp64 checksums and strided slices (full tensors are 123 MB/image).

Weights and inputs are NOT stored: they are re-derived anywhere from
``oracle.stylegan2_ref.seeded_state_dict / seeded_inputs`` (frozen ``numpy.random.RandomState``
streams, platform independent).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import c_ops, load_reference, ops_ref  # noqa: E402
from oracle import stylegan2_ref as R  # noqa: E402

OP_CASES = [
    # (major, in_h, in_w, minor, kh, kw, up, down, pad0, pad1)  -- generator modes first
    (6, 9, 9, 1, 4, 4, 1, 1, 1, 1),       # Blur after up-conv (mode 1), model.py:203-209
    (6, 17, 17, 1, 4, 4, 1, 1, 1, 1),
    (3, 4, 4, 1, 4, 4, 2, 1, 2, 1),       # RGB-skip Upsample (mode 3), model.py:39-47
    (3, 16, 16, 1, 4, 4, 2, 1, 2, 1),
    (4, 8, 8, 1, 4, 4, 1, 2, 1, 1),       # Downsample (mode 5)
    (4, 9, 7, 1, 4, 4, 1, 2, 2, 2),
    (2, 6, 5, 1, 2, 2, 2, 1, 1, 0),       # mode 4 (Haar-sized taps)
    (2, 6, 6, 1, 2, 2, 1, 2, 0, 0),       # mode 6
    (2, 7, 5, 1, 3, 3, 1, 1, 1, 1),       # mode 2
    (2, 7, 5, 3, 4, 3, 1, 1, 2, -1),      # minor > 1, non-square taps, negative pad
    (2, 5, 6, 2, 5, 5, 3, 2, 2, 3),       # no reference fast path exists for this one
    (1, 8, 8, 1, 1, 1, 1, 1, 0, 0),       # identity taps
]


def make_ops():
    rng = np.random.RandomState(1234)
    out = {}
    native = load_reference.load_reference_upfirdn2d_native()  # the reference's own statement of K2
    for ci, (major, ih, iw, minor, kh, kw, up, down, p0, p1) in enumerate(OP_CASES):
        x = rng.standard_normal((major, ih, iw, minor))
        k = rng.standard_normal((kh, kw))  # asymmetric taps pin the flip
        y_c = c_ops.upfirdn2d_nhwc(x, k, up, up, down, down, p0, p1, p0, p1)
        y_t = ops_ref.upfirdn2d_nhwc(torch.from_numpy(x), torch.from_numpy(k), up, up, down, down, p0, p1, p0,
                                     p1).numpy()
        assert y_c.shape == y_t.shape and np.abs(y_c - y_t).max() < 1e-12
        y_ref = native(torch.from_numpy(x), torch.from_numpy(k), up, up, down, down, p0, p1, p0, p1).numpy()
        assert y_ref.shape == y_c.shape and np.abs(y_ref - y_c).max() < 1e-12 and np.abs(y_ref - y_t).max() < 1e-12
        out[f"up{ci}_cfg"] = np.array([major, ih, iw, minor, kh, kw, up, down, p0, p1])
        out[f"up{ci}_x"], out[f"up{ci}_k"], out[f"up{ci}_y"] = x, k, y_ref  # the vectors ARE the reference's output
    x = rng.standard_normal((3, 5, 4, 6))
    b = rng.standard_normal(5)
    ref = rng.standard_normal((3, 5, 4, 6))
    out["fba_x"], out["fba_b"], out["fba_ref"] = x, b, ref
    for act, grad, use_b in [(3, 0, 1), (3, 1, 0), (3, 2, 0), (1, 0, 1), (1, 1, 0), (1, 2, 0), (3, 0, 0)]:
        y_c = c_ops.fused_bias_act(x, b if use_b else None, ref if grad else None, act, grad, 0.2, 2 ** 0.5)
        y_t = ops_ref.fused_bias_act(torch.from_numpy(x), torch.from_numpy(b) if use_b else None,
                                     torch.from_numpy(ref) if grad else None, act, grad, 0.2, 2 ** 0.5).numpy()
        assert np.abs(y_c - y_t).max() < 1e-12
        out[f"fba_y_{act}{grad}{use_b}"] = y_c
    x2 = rng.standard_normal((7, 12))  # the [B, 512]-shaped mapping-network call, model.py:153-155
    out["fba2_x"], out["fba2_b"] = x2, rng.standard_normal(12)
    out["fba2_y"] = c_ops.fused_bias_act(x2, out["fba2_b"], None, 3, 0, 0.2, 2 ** 0.5)
    np.savez_compressed(os.path.join(HERE, "ops_known_answers.npz"), **out)


def checksums(t):
    t = t.double()
    n = t.shape[0]
    return np.stack([t.reshape(n, -1).sum(1).numpy(), t.reshape(n, -1).abs().sum(1).numpy(),
                     (t.reshape(n, -1) ** 2).sum(1).numpy()], 1)


def act_slice(t):
    c, h = t.shape[1], t.shape[2]
    return t[:, ::max(1, c // 4), ::max(1, h // 32), ::max(1, h // 32)].contiguous().numpy()


def make_generators():
    ref = load_reference.load_reference_stylegan2()
    torch.set_grad_enabled(False)

    # ---- G2: 16 px, everything
    cfg = dict(size=16, style_dim=64, n_mlp=3, channel_multiplier=1)
    g = ref.Generator(16, 64, 3, channel_multiplier=1).eval()
    g.load_state_dict(R.seeded_state_dict(seed=11, **cfg), strict=True)
    z, noise = R.seeded_inputs(16, 2, 64, seed=12)
    img, acts = g([z], noise=noise, return_intermediate_activations=True)
    out = {"cfg": np.array([16, 64, 3, 1, 11, 12, 2]), "image": img.numpy()}
    out.update({f"act{k}": v.numpy() for k, v in acts.items()})
    np.savez_compressed(os.path.join(HERE, "gen16.npz"), **out)

    # ---- G2/G4: 32 px
    cfg = dict(size=32, style_dim=512, n_mlp=8, channel_multiplier=2)
    g = ref.Generator(32, 512, 8, channel_multiplier=2).eval()
    g.load_state_dict(R.seeded_state_dict(seed=21, **cfg), strict=True)
    z, noise = R.seeded_inputs(32, 2, 512, seed=22)
    img, acts = g([z], noise=noise, return_intermediate_activations=True)
    out = {"cfg": np.array([32, 512, 8, 2, 21, 22, 2]), "image": img.numpy()}
    out.update({f"act{k}": v[:, ::8].contiguous().numpy() for k, v in acts.items()})
    zm, _ = R.seeded_inputs(32, 64, 512, seed=23)
    mean_latent = g.style(zm).mean(0, keepdim=True)  # == Generator.mean_latent on a fixed z batch
    out["mean_latent"] = mean_latent.numpy()
    out["image_trunc07"] = g([z], noise=noise, truncation=0.7, truncation_latent=mean_latent)[0].numpy()
    z2, _ = R.seeded_inputs(32, 2, 512, seed=24)
    out["image_mix_inject3"] = g([z, z2], noise=noise, inject_index=3)[0].numpy()
    out["image_stored_noise"] = g([z], randomize_noise=False)[0].numpy()
    w = g.get_latent(z)
    out["latent_w"] = w.numpy()
    out["image_from_w"] = g([w], input_is_latent=True, noise=noise)[0].numpy()
    np.savez_compressed(os.path.join(HERE, "gen32.npz"), **out)

    # ---- G3: 256 px (BASELINE.json configs[0]/[1] model)
    cfg = dict(size=256, style_dim=512, n_mlp=8, channel_multiplier=2)
    g = ref.Generator(256, 512, 8, channel_multiplier=2).eval()
    g.load_state_dict(R.seeded_state_dict(seed=0, **cfg), strict=True)
    z, noise = R.seeded_inputs(256, 2, 512, seed=1)
    img, acts = g([z], noise=noise, return_intermediate_activations=True)
    out = {"cfg": np.array([256, 512, 8, 2, 0, 1, 2]), "image": img.numpy()}
    for k, v in acts.items():
        out[f"act{k}_shape"] = np.array(v.shape)
        out[f"act{k}_sums"] = checksums(v)
        out[f"act{k}_slice"] = act_slice(v)
    np.savez_compressed(os.path.join(HERE, "gen256.npz"), **out)


if __name__ == "__main__":
    assert load_reference.reference_available(), "run in the build container (needs /root/reference)"
    make_ops()
    make_generators()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
