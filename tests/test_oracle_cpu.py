"""CPU suite, part 1: the oracle is pinned.

(1) the two independent restatements of the reference's native ops agree (torch formulation of
    upfirdn2d.py:152-186 vs the index-level C transcription of upfirdn2d_kernel.cu:83-134);
(2) both agree with the committed known-answer vectors;
(3) the functional generator oracle reproduces the golden outputs that tests/golden/make_golden.py
    captured from the UNMODIFIED reference model.py;
(4) where /root/reference exists (this container) the oracle is bit-identical to the imported
    reference on a fresh seed -- skipped on the GPU box, where the reference never travels.
"""
import os

import numpy as np
import pytest
import torch

from oracle import c_ops, load_reference, ops_ref
from oracle import stylegan2_ref as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_ops_known_answers(golden_dir):
    g = _load(golden_dir, "ops_known_answers.npz")
    n = len([k for k in g.files if k.endswith("_cfg")])
    assert n >= 12
    for ci in range(n):
        major, ih, iw, minor, kh, kw, up, down, p0, p1 = g[f"up{ci}_cfg"].tolist()
        x, k, y = g[f"up{ci}_x"], g[f"up{ci}_k"], g[f"up{ci}_y"]
        y_c = c_ops.upfirdn2d_nhwc(x, k, up, up, down, down, p0, p1, p0, p1)
        y_t = ops_ref.upfirdn2d_nhwc(torch.from_numpy(x), torch.from_numpy(k), up, up, down, down, p0, p1, p0, p1)
        assert y_c.shape == y.shape == tuple(y_t.shape)
        np.testing.assert_allclose(y_c, y, rtol=0, atol=1e-12)
        np.testing.assert_allclose(y_t.numpy(), y, rtol=0, atol=1e-12)
        oh = ops_ref.upfirdn2d_out_size(ih, up, down, p0, p1, kh)
        assert y.shape[1] == oh
    x, b, ref = g["fba_x"], g["fba_b"], g["fba_ref"]
    for act, grad, use_b in [(3, 0, 1), (3, 1, 0), (3, 2, 0), (1, 0, 1), (1, 1, 0), (1, 2, 0), (3, 0, 0)]:
        y = g[f"fba_y_{act}{grad}{use_b}"]
        y_c = c_ops.fused_bias_act(x, b if use_b else None, ref if grad else None, act, grad, 0.2, 2 ** 0.5)
        y_t = ops_ref.fused_bias_act(torch.from_numpy(x), torch.from_numpy(b) if use_b else None,
                                     torch.from_numpy(ref) if grad else None, act, grad, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y_c, y, rtol=0, atol=1e-13)
        np.testing.assert_allclose(y_t.numpy(), y, rtol=0, atol=1e-13)


def test_ops_float32_c_vs_torch():
    rng = np.random.RandomState(7)
    for up, down, k, pad in [(1, 1, 4, (1, 1)), (2, 1, 4, (2, 1)), (1, 2, 4, (1, 1)), (2, 2, 3, (0, 1))]:
        x = rng.standard_normal((5, 13, 10, 1)).astype(np.float32)
        kern = rng.standard_normal((k, k)).astype(np.float32)
        a = c_ops.upfirdn2d_nhwc(x, kern, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
        b = ops_ref.upfirdn2d_nhwc(torch.from_numpy(x), torch.from_numpy(kern), up, up, down, down, pad[0], pad[1],
                                   pad[0], pad[1]).numpy()
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)


def test_leaky_relu_gradient_rule():
    """fused_act.py:28-37: the gradient is gated on the saved OUTPUT, bias grad sums all but dim 1."""
    x = torch.randn(3, 4, 5, 6, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, dtype=torch.float64, requires_grad=True)
    y = ops_ref.fused_leaky_relu(x, b)
    gy = torch.randn_like(y)
    gx, gb = torch.autograd.grad(y, (x, b), gy)
    manual = ops_ref.fused_bias_act(gy, None, y.detach(), 3, 1, 0.2, 2 ** 0.5)
    assert torch.allclose(gx, manual, atol=1e-14)
    assert torch.allclose(gb, manual.sum([0, 2, 3]), atol=1e-12)


def test_schema_has_135_keys():
    schema = R.state_dict_schema(256)
    assert len(schema) == 135
    assert sum(int(np.prod(s)) for n, s in schema if not n.startswith("noises.") and not n.endswith("kernel")) \
        == 30034338 - 0  # parameters only (30 034 338)


@pytest.mark.parametrize("name", ["gen16.npz", "gen32.npz"])
def test_generator_oracle_vs_golden_small(golden_dir, name):
    g = _load(golden_dir, name)
    size, sdim, n_mlp, cm, wseed, iseed, batch = g["cfg"].tolist()
    sd = R.seeded_state_dict(size, sdim, n_mlp, cm, seed=wseed)
    z, noise = R.seeded_inputs(size, batch, sdim, seed=iseed)
    with torch.no_grad():
        img, acts = R.generator_forward(sd, [z], noise=noise, return_intermediate_activations=True)
        np.testing.assert_allclose(img.numpy(), g["image"], rtol=1e-5, atol=1e-5)
        for k, v in acts.items():
            ref = g[f"act{k}"]
            got = v.numpy() if ref.shape == tuple(v.shape) else v[:, ::8].numpy()
            np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5)
        if "image_trunc07" in g.files:
            ml = torch.from_numpy(g["mean_latent"])
            a, _ = R.generator_forward(sd, [z], noise=noise, truncation=0.7, truncation_latent=ml)
            np.testing.assert_allclose(a.numpy(), g["image_trunc07"], rtol=1e-5, atol=1e-5)
            z2, _ = R.seeded_inputs(size, batch, sdim, seed=24)
            a, _ = R.generator_forward(sd, [z, z2], noise=noise, inject_index=3)
            np.testing.assert_allclose(a.numpy(), g["image_mix_inject3"], rtol=1e-5, atol=1e-5)
            a, _ = R.generator_forward(sd, [z])
            np.testing.assert_allclose(a.numpy(), g["image_stored_noise"], rtol=1e-5, atol=1e-5)
            w = R.mapping(sd, z)
            np.testing.assert_allclose(w.numpy(), g["latent_w"], rtol=1e-5, atol=1e-6)
            a, _ = R.generator_forward(sd, [torch.from_numpy(g["latent_w"])], noise=noise, input_is_latent=True)
            np.testing.assert_allclose(a.numpy(), g["image_from_w"], rtol=1e-5, atol=1e-5)


def test_generator_oracle_vs_golden_256(golden_dir):
    g = _load(golden_dir, "gen256.npz")
    size, sdim, n_mlp, cm, wseed, iseed, batch = g["cfg"].tolist()
    sd = R.seeded_state_dict(size, sdim, n_mlp, cm, seed=wseed)
    z, noise = R.seeded_inputs(size, batch, sdim, seed=iseed)
    with torch.no_grad():
        img, acts = R.generator_forward(sd, [z], noise=noise, return_intermediate_activations=True)
    np.testing.assert_allclose(img.numpy(), g["image"], rtol=1e-4, atol=1e-4)
    for k, v in acts.items():
        assert tuple(v.shape) == tuple(g[f"act{k}_shape"])
        c, h = v.shape[1], v.shape[2]
        sl = v[:, ::max(1, c // 4), ::max(1, h // 32), ::max(1, h // 32)].numpy()
        np.testing.assert_allclose(sl, g[f"act{k}_slice"], rtol=1e-4, atol=1e-4)
        t = v.double().reshape(batch, -1)
        sums = np.stack([t.sum(1).numpy(), t.abs().sum(1).numpy(), (t ** 2).sum(1).numpy()], 1)
        np.testing.assert_allclose(sums[:, 1:], g[f"act{k}_sums"][:, 1:], rtol=1e-5)


@pytest.mark.skipif(not load_reference.reference_available(), reason="reference tree only exists in the build container")
def test_oracle_bit_identical_to_imported_reference():
    ref = load_reference.load_reference_stylegan2()
    g = ref.Generator(16, 64, 3, channel_multiplier=1).eval()
    assert [k for k, _ in R.state_dict_schema(16, 64, 3, 1)] == list(g.state_dict().keys())
    sd = R.seeded_state_dict(16, 64, 3, 1, seed=99)
    g.load_state_dict(sd, strict=True)
    z, noise = R.seeded_inputs(16, 3, 64, seed=98)
    with torch.no_grad():
        a, acts_a = g([z], noise=noise, return_intermediate_activations=True)
        b, acts_b = R.generator_forward(sd, [z], noise=noise, return_intermediate_activations=True)
    assert torch.equal(a, b)
    for k in acts_a:
        assert torch.equal(acts_a[k], acts_b[k])


@pytest.mark.skipif(not load_reference.reference_available(), reason="reference tree only exists in the build container")
def test_k2_restatements_equal_the_references_own_native_statement():
    """``upfirdn2d_native`` (op/upfirdn2d.py:152-186), lifted from the reference file with ``ast``, on fresh inputs."""
    native = load_reference.load_reference_upfirdn2d_native()
    rng = np.random.RandomState(77)
    for major, ih, iw, minor, kh, kw, up, down, p0, p1 in [(3, 9, 7, 1, 4, 4, 1, 1, 1, 1), (2, 5, 6, 2, 4, 4, 2, 1, 2, 1),
                                                           (2, 8, 9, 1, 4, 4, 1, 2, 1, 1), (1, 6, 5, 3, 3, 2, 2, 2, 1, -1),
                                                           (2, 4, 4, 1, 2, 2, 2, 1, 1, 0)]:
        x, k = rng.standard_normal((major, ih, iw, minor)), rng.standard_normal((kh, kw))
        want = native(torch.from_numpy(x), torch.from_numpy(k), up, up, down, down, p0, p1, p0, p1).numpy()
        got_c = c_ops.upfirdn2d_nhwc(x, k, up, up, down, down, p0, p1, p0, p1)
        got_t = ops_ref.upfirdn2d_nhwc(torch.from_numpy(x), torch.from_numpy(k), up, up, down, down, p0, p1, p0, p1).numpy()
        np.testing.assert_allclose(got_c, want, rtol=0, atol=1e-12)
        np.testing.assert_allclose(got_t, want, rtol=0, atol=1e-12)


def _kmeans_cases():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_kmeans", os.path.join(os.path.dirname(__file__), "golden",
                                                                                     "make_golden_kmeans.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kmeans_oracle_reproduces_the_reference_label_maps(golden_dir):
    """tests/golden/kmeans_reference.npz: label maps of the reference's own FactorCatalog.predict on inputs whose
    centres come in near-duplicate pairs (most pixels decided by fp32 rounding)."""
    from oracle import kmeans_ref
    mk = _kmeans_cases()
    g = _load(golden_dir, "kmeans_reference.npz")
    for i in range(len(mk.CASES)):
        x, centres = mk.case_inputs(i)
        want = torch.from_numpy(g[f"labels{i}"].astype(np.int64))
        assert torch.equal(kmeans_ref.predict(x, centres)[0], want)
        assert torch.equal(kmeans_ref.predict_ordered(x, centres)[0], want)


@pytest.mark.parametrize("size", [1, 3, 7, 8, 9, 16, 17, 20, 31, 32, 100, 128, 300, 512, 543, 544, 1056, 2100, 4100])
def test_ordered_sum_is_torchs_cpu_sum(size):
    """The written-out association (oracle/kmeans_ref.py::ordered_sum) equals torch's inner-dimension float sum bit
    for bit -- the contract csrc/dataset_ops.hip documents."""
    from oracle import kmeans_ref
    gen = torch.Generator().manual_seed(size)
    t = torch.randn(37, 5, size, generator=gen) ** 2
    assert torch.equal(kmeans_ref.ordered_sum(t), t.sum(dim=-1))
