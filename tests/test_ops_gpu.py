"""GPU parity, part 1: K1 (fused bias + activation) and K2 (upfirdn2d) through the C ABI
(ctypes -> libsis_hip.so) against the oracle and the committed known-answer vectors.

Bar: fp64 within 1e-12 of the oracle (same operations, different association at most);
fp32 within 1e-5 relative; fp16/bf16 within a few ulp of the rounded oracle result.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ops_ref

pytestmark = pytest.mark.gpu


def _op_cases():
    # (major, ih, iw, minor, kh, kw, up, down, p0, p1)
    return [
        (6, 9, 9, 1, 4, 4, 1, 1, 1, 1), (40, 65, 65, 1, 4, 4, 1, 1, 1, 1), (3, 257, 257, 1, 4, 4, 1, 1, 1, 1),
        (3, 4, 4, 1, 4, 4, 2, 1, 2, 1), (3, 128, 128, 1, 4, 4, 2, 1, 2, 1), (4, 8, 8, 1, 4, 4, 1, 2, 1, 1),
        (4, 64, 48, 1, 4, 4, 1, 2, 2, 2), (2, 6, 5, 1, 2, 2, 2, 1, 1, 0), (2, 6, 6, 1, 2, 2, 1, 2, 0, 0),
        (2, 7, 5, 1, 3, 3, 1, 1, 1, 1), (2, 7, 5, 3, 4, 3, 1, 1, 2, -1), (2, 5, 6, 2, 5, 5, 3, 2, 2, 3),
        (1, 8, 8, 1, 1, 1, 1, 1, 0, 0), (2, 40, 70, 1, 3, 4, 1, 1, -1, 2), (1, 33, 65, 1, 4, 4, 1, 1, 0, 0),
    ]


def test_upfirdn2d_known_answers(device, golden_dir):
    import sis_hip
    g = np.load(os.path.join(golden_dir, "ops_known_answers.npz"))
    n = len([k for k in g.files if k.endswith("_cfg")])
    for ci in range(n):
        major, ih, iw, minor, kh, kw, up, down, p0, p1 = g[f"up{ci}_cfg"].tolist()
        x = torch.from_numpy(g[f"up{ci}_x"]).to(device)
        k = torch.from_numpy(g[f"up{ci}_k"]).to(device)
        y = sis_hip.upfirdn2d(x, k, up, up, down, down, p0, p1, p0, p1)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"up{ci}_y"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 2e-5), (torch.float16, 2e-2),
                                        (torch.bfloat16, 1.5e-1)])
def test_upfirdn2d_vs_oracle(device, dtype, tol):
    import sis_hip
    gen = torch.Generator().manual_seed(3)
    for (major, ih, iw, minor, kh, kw, up, down, p0, p1) in _op_cases():
        x = torch.randn(major, ih, iw, minor, generator=gen, dtype=torch.float64)
        k = torch.randn(kh, kw, generator=gen, dtype=torch.float64)
        xq, kq = x.to(dtype), k.to(dtype)
        ref = ops_ref.upfirdn2d_nhwc(xq.double(), kq.double(), up, up, down, down, p0, p1, p0, p1)
        y = sis_hip.upfirdn2d(xq.to(device), kq.to(device), up, up, down, down, p0, p1, p0, p1)
        assert y.dtype == dtype and tuple(y.shape) == tuple(ref.shape)
        err = (y.double().cpu() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (dtype, (major, ih, iw, minor, kh, kw, up, down, p0, p1), err)


def test_upfirdn2d_public_wrapper_and_gradients(device):
    """NCHW wrapper + first-order gradient (adjoint geometry, upfirdn2d.py:110-115) + second order."""
    from networks.stylegan2.op import upfirdn2d
    gen = torch.Generator().manual_seed(5)
    for up, down, k, pad in [(1, 1, 4, (1, 1)), (2, 1, 4, (2, 1)), (1, 2, 4, (1, 1)), (2, 1, 2, (1, 0)),
                             (1, 2, 2, (0, 0)), (1, 1, 3, (1, 1)), (1, 1, 4, (2, 2))]:
        x = torch.randn(2, 3, 12, 10, generator=gen, dtype=torch.float64)
        kern = torch.randn(k, k, generator=gen, dtype=torch.float64)
        xr = x.clone().requires_grad_(True)
        yr = ops_ref.upfirdn2d(xr, kern, up=up, down=down, pad=pad)
        gy = torch.randn(yr.shape, generator=gen, dtype=torch.float64)
        (gxr,) = torch.autograd.grad(yr, xr, gy)
        xd = x.to(device).requires_grad_(True)
        yd = upfirdn2d(xd, kern.to(device), up=up, down=down, pad=pad)
        assert torch.allclose(yd.detach().cpu(), yr.detach(), atol=1e-12)
        (gxd,) = torch.autograd.grad(yd, xd, gy.to(device), create_graph=True)
        assert torch.allclose(gxd.detach().cpu(), gxr, atol=1e-12), (up, down, k, pad)
        # second order: d/d(gy) of <gx, v> is the forward op applied to v
        v = torch.randn(x.shape, generator=gen, dtype=torch.float64)
        gyd = gy.to(device).requires_grad_(True)
        (gx2,) = torch.autograd.grad(upfirdn2d(xd, kern.to(device), up=up, down=down, pad=pad), xd, gyd, create_graph=True)
        (ggy,) = torch.autograd.grad(gx2, gyd, v.to(device))
        assert torch.allclose(ggy.cpu(), ops_ref.upfirdn2d(v, kern, up=up, down=down, pad=pad), atol=1e-12)


def test_upfirdn2d_empty_and_errors(device):
    import sis_hip
    k = torch.ones(4, 4, device=device)
    y = sis_hip.upfirdn2d(torch.empty(0, 8, 8, 1, device=device), k, 1, 1, 1, 1, 1, 1, 1, 1)
    assert tuple(y.shape) == (0, 7, 7, 1)
    with pytest.raises(RuntimeError):
        sis_hip.upfirdn2d(torch.randn(1, 2, 2, 1, device=device), k, 1, 1, 1, 1, -3, -3, -3, -3)
    with pytest.raises(RuntimeError):
        sis_hip.upfirdn2d(torch.randn(1, 8, 8, 1, device=device), k, 0, 1, 1, 1, 0, 0, 0, 0)


def test_fused_bias_act_known_answers(device, golden_dir):
    import sis_hip
    g = np.load(os.path.join(golden_dir, "ops_known_answers.npz"))
    x = torch.from_numpy(g["fba_x"]).to(device)
    b = torch.from_numpy(g["fba_b"]).to(device)
    ref = torch.from_numpy(g["fba_ref"]).to(device)
    e = x.new_empty(0)
    for act, grad, use_b in [(3, 0, 1), (3, 1, 0), (3, 2, 0), (1, 0, 1), (1, 1, 0), (1, 2, 0), (3, 0, 0)]:
        y = sis_hip.fused_bias_act(x, b if use_b else e, ref if grad else e, act, grad, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(y.cpu().numpy(), g[f"fba_y_{act}{grad}{use_b}"], rtol=0, atol=1e-14)
    y = sis_hip.fused_bias_act(torch.from_numpy(g["fba2_x"]).to(device), torch.from_numpy(g["fba2_b"]).to(device), e,
                               3, 0, 0.2, 2 ** 0.5)
    np.testing.assert_allclose(y.cpu().numpy(), g["fba2_y"], rtol=0, atol=1e-14)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-14), (torch.float32, 1e-6), (torch.float16, 2e-3),
                                        (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("shape", [(4, 512), (2, 16, 7, 9), (3, 8, 16, 16), (1, 5, 3), (2, 3, 4, 5, 6), (0, 4, 2, 2)])
def test_fused_leaky_relu_vs_oracle(device, dtype, tol, shape):
    """Covers the vectorised fp32 path ((3,8,16,16): numel and step_b multiples of 4) and the scalar one."""
    from networks.stylegan2.op import fused_leaky_relu
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(*shape, generator=gen, dtype=torch.float64).to(dtype)
    b = torch.randn(shape[1], generator=gen, dtype=torch.float64).to(dtype)
    ref = ops_ref.fused_leaky_relu(x.double(), b.double())
    y = fused_leaky_relu(x.to(device), b.to(device))
    assert y.dtype == dtype and y.shape == x.shape
    if x.numel():
        assert (y.double().cpu() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_fused_leaky_relu_gradients(device):
    from networks.stylegan2.op import FusedLeakyReLU, fused_leaky_relu
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(3, 6, 5, 4, generator=gen, dtype=torch.float64)
    b = torch.randn(6, generator=gen, dtype=torch.float64)
    gy = torch.randn(3, 6, 5, 4, generator=gen, dtype=torch.float64)
    xr, br = x.clone().requires_grad_(True), b.clone().requires_grad_(True)
    gxr, gbr = torch.autograd.grad(ops_ref.fused_leaky_relu(xr, br), (xr, br), gy)
    xd, bd = x.to(device).requires_grad_(True), b.to(device).requires_grad_(True)
    gxd, gbd = torch.autograd.grad(fused_leaky_relu(xd, bd), (xd, bd), gy.to(device))
    assert torch.allclose(gxd.cpu(), gxr, atol=1e-14) and torch.allclose(gbd.cpu(), gbr, atol=1e-12)
    m = FusedLeakyReLU(6).to(device).double()
    assert list(m.state_dict().keys()) == ["bias"]
    assert torch.allclose(m(x.to(device)).cpu(), ops_ref.fused_leaky_relu(x, torch.zeros(6, dtype=torch.float64)))
    # second order through FusedLeakyReLUFunctionBackward (fused_act.py:41-48)
    gyd = gy.to(device).requires_grad_(True)
    gx, gb = torch.autograd.grad(fused_leaky_relu(xd, bd), (xd, bd), gyd, create_graph=True)
    v = torch.randn(3, 6, 5, 4, generator=gen, dtype=torch.float64)
    vb = torch.randn(6, generator=gen, dtype=torch.float64)
    (ggy,) = torch.autograd.grad([gx, gb], gyd, [v.to(device), vb.to(device)])
    out = ops_ref.fused_leaky_relu(x, b)
    expect = ops_ref.fused_bias_act(v, vb, out, 3, 1, 0.2, 2 ** 0.5)
    assert torch.allclose(ggy.cpu(), expect, atol=1e-13)


def test_fused_bias_act_large_streams(device):
    """Size-independent properties at the generator's largest call (B=4 slice of [B,128,256,256]):
    positive homogeneity and the identity lrelu(x) - lrelu(-x)*... via linear mode."""
    import sis_hip
    x = torch.randn(4, 128, 256, 256, device=device)
    b = torch.randn(128, device=device)
    e = x.new_empty(0)
    y = sis_hip.fused_bias_act(x, b, e, 3, 0, 0.2, 2 ** 0.5)
    ref = torch.nn.functional.leaky_relu(x + b.view(1, -1, 1, 1), 0.2) * 2 ** 0.5
    assert torch.allclose(y, ref, rtol=1e-6, atol=1e-6)
    lin = sis_hip.fused_bias_act(x, b, e, 1, 0, 0.2, 1.0)
    assert torch.equal(lin, x + b.view(1, -1, 1, 1))
