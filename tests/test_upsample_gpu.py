"""GPU parity: bilinear upsampling with align_corners (csrc/upsample_ops.hip) against torch's fp32 reference, forward
and backward, f32 / bf16 / f16.  The interpolation weights are formed in fp32 exactly as ATen forms them (src = dst * scale):
against a float64 reference both this kernel and ATen's fp32 kernel sit at ~1e-5 on unit-variance data (measured
1.04e-5 vs 1.08e-5 at 64 -> 128), hence atol 4e-5; 16-bit types: one rounding of the fp32 result."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape,size", [((2, 3, 8, 8), (16, 16)), ((1, 5, 7, 9), (14, 18)), ((2, 4, 16, 12), (32, 27)),
                                        ((1, 2, 1, 5), (2, 10)), ((1, 16, 64, 64), (128, 128)), ((2, 2, 6, 6), (6, 6))])
def test_forward_backward_f32(device, shape, size):
    import sis_hip
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(device)
    gy = torch.randn(shape[0], shape[1], *size, generator=g).to(device)
    y = sis_hip.upsample_bilinear(x, *size)
    xr = x.double().requires_grad_(True)
    ref = F.interpolate(xr, size=size, mode="bilinear", align_corners=True)
    ref.backward(gy.double())
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-5, atol=4e-5)
    gx = sis_hip.upsample_bilinear(x, *size, grad_output=gy)
    np.testing.assert_allclose(gx.cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("dtype,tol", [(torch.bfloat16, 1e-2), (torch.float16, 2e-3)])
def test_module_16bit_and_autograd(device, dtype, tol):
    from networks.hip_upsample import HipUpsamplingBilinear2d
    up = HipUpsamplingBilinear2d(scale_factor=2)
    x = torch.randn(2, 8, 16, 16, device=device).to(dtype).requires_grad_(True)
    y = up(x)
    assert y.dtype == dtype and tuple(y.shape) == (2, 8, 32, 32)
    gy = torch.randn_like(y)
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    ref = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    ref.backward(gy.float())
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), ref.detach().cpu().numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(x.grad.float().cpu().numpy(), xr.grad.cpu().numpy(), rtol=tol, atol=4 * tol)
    assert up(torch.zeros(1, 1, 4, 4)).shape == (1, 1, 8, 8)  # CPU tensors take the torch path
    assert list(up.state_dict().keys()) == []


@pytest.mark.parametrize("shape", [(2, 3, 300, 280), (1, 2, 2, 2), (3, 4, 17, 33)])
def test_x2_forward_tiles(device, shape):
    """The LDS-tiled x2 forward: several row / column tiles per plane (560 output columns > one 512-wide tile), the smallest
    plane, odd sizes (scalar stores) -- against the float64 reference."""
    import sis_hip
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(shape[2])).to(device)
    size = (2 * shape[2], 2 * shape[3])
    ref = F.interpolate(x.double(), size=size, mode="bilinear", align_corners=True)
    # fp32 source coordinates (dst * scale, as ATen forms them) carry ~dst * 2^-24 of error: at 600 outputs the weights are off
    # by up to ~4e-5, times the difference of two unit-variance neighbours -> atol 3e-4 against float64 (measured 7.3e-5 here,
    # ATen's own fp32 kernel 1.3e-4 on the same input)
    y = sis_hip.upsample_bilinear(x, *size)
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-5, atol=3e-4)
    yb = sis_hip.upsample_bilinear(x.bfloat16(), *size)
    refb = F.interpolate(x.bfloat16().double(), size=size, mode="bilinear", align_corners=True)
    np.testing.assert_allclose(yb.float().cpu().numpy(), refb.float().cpu().numpy(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("shape", [(1, 2, 4, 512), (2, 3, 40, 256), (1, 4, 32, 32), (2, 2, 2, 4), (1, 1, 3, 1024), (8, 8, 64, 64), (2, 2, 5, 16)])
def test_x2_direct_kernels(device, shape):
    """The LDS-free x2 kernels (csrc/upsample_ops.hip, bilinear_up2_*_direct_kernel: static source indices, 4 source columns per
    lane, neighbours by wave shuffle): rows narrower than a wave (several rows per wave), exactly a wave, wider than a wave (the
    neighbour across the wave boundary is loaded), two-row planes -- forward and backward against float64, f32 and bf16."""
    import sis_hip
    g = torch.Generator().manual_seed(shape[2] + shape[3])
    x = torch.randn(*shape, generator=g).to(device)
    size = (2 * shape[2], 2 * shape[3])
    gy = torch.randn(shape[0], shape[1], *size, generator=g).to(device)
    xr = x.double().requires_grad_(True)
    ref = F.interpolate(xr, size=size, mode="bilinear", align_corners=True)
    ref.backward(gy.double())
    y = sis_hip.upsample_bilinear(x, *size)
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-5, atol=3e-4)
    gx = sis_hip.upsample_bilinear(x, *size, grad_output=gy)
    np.testing.assert_allclose(gx.cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=1e-5, atol=1e-3)
    xb, gb = x.bfloat16(), gy.bfloat16()
    xbr = xb.double().requires_grad_(True)
    refb = F.interpolate(xbr, size=size, mode="bilinear", align_corners=True)
    refb.backward(gb.double())
    yb = sis_hip.upsample_bilinear(xb, *size)
    np.testing.assert_allclose(yb.float().cpu().numpy(), refb.detach().float().cpu().numpy(), rtol=1e-2, atol=1e-2)
    gxb = sis_hip.upsample_bilinear(xb, *size, grad_output=gb)
    np.testing.assert_allclose(gxb.float().cpu().numpy(), xbr.grad.float().cpu().numpy(), rtol=1e-2, atol=4e-2)
    assert torch.equal(y, sis_hip.upsample_bilinear(x, *size)) and torch.equal(gx, sis_hip.upsample_bilinear(x, *size, grad_output=gy))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("b,c,s,h,w", [(2, 16, 8, 16, 16), (3, 5, 3, 12, 20), (1, 64, 64, 64, 64)])
def test_upsample_cat_fused(device, dtype, b, c, s, h, w):
    """torch.cat([up2x(x), skip], 1) with the upsampling written straight into the wide tensor and its gradient read from the
    wide gradient in place: bitwise the two-step result (same kernels, strided addressing), gradients of both inputs included."""
    from networks.hip_upsample import HipUpsamplingBilinear2d, upsample2x_cat
    g = torch.Generator().manual_seed(b + c + h)
    x = torch.randn(b, c, h, w, generator=g).to(device).to(dtype).requires_grad_(True)
    skip = torch.randn(b, s, 2 * h, 2 * w, generator=g).to(device).to(dtype).requires_grad_(True)
    gy = torch.randn(b, c + s, 2 * h, 2 * w, generator=g).to(device).to(dtype)
    fused = upsample2x_cat(x, skip)
    assert fused is not None
    fused.backward(gy)
    gx, gs = x.grad.clone(), skip.grad.clone()
    x.grad = skip.grad = None
    two = torch.cat([HipUpsamplingBilinear2d(scale_factor=2)(x), skip], 1)
    two.backward(gy)
    assert torch.equal(fused, two) and torch.equal(gx, x.grad) and torch.equal(gs, skip.grad)


@pytest.mark.parametrize("shape", [(64, 3, 7, 7), (256, 64, 1, 1), (128, 128, 3, 3), (1024, 1024, 1, 1), (5, 3, 3, 3)])
def test_weight_standardisation(device, shape):
    """csrc/weight_std.hip against the reference formula (vit_seg_modeling_resnet_skip.py:22-27), both directions."""
    import sis_hip
    g = torch.Generator().manual_seed(shape[0])
    w = (torch.randn(*shape, generator=g) * 0.3 + 0.1).to(device)
    gy = torch.randn(*shape, generator=g).to(device)
    w_hat, invstd = sis_hip.weight_std_fwd(w, 1e-5)
    wr = w.double().requires_grad_(True)
    var, mean = torch.var_mean(wr, dim=[1, 2, 3], keepdim=True, unbiased=False)
    ref = (wr - mean) / torch.sqrt(var + 1e-5)
    ref.backward(gy.double())
    np.testing.assert_allclose(w_hat.cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    dw = sis_hip.weight_std_bwd(gy, w, invstd, 1e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), wr.grad.float().cpu().numpy(), rtol=1e-4, atol=1e-4 * float(wr.grad.abs().max()))
    w16, _ = sis_hip.weight_std_fwd(w, 1e-5, torch.bfloat16)
    np.testing.assert_allclose(w16.float().cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-2, atol=1e-2)
    dw16 = sis_hip.weight_std_bwd(gy.bfloat16(), w, invstd, 1e-5)
    np.testing.assert_allclose(dw16.cpu().numpy(), wr.grad.float().cpu().numpy(), rtol=5e-2, atol=2e-2 * float(wr.grad.abs().max()))


@pytest.mark.parametrize("shape,groups,relu", [((2, 64, 16, 16), 32, True), ((3, 256, 7, 9), 32, False), ((2, 96, 5, 5), 96, True),
                                               ((1, 64, 127, 127), 32, True), ((1, 6, 5, 5), 3, True)])
def test_group_norm_relu(device, shape, groups, relu):
    """csrc/group_norm.hip against F.group_norm (+ relu) in float64, fp32 and bf16 tensors, both directions."""
    import sis_hip
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(*shape, generator=g) * 2 + 0.5).to(device)
    gamma = (1 + 0.2 * torch.randn(shape[1], generator=g)).to(device)
    beta = (0.3 * torch.randn(shape[1], generator=g)).to(device)
    gy = torch.randn(*shape, generator=g).to(device)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.group_norm(xr, groups, gr, br, 1e-6)
    if relu:
        ref = F.relu(ref)
    ref.backward(gy.double())
    y, mean, rstd = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, relu)
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-4, atol=1e-5)
    dx, dg, db = sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, relu)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    # bf16 tensors in and out (fp32 arithmetic inside), fp32 output on request
    xb = x.bfloat16()
    yb, mb, rb = sis_hip.group_norm_fwd(xb, gamma, beta, groups, 1e-6, relu)
    assert yb.dtype == torch.bfloat16
    refb = F.group_norm(xb.double(), groups, gamma.double(), beta.double(), 1e-6)
    refb = F.relu(refb) if relu else refb
    np.testing.assert_allclose(yb.float().cpu().numpy(), refb.float().cpu().numpy(), rtol=1e-2, atol=1e-2)
    y32, _, _ = sis_hip.group_norm_fwd(xb, gamma, beta, groups, 1e-6, relu, torch.float32)
    np.testing.assert_allclose(y32.cpu().numpy(), refb.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    dxb, dgb, dbb = sis_hip.group_norm_bwd(gy.bfloat16(), xb, mb, rb, gamma, beta, groups, relu)
    assert dxb.dtype == torch.bfloat16
    xq = xb.double().requires_grad_(True)  # reference on the SAME (bf16-rounded) input: the ReLU mask depends on it
    rq = F.group_norm(xq, groups, gamma.double(), beta.double(), 1e-6)
    (F.relu(rq) if relu else rq).backward(gy.bfloat16().double())
    scale = float(xq.grad.abs().max())
    np.testing.assert_allclose(dxb.float().cpu().numpy(), xq.grad.float().cpu().numpy(), rtol=2e-2, atol=1e-2 * scale)


@pytest.mark.parametrize("shape,groups", [((8, 256, 24, 24), 32), ((2, 64, 200, 200), 32), ((3, 96, 5, 5), 96), ((8, 1024, 8, 8), 32)])
def test_group_norm_merge_inside_the_statistics_launch(device, shape, groups, monkeypatch):
    """The per-group merge done by the workgroup that completes the group (device-scope counters, csrc/group_norm.hip) gives
    bitwise the results of the merge as a launch of its own, call after call (the counters are left at zero), incl. planes cut
    into several slices (200 x 200 > 16 384 elements) and one-channel groups."""
    import sis_hip
    g = torch.Generator().manual_seed(shape[1] + shape[2])
    x = (torch.randn(*shape, generator=g) * 2 + 0.5).to(device).bfloat16()
    gamma = (1 + 0.2 * torch.randn(shape[1], generator=g)).to(device)
    beta = (0.3 * torch.randn(shape[1], generator=g)).to(device)
    gy = torch.randn(*shape, generator=g).to(device).bfloat16()

    def run():
        y, mean, rstd = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True)
        return (y, mean, rstd) + tuple(sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, True))

    monkeypatch.setattr(sis_hip, "_GN_FUSED_FINISH", False)
    want = run()
    monkeypatch.setattr(sis_hip, "_GN_FUSED_FINISH", True)
    for _ in range(3):
        got = run()
        for u, v in zip(got, want):
            assert torch.equal(u, v)
    counters = sis_hip._group_counters(x.device, shape[0] * groups)
    assert int(counters.abs().sum()) == 0


@pytest.mark.parametrize("shape,groups,residual", [((2, 256, 32, 32), 32, False), ((3, 1024, 32, 32), 32, True), ((2, 128, 64, 64), 32, False),
                                                   ((2, 64, 32, 32), 64, False), ((2, 32, 64, 64), 32, True), ((8, 256, 64, 64), 32, True)])
def test_group_norm_single_pass_groups(device, shape, groups, residual, monkeypatch):
    """Groups that fit one workgroup's registers (bf16 x, hw % 512 == 0, <= 32 768 elements: blocks 2 and 3 of the trunk) take
    the single-pass kernels of csrc/group_norm.hip -- all three iteration counts, one-channel groups, the residual + dual
    output form with fp32 gradients, the plain bf16 form: against F.group_norm in float64 on the same bf16-rounded input
    (tolerances of test_group_norm_relu / _residual_relu) and against the two-launch kernels (fp32 results to 1e-5)."""
    import sis_hip
    g = torch.Generator().manual_seed(shape[1] + shape[2])
    c = shape[1]
    x = (torch.randn(*shape, generator=g) * 2 + 0.5).to(device).bfloat16()
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(device)
    beta = (0.3 * torch.randn(c, generator=g)).to(device)
    res = torch.randn(*shape, generator=g).to(device) if residual else None
    gy = torch.randn(*shape, generator=g).to(device)
    g_lp = torch.randn(*shape, generator=g).to(device).bfloat16() if residual else None
    if not residual:
        gy = gy.bfloat16()

    def run():
        if residual:
            y, mean, rstd, y_lp = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True, residual=res, low_precision_copy=True)
            assert torch.equal(y_lp, y.to(x.dtype))
            dx, dg, db, dres = sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, True, y_mask=y, want_residual_grad=True,
                                                      grad_y_lp=g_lp)
            return y, mean, rstd, dx, dg, db, dres
        y, mean, rstd = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True)
        return (y, mean, rstd) + tuple(sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, True))

    got = run()
    assert all(torch.equal(u, v) for u, v in zip(got, run()))   # repeatable, counters left at zero
    if residual:   # gate bits instead of the saved output: same values
        y, mean, rstd, y_lp, gate = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True, residual=res, low_precision_copy=True,
                                                           want_gate=True)
        bits = ((gate.cpu().numpy()[:, None] >> np.arange(8)) & 1).reshape(-1)[:y.numel()]
        assert torch.equal(y, got[0]) and np.array_equal(bits.astype(bool), (y > 0).cpu().numpy().reshape(-1))
        via = sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, True, want_residual_grad=True, grad_y_lp=g_lp, gate=gate)
        for u, v in zip(via, (got[3], got[4], got[5], got[6])):
            assert torch.equal(u, v)
    monkeypatch.setattr(sis_hip, "_GN_FUSED_FINISH", False)     # no counters: the two-launch kernels
    two = run()
    monkeypatch.setattr(sis_hip, "_GN_FUSED_FINISH", True)
    for u, v in zip(got[1:3], two[1:3]):
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-5, atol=1e-6)   # mean, rstd
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.group_norm(xr, groups, gr, br, 1e-6)
    if residual:
        rr = res.double().requires_grad_(True)
        ref = F.relu(ref + rr)
        ref.backward(gy.double() + g_lp.double())
        np.testing.assert_allclose(got[0].cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(got[6].cpu().numpy(), rr.grad.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    else:
        ref = F.relu(ref)
        ref.backward(gy.double())
        np.testing.assert_allclose(got[0].float().cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-2, atol=1e-2)
    scale = float(xr.grad.abs().max())
    np.testing.assert_allclose(got[3].float().cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=2e-2, atol=1e-2 * scale)
    np.testing.assert_allclose(got[4].cpu().numpy(), gr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3 * float(gr.grad.abs().max()))
    np.testing.assert_allclose(got[5].cpu().numpy(), br.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3 * float(br.grad.abs().max()))


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("shape,relu", [((4, 16, 12, 12), True), ((2, 64, 33, 31), False), ((8, 256, 16, 16), True), ((3, 16, 128, 160), True),
                                        ((70, 8, 64, 64), False)])
def test_batch_norm_train_relu(device, monkeypatch, shape, relu, fused):
    """Batch-norm mode of csrc/group_norm.hip against F.batch_norm(training=True) (+ relu), incl. running statistics; the
    per-channel merges inside the statistics launches (``fused``: the workgroup that completes a channel, csrc/sis_xwg.h) or
    as launches of their own -- planes cut into several slices, and more samples than the 64 lanes of the merging wave."""
    import sis_hip
    monkeypatch.setattr(sis_hip, "_GN_FUSED_FINISH", fused)
    g = torch.Generator().manual_seed(shape[1] + shape[2])
    x = (torch.randn(*shape, generator=g) * 1.5 - 0.3).to(device)
    c = shape[1]
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).to(device)
    beta = (0.3 * torch.randn(c, generator=g)).to(device)
    gy = torch.randn(*shape, generator=g).to(device)
    rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    ref = F.relu(ref) if relu else ref
    ref.backward(gy.double())
    y, mean, rstd = sis_hip.batch_norm_train_fwd(x, gamma, beta, rm, rv, 1e-5, 0.1, relu)
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(rm.cpu().numpy(), rm_ref.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv.cpu().numpy(), rv_ref.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    dx, dg, db = sis_hip.batch_norm_train_bwd(gy, x, mean, rstd, gamma, beta, relu)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    yb, _, _ = sis_hip.batch_norm_train_fwd(x.bfloat16(), gamma, beta, None, None, 1e-5, 0.1, relu)
    assert yb.dtype == torch.bfloat16
    np.testing.assert_allclose(yb.float().cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=3e-2, atol=3e-2)
    assert sis_hip.group_counters_are_zero()


@pytest.mark.parametrize("shape", [(2, 64, 12, 12), (2, 64, 9, 7), (1, 64, 3, 3)])
def test_group_norm_residual_relu(device, shape):
    """y = relu(group_norm(x) + residual) in one pass (bottleneck tail), bf16 x, fp32 residual / output / gradients;
    odd planes take the flat-vector kernels (a lane's 4 elements straddle two planes) or, total % 4 != 0, the scalar ones."""
    import sis_hip
    g = torch.Generator().manual_seed(9)
    groups = 32
    x = (torch.randn(*shape, generator=g) * 2).to(device).bfloat16()
    res = torch.randn(*shape, generator=g).to(device)
    gamma = (1 + 0.2 * torch.randn(64, generator=g)).to(device)
    beta = (0.3 * torch.randn(64, generator=g)).to(device)
    gy = torch.randn(*shape, generator=g).to(device)
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = F.relu(F.group_norm(xr, groups, gr, br, 1e-6) + rr)
    ref.backward(gy.double())
    y, mean, rstd = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True, residual=res)
    assert y.dtype == torch.float32
    np.testing.assert_allclose(y.cpu().numpy(), ref.detach().float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    dx, dg, db, dres = sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, True, y_mask=y, want_residual_grad=True)
    np.testing.assert_allclose(dres.cpu().numpy(), rr.grad.float().cpu().numpy(), rtol=1e-5, atol=1e-6)
    scale = float(xr.grad.abs().max())
    np.testing.assert_allclose(dx.float().cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=2e-2, atol=1e-2 * scale)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    # the ReLU gate as one bit per element written by the forward: bitwise the backward that reads the saved output's sign
    y3, _, _, gate = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True, residual=res, want_gate=True)
    assert torch.equal(y3, y) and gate.dtype == torch.uint8
    bits = ((gate.cpu().numpy()[:, None] >> np.arange(8)) & 1).reshape(-1)[:y.numel()]
    assert np.array_equal(bits.astype(bool), (y > 0).cpu().numpy().reshape(-1))
    via_bits = sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, True, want_residual_grad=True, gate=gate)
    for u, v in zip(via_bits, (dx, dg, db, dres)):
        assert torch.equal(u, v)
    # dual output: the same pass also writes y rounded to x's dtype; the gradient arriving through that copy is added on load
    y2, _, _, y_lp = sis_hip.group_norm_fwd(x, gamma, beta, groups, 1e-6, True, residual=res, low_precision_copy=True)
    assert torch.equal(y2, y) and y_lp.dtype == x.dtype and torch.equal(y_lp, y.to(x.dtype))
    g_lp = torch.randn(*shape, generator=g).to(device).bfloat16()
    got = sis_hip.group_norm_bwd(gy, x, mean, rstd, gamma, beta, groups, True, y_mask=y, want_residual_grad=True, grad_y_lp=g_lp)
    want = sis_hip.group_norm_bwd(gy + g_lp.float(), x, mean, rstd, gamma, beta, groups, True, y_mask=y, want_residual_grad=True)
    for u, v in zip(got, want):
        assert torch.equal(u, v)


@pytest.mark.parametrize("rows,n", [(37, 768), (1024, 256), (5, 1024)])
def test_layer_norm(device, rows, n):
    """csrc/layer_norm.hip against F.layer_norm in float64, fp32 and bf16 tensors, both directions."""
    import sis_hip
    g = torch.Generator().manual_seed(n + rows)
    x = (torch.randn(rows, n, generator=g) * 2 + 0.5).to(device)
    gamma = (1 + 0.2 * torch.randn(n, generator=g)).to(device)
    beta = (0.3 * torch.randn(n, generator=g)).to(device)
    gy = torch.randn(rows, n, generator=g).to(device)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    F.layer_norm(xr, (n,), gr, br, 1e-6).backward(gy.double())
    ref = F.layer_norm(x.double(), (n,), gamma.double(), beta.double(), 1e-6)
    y, mean, rstd = sis_hip.layer_norm_fwd(x, gamma, beta, 1e-6)
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-5)
    dx, dg, db = sis_hip.layer_norm_bwd(gy, x, mean, rstd, gamma)
    np.testing.assert_allclose(dx.cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(dg.cpu().numpy(), gr.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db.cpu().numpy(), br.grad.float().cpu().numpy(), rtol=1e-3, atol=1e-3)
    yb, _, _ = sis_hip.layer_norm_fwd(x, gamma, beta, 1e-6, torch.bfloat16)  # fp32 in, bf16 out (autocast case)
    assert yb.dtype == torch.bfloat16
    np.testing.assert_allclose(yb.float().cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-2, atol=1e-2)
    dxb, dgb, dbb = sis_hip.layer_norm_bwd(gy.bfloat16(), x, mean, rstd, gamma)  # bf16 gradient, fp32 x
    assert dxb.dtype == torch.float32
    np.testing.assert_allclose(dxb.cpu().numpy(), xr.grad.float().cpu().numpy(), rtol=5e-2, atol=2e-2 * float(xr.grad.abs().max()))
    assert not sis_hip.layer_norm_supported(x[:, :100], 100)


@pytest.mark.parametrize("rows,n,dt", [(8192, 768, torch.bfloat16), (77, 3072, torch.bfloat16), (513, 260, torch.float32),
                                        (5, 4, torch.float16)])
def test_column_sum(device, rows, n, dt):
    """csrc/column_sum.hip: float32 column sums of 16-bit / fp32 matrices against a float64 sum."""
    import sis_hip
    x = torch.randn(rows, n, generator=torch.Generator().manual_seed(rows + n)).to(device).to(dt)
    got = sis_hip.column_sum(x)
    want = x.double().sum(0)
    assert got.dtype == torch.float32
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-4 * rows ** 0.5)
    assert torch.equal(got, sis_hip.column_sum(x))  # deterministic


def test_amp_linear_matches_autocast_linear(device):
    """networks.trans_u_net.vit_encoder.linear under bf16 autocast: same output as F.linear, weight / bias gradients equal
    to the float64 ones up to the 16-bit inputs' rounding (they are accumulated and returned in float32)."""
    from networks.trans_u_net import vit_encoder as V
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 96, 768, generator=g).to(device).bfloat16().requires_grad_(True)
    w = (torch.randn(2304, 768, generator=g) * 0.03).to(device).requires_grad_(True)
    b = (torch.randn(2304, generator=g) * 0.1).to(device).requires_grad_(True)
    gy = torch.randn(4, 96, 2304, generator=g).to(device).bfloat16()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = V.linear(x, w, b)
        y_ref = F.linear(x, w, b)
    assert y.dtype == torch.bfloat16 and y.grad_fn.name().startswith("_AmpLinearFn")
    np.testing.assert_allclose(y.float().detach().cpu().numpy(), y_ref.float().detach().cpu().numpy(), rtol=2e-2, atol=2e-2)
    dx, dw, db = torch.autograd.grad(y, (x, w, b), gy)
    assert dw.dtype == torch.float32 and db.dtype == torch.float32 and dx.dtype == torch.bfloat16
    x64, w64, g64 = x.detach().double(), w.detach().bfloat16().double(), gy.double()
    np.testing.assert_allclose(db.cpu().numpy(), g64.sum((0, 1)).cpu().numpy(), rtol=1e-5, atol=1e-4)
    want_dw = g64.reshape(-1, 2304).t() @ x64.reshape(-1, 768)
    np.testing.assert_allclose(dw.cpu().numpy(), want_dw.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(want_dw.abs().max()))
    want_dx = g64 @ w64
    np.testing.assert_allclose(dx.float().cpu().numpy(), want_dx.cpu().numpy(), rtol=2e-2, atol=2e-2 * float(want_dx.abs().max()))
