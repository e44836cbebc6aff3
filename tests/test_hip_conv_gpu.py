"""GPU parity: the plain 3x3 Winograd convolution (sis_conv3x3) and its autograd wrapper against ATen fp32.
Tolerance: Winograd F(2x2,3x3) in fp32 re-associates the 9-tap sums; 2e-5 of the output range per layer forward, and
the same for the data gradient (the library path it replaces is itself a Winograd kernel)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

SHAPES = [(2, 64, 128, 32, 32), (2, 128, 64, 16, 32), (3, 64, 64, 64, 64), (3, 128, 64, 16, 24), (1, 256, 8, 8, 8), (16, 64, 64, 64, 64), (2, 2048, 512, 32, 32),
          (5, 8, 8, 2, 4),
          # round-2 chunk loop: 4 / 8 DMA parts per input tile (tall, narrow and multi-sample tiles), odd chunk counts, partial co block
          (3, 64, 64, 64, 8), (2, 64, 64, 8, 8), (1, 16, 64, 2, 128), (2, 24, 96, 32, 32), (9, 40, 72, 4, 8)]


@pytest.mark.parametrize("batch,cin,cout,h,w,dil", [s + (1,) for s in SHAPES] + [(2, 64, 32, 32, 32, 2), (3, 256, 256, 32, 32, 2),
                                                   (2, 64, 64, 32, 32, 4), (1, 8, 16, 16, 24, 2)])
def test_forward_and_data_gradient(device, batch, cin, cout, h, w, dil):
    import sis_hip
    from networks.hip_conv import conv3x3
    g = torch.Generator().manual_seed(batch * 1000 + cin)
    x = torch.randn(batch, cin, h, w, generator=g).to(device).requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (cin * 9) ** -0.5).to(device).requires_grad_(True)
    gy = torch.randn(batch, cout, h, w, generator=g).to(device)
    assert sis_hip.conv3x3_supported(x, wt, dil)
    y = conv3x3(x, wt, dil)
    y.backward(gy)
    gx, gw = x.grad.clone(), wt.grad.clone()
    x.grad = wt.grad = None
    ref = F.conv2d(x.double(), wt.double(), padding=dil, dilation=dil)
    ref.backward(gy.double())
    for got, want, name in ((y, ref, "y"), (gx, x.grad, "dx"), (gw, wt.grad, "dw")):
        want = want.detach().float()
        err = (got.detach() - want).abs().max().item() / want.abs().max().item()
        assert err < (2e-5 if name != "dw" else 2e-4), (name, err)


@pytest.mark.parametrize("batch,cin,cout,h,w", [(2, 64, 128, 32, 32), (1, 128, 64, 16, 16), (3, 64, 64, 64, 48), (16, 512, 256, 32, 32), (64, 64, 64, 4, 4), (4, 128, 64, 6, 20)])
def test_weight_gradient_kernel(device, batch, cin, cout, h, w):
    """sis_conv3x3_wgrad directly (the autograd wrapper only takes it above a work threshold)."""
    import sis_hip
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(batch, cin, h, w, generator=g).to(device)
    gy = torch.randn(batch, cout, h, w, generator=g).to(device)
    assert sis_hip.conv3x3_wgrad_supported(batch, cin, cout, h, w, min_work=0)
    got = sis_hip.conv3x3_wgrad(x, gy)
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, device=device, requires_grad=True)
    F.conv2d(x.double(), wt, padding=1).backward(gy.double())
    err = (got - wt.grad.float()).abs().max().item() / wt.grad.abs().max().item()
    assert err < 2e-4, err
    assert not sis_hip.conv3x3_wgrad_supported(batch, cin, cout, h, w + 1, min_work=0)  # odd width


def test_module_dispatch_and_fallback(device):
    from networks.hip_conv import HipConv2d
    import sis_hip
    conv = HipConv2d(64, 64, 3, 1, 1, bias=False).to(device)
    x = torch.randn(2, 64, 16, 16, device=device)
    rec = []
    sis_hip.set_profiler(rec)
    try:
        y = conv(x)
    finally:
        sis_hip.set_profiler(None)
    assert [r[0] for r in rec] == ["modconv_wino2_kernel"], "the 3x3 layer did not run on the HIP kernel"
    np.testing.assert_allclose(y.detach().cpu().numpy(), F.conv2d(x, conv.weight, padding=1).detach().cpu().numpy(),
                               atol=2e-5 * float(y.detach().abs().max()))
    # not eligible: dilation that does not divide the image, padding != dilation, stride 2 on an odd height, odd width -> ATen,
    # same result as nn.Conv2d
    for kwargs, shape in (({"dilation": 3, "padding": 3}, (2, 64, 32, 32)), ({"dilation": 2, "padding": 1}, (2, 64, 16, 16)),
                          ({"stride": 2, "padding": 1}, (2, 64, 15, 16)),
                          ({"padding": 1}, (2, 64, 16, 15))):
        c = HipConv2d(64, 32, 3, bias=False, **kwargs).to(device)
        xi = torch.randn(*shape, device=device)
        rec = []
        sis_hip.set_profiler(rec)
        try:
            out = c(xi)
        finally:
            sis_hip.set_profiler(None)
        assert rec == []
        assert torch.equal(out, F.conv2d(xi, c.weight, None, c.stride, c.padding, c.dilation))
    assert sorted(HipConv2d(8, 8, 3).state_dict().keys()) == ["bias", "weight"]


@pytest.mark.parametrize("b,cin,cout", [(3, 16, 24), (2, 512, 512)])
def test_half_image_dilation_is_the_dilated_convolution(device, b, cin, cout):
    """(2, 512, 512): EMANet-50's layer -- gather kernel + fp32 MFMA pointwise kernels in all three directions, nothing handed
    to the ROCm libraries; (3, 16, 24): a shape whose weight gradient has no tile plan (counted as a fallback)."""
    import sis_hip
    from networks.hip_conv import HipConv2d
    g = torch.Generator().manual_seed(16)
    x = torch.randn(b, cin, 32, 32, generator=g).to(device).requires_grad_(True)
    conv = HipConv2d(cin, cout, 3, 1, 16, 16, bias=False).to(device)
    gy = torch.randn(b, cout, 32, 32, generator=g).to(device)
    assert conv._half_image_dilation(x)
    sis_hip.library_calls(reset=True)
    y = conv(x)
    y.backward(gy)
    calls = sis_hip.library_calls(reset=True)
    if cin == 512:
        assert calls["fallback"] == {}, calls
    got = (y.detach(), x.grad.clone(), conv.weight.grad.clone())
    x.grad = conv.weight.grad = None
    ref = F.conv2d(x.double(), conv.weight.double(), padding=16, dilation=16)
    ref.backward(gy.double())
    for a, r in zip(got, (ref, x.grad, conv.weight.grad)):
        r = r.detach().float()
        assert (a - r).abs().max().item() < 1e-5 * r.abs().max().item()
    assert not conv._half_image_dilation(torch.zeros(1, cin, 64, 64, device=device))


def test_pointwise_convolution_is_a_batched_gemm(device):
    from networks.hip_conv import HipConv2d
    g = torch.Generator().manual_seed(5)
    for cin, cout, bias in ((64, 256, False), (256, 3, True)):
        conv = HipConv2d(cin, cout, 1, bias=bias).to(device)
        x = torch.randn(3, cin, 16, 12, generator=g).to(device).requires_grad_(True)
        gy = torch.randn(3, cout, 16, 12, generator=g).to(device)
        assert conv._pointwise(x)
        y = conv(x)
        y.backward(gy)
        got = [y.detach(), x.grad.clone(), conv.weight.grad.clone()] + ([conv.bias.grad.clone()] if bias else [])
        x.grad = conv.weight.grad = None
        if bias:
            conv.bias.grad = None
        ref = F.conv2d(x.double(), conv.weight.double(), conv.bias.double() if bias else None)
        ref.backward(gy.double())
        want = [ref, x.grad, conv.weight.grad] + ([conv.bias.grad] if bias else [])
        for a, b in zip(got, want):
            b = b.detach().float()
            assert (a - b).abs().max().item() < 2e-5 * b.abs().max().item()
    assert not HipConv2d(8, 8, 1, stride=2)._pointwise(torch.zeros(1, 8, 4, 4, device=device))


@pytest.mark.parametrize("batch,cin,cout,h,w,k", [(4, 128, 128, 64, 64, 3), (2, 64, 64, 34, 48, 3), (3, 256, 512, 64, 64, 1), (2, 64, 128, 17, 20, 1)])
def test_stride_2_layers_run_on_the_stride_1_kernels(device, batch, cin, cout, h, w, k):
    """EMANet's fp32 stride-2 convolutions (layer2: 3x3 padding 1 and the 1x1 shortcut) as HipConv2d dispatches them:
    dense kernel + even-pixel sampling.  Same tolerances as the stride-1 tests."""
    import sis_hip
    from networks.hip_conv import HipConv2d
    g = torch.Generator().manual_seed(batch + cin + h)
    conv = HipConv2d(cin, cout, k, 2, k // 2, bias=False).to(device)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5)
    x = torch.randn(batch, cin, h, w, generator=g).to(device).requires_grad_(True)
    assert conv._stride2(x) is not None
    y = conv(x)
    gy = torch.randn(y.shape, generator=g).to(device)
    y.backward(gy)
    gx, gw = x.grad.clone(), conv.weight.grad.clone()
    x.grad = conv.weight.grad = None
    ref = F.conv2d(x.double(), conv.weight.double(), stride=2, padding=k // 2)
    assert ref.shape == y.shape
    ref.backward(gy.double())
    for got, want, name in ((y, ref, "y"), (gx, x.grad, "dx"), (gw, conv.weight.grad, "dw")):
        want = want.detach().float()
        err = (got.detach() - want).abs().max().item() / want.abs().max().item()
        assert err < (2e-5 if name != "dw" else 2e-4), (name, err)


def test_bottleneck_shortcut_gradient_is_summed_in_the_data_gradient_kernel(device, monkeypatch):
    """EMANet bottleneck with an identity shortcut (networks/ema_net/network.py:37-56): the first 1x1 convolution's data
    gradient adds the shortcut's gradient in its epilogue (csrc/conv1x1_f32.hip, ``sis_conv1x1_f32_dgrad_add``).  Output and
    every gradient must equal the unfused block (autograd's separate sum) -- same kernels, the sum is the only difference, so
    the input gradient agrees to fp32 round-off of one addition and everything else bitwise."""
    import networks.hip_conv as hc
    from networks.ema_net.network import Bottleneck
    torch.manual_seed(3)
    block = Bottleneck(256, 64).to(device).train()
    x = torch.randn(4, 256, 32, 32, device=device)
    gy = torch.randn(4, 256, 32, 32, device=device)

    def run(fuse):
        monkeypatch.setattr(hc, "_FUSE_SKIP_GRAD", fuse)
        for p in block.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y = block(xi)
        y.backward(gy)
        return y.detach(), xi.grad, [p.grad.clone() for p in block.parameters()]

    y0, dx0, g0 = run(False)
    y1, dx1, g1 = run(True)
    assert torch.equal(y0, y1)
    assert (dx0 - dx1).abs().max().item() <= 1e-6 * dx0.abs().max().item()
    for a, b in zip(g0, g1):
        assert torch.equal(a, b)
    # and against plain torch modules in fp64 on the CPU (the reference's own composition)
    import copy
    ref = copy.deepcopy(block).cpu().double()
    xr = x.cpu().double().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gy.cpu().double())
    assert (dx1.cpu().double() - xr.grad).abs().max().item() <= 2e-4 * xr.grad.abs().max().item()


def test_winograd_pack_bank_writes_the_images_of_the_single_layer_prepack(device):
    """``sis_hip.WinogradPackBank`` (every 3x3 layer of EMANet, one launch per step): each layer's forward / adjoint image is bit
    for bit what ``conv3x3_prepack_both`` writes for that layer alone, also after the weights changed in place."""
    import sis_hip
    gen = torch.Generator().manual_seed(11)
    weights = [torch.randn(co, ci, 3, 3, generator=gen).to(device) for co, ci in ((64, 3), (64, 64), (128, 64), (512, 2048), (72, 40))]
    bank = sis_hip.WinogradPackBank(weights)
    for _ in range(2):
        bank.refresh()
        for w, u, ua in zip(weights, bank.u, bank.u_adjoint):
            ref_u, ref_ua = sis_hip.conv3x3_prepack_both(w)
            assert torch.equal(u, ref_u) and torch.equal(ua, ref_ua)
        weights[1].mul_(-0.5)
        weights[3].add_(1.0)
    assert bank.current()


@pytest.mark.parametrize("kind,jobs,batch,cin,cout,h,w", [("f3", 5, 16, 256, 256, 16, 16), ("f3", 3, 4, 128, 128, 32, 32), ("f3", 18, 2, 64, 64, 16, 16),
                                                          ("f1", 5, 8, 1024, 256, 32, 32), ("f1", 3, 2, 128, 512, 64, 64), ("f1", 17, 1, 64, 256, 16, 16),
                                                          ("f1", 2, 4, 2048, 512, 16, 16)])
def test_batched_fp32_weight_gradients_of_one_shape(device, kind, jobs, batch, cin, cout, h, w):
    """EMANet's repeated bottleneck units: several fp32 layers of ONE shape through one tile launch + one finish / reduction launch
    (sis_conv3x3_wgrad_multi / sis_conv1x1_wgrad_f32_multi, what ``sis_hip.flush_deferred`` runs).  Every layer's dW against its own
    single-layer call: same products, K slices cut for the joint tile count -> another association of the fp32 partial sums
    (1e-4 of the largest entry); more layers than one launch takes; bitwise repeatable."""
    import sis_hip
    gen = torch.Generator().manual_seed(len(kind) * 100 + jobs * 10 + cin)
    xs = [torch.randn(batch, cin, h, w, generator=gen).to(device) for _ in range(jobs)]
    gys = [torch.randn(batch, cout, h, w, generator=gen).to(device) for _ in range(jobs)]
    if kind == "f3":
        assert sis_hip.conv3x3_wgrad_supported(batch, cin, cout, h, w, min_work=0)
        ref = [sis_hip.conv3x3_wgrad(x, gy) for x, gy in zip(xs, gys)]
        dims = (batch, cin, cout, h, w)
    else:
        assert sis_hip.conv1x1_wgrad_f32_supported(gys[0], xs[0])
        ref = [sis_hip.conv1x1_wgrad_f32(gy, x) for x, gy in zip(xs, gys)]
        dims = (batch, cin, cout, h * w)
    runs = []
    for _ in range(2):
        dws = [torch.empty_like(r) for r in ref]
        for x, gy, dw in zip(xs, gys, dws):
            sis_hip._defer_conv_wgrad(kind, x, gy, dw, dims)
        sis_hip.flush_deferred()
        assert sis_hip.deferred_pending() == 0
        runs.append(dws)
    for j in range(jobs):
        scale = float(ref[j].abs().max())
        assert float((runs[0][j] - ref[j]).abs().max()) <= 1e-4 * scale, j
        assert torch.equal(runs[0][j], runs[1][j])
