"""GPU parity: the fp32 pointwise convolution (csrc/conv1x1_f32.hip) through the C ABI against ``F.conv2d`` in float64.
The MFMA is an exact fp32 fma chain in k order: |err| <= 2e-6 * sum|a b| (stated: 1e-5 * max|ref| on unit-variance data)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [(2, 64, 256, 64, 64, False), (16, 256, 64, 64, 64, False), (3, 512, 128, 32, 32, False), (2, 1024, 256, 32, 32, False),
         (2, 256, 1024, 32, 32, False), (1, 2048, 512, 32, 32, True), (2, 64, 64, 20, 12, True), (1, 128, 160, 8, 10, False),
         (4, 512, 2048, 16, 16, False)]


@pytest.mark.parametrize("batch,cin,cout,h,w,bias", CASES)
def test_conv1x1_f32_forward_and_data_gradient(device, batch, cin, cout, h, w, bias):
    import sis_hip
    gen = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(batch, cin, h, w, generator=gen).to(device)
    wt = (torch.randn(cout, cin, 1, 1, generator=gen) / cin ** 0.5).to(device)
    b = torch.randn(cout, generator=gen).to(device) if bias else None
    gy = torch.randn(batch, cout, h, w, generator=gen).to(device)
    assert sis_hip.conv1x1_f32_supported(x, wt)
    xr = x.double().requires_grad_(True)
    ref = F.conv2d(xr, wt.double(), b.double() if bias else None)
    ref.backward(gy.double())
    y = sis_hip.conv1x1_f32(x, wt, b)
    assert (y.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    gx = sis_hip.conv1x1_f32(gy, wt, data_gradient=True)
    assert (gx.double() - xr.grad).abs().max().item() <= 1e-5 * xr.grad.abs().max().item()


def test_pointwise_autograd_path(device):
    """HipConv2d 1x1 (what EMANet's bottlenecks call): forward / data gradient on the kernel, weight gradient as the
    batched GEMM, against autograd of the library convolution."""
    from networks.hip_conv import HipConv2d
    torch.manual_seed(0)
    conv = HipConv2d(256, 128, 1, bias=False).to(device)
    x = torch.randn(4, 256, 32, 32, device=device, requires_grad=True)
    gy = torch.randn(4, 128, 32, 32, device=device)
    conv(x).backward(gy)
    xr = x.detach().clone().requires_grad_(True)
    wr = conv.weight.detach().clone().requires_grad_(True)
    F.conv2d(xr, wr).backward(gy)
    for got, ref in ((x.grad, xr.grad), (conv.weight.grad, wr.grad)):
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("batch,cin,cout,h,w", [(16, 64, 256, 64, 64), (16, 256, 64, 64, 64), (16, 512, 2048, 32, 32), (16, 2048, 512, 32, 32),
                                                (3, 128, 128, 8, 8), (2, 32, 256, 16, 16), (5, 192, 384, 8, 16), (2, 1024, 1024, 8, 8), (1, 64, 128, 8, 8)])
def test_weight_gradient_kernel(device, batch, cin, cout, h, w):
    """dW = sum_b dy_b x_b^T on the fp32 matrix cores against the fp64 product; split-K slices are added in a fixed order:
    two runs are bit-equal."""
    import sis_hip
    g = torch.Generator().manual_seed(batch * 7 + cin + cout)
    x = torch.randn(batch, cin, h, w, generator=g).to(device)
    gy = torch.randn(batch, cout, h, w, generator=g).to(device)
    assert sis_hip.conv1x1_wgrad_f32_supported(gy, x)
    dw = sis_hip.conv1x1_wgrad_f32(gy, x)
    ref = torch.einsum("bop,bip->oi", gy.double().flatten(2), x.double().flatten(2)).view(cout, cin, 1, 1)
    err = ((dw.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err
    assert torch.equal(dw, sis_hip.conv1x1_wgrad_f32(gy, x))
    assert not sis_hip.conv1x1_wgrad_f32_supported(gy[:, :, :1, :7].contiguous(), x[:, :, :1, :7].contiguous())  # pixels % 64
