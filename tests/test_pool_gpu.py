"""GPU parity: max pooling (csrc/pool_ops.hip) against ATen -- bit-equal in both directions, ties included (the maps it
sees in the networks are post-ReLU: many equal zeros), for the two call sites' geometries and odd sizes."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 5, 16, 16), (3, 4, 17, 23), (1, 2, 7, 9), (2, 64, 128, 128), (1, 3, 3, 3)])
@pytest.mark.parametrize("geom", [(3, 2, 1), (3, 2, 0), (2, 2, 0), (3, 1, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_max_pool_equals_aten(device, shape, geom, dtype):
    from networks.hip_pool import max_pool2d
    k, s, p = geom
    g = torch.Generator().manual_seed(sum(shape) + k * 10 + s)
    x = torch.relu(torch.randn(*shape, generator=g)).mul(4).round().div(4)  # coarse values: ties in most windows
    x[0, 0, 0, 0] = float("nan")
    x = x.to(device=device, dtype=dtype).requires_grad_(True)
    ref_in = x.detach().clone().requires_grad_(True)
    ref = F.max_pool2d(ref_in, k, s, p)
    got = max_pool2d(x, k, s, p)
    assert got.shape == ref.shape
    assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(ref, nan=-7.0))
    gy = torch.randn(ref.shape, generator=g).to(device=device, dtype=dtype)
    ref.backward(gy)
    got.backward(gy)
    assert torch.equal(x.grad, ref_in.grad)


def test_module_and_fallbacks(device):
    from networks.hip_pool import HipMaxPool2d
    m = HipMaxPool2d(kernel_size=3, stride=2, padding=1)
    x = torch.randn(2, 3, 10, 12)
    assert torch.equal(m(x), F.max_pool2d(x, 3, 2, 1))                       # CPU tensor: ATen
    xd = x.to(device)
    assert torch.equal(m(xd).cpu(), F.max_pool2d(x, 3, 2, 1))
    assert torch.equal(HipMaxPool2d(3, 2, 1, dilation=2)(xd).cpu(), F.max_pool2d(x, 3, 2, 1, dilation=2))  # dilation: ATen
