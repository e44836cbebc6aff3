"""GPU: bitwise repeatability of the convolution / GEMM families that accumulate over several workgroups or stage through
memory the allocator recycles (the loop of tools/check_conv_determinism.py as a test).  Between repeats a NaN-filled block of
varying size is allocated and freed, so a kernel that reads memory it never wrote -- or sums partials in arrival order --
shows up as a differing or NaN element.  Families: fp32 Winograd forward (fused and plain tail), transposed convolution,
plain 3x3 forward / data gradient / weight gradient (networks/hip_conv.py), fp32 1x1 forward / data / weight gradient,
bf16 GEMM split-K, attention backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _repeat(fn, n=8):
    dev = torch.device("cuda")
    first = [t.clone() for t in fn()]
    for i in range(n):
        junk = torch.full((1 << (18 + i % 5),), float("nan"), device=dev)   # freed blocks hold NaNs
        got = fn()
        del junk
        for a, b in zip(got, first):
            assert not torch.isnan(a).any(), "NaN: a read of memory the kernel never wrote"
            assert torch.equal(a, b), f"{int((a != b).sum())} elements differ between two runs on the same inputs"


@pytest.mark.parametrize("b,cin,cout,h", [(5, 64, 64, 16), (4, 512, 512, 8), (8, 512, 512, 32), (3, 128, 64, 32), (32, 128, 128, 256)])
def test_modulated_convolutions_repeat_bitwise(device, b, cin, cout, h):
    import sis_hip
    gen = torch.Generator().manual_seed(b + cin + h)
    x = torch.randn(b, cin, h, h, generator=gen).to(device)
    w = torch.randn(1, cout, cin, 3, 3, generator=gen).to(device)
    s = (1 + 0.1 * torch.randn(b, cin, generator=gen)).to(device)
    wpk, wsq = sis_hip.modconv_prepack(w)
    ds = sis_hip.modconv_demod(s, wsq, 1 / (cin * 9) ** 0.5, True)
    noise, nw, bias = torch.randn(1, 1, h, h, generator=gen).to(device), torch.full((1,), 0.1, device=device), torch.randn(cout, generator=gen).to(device)
    u = sis_hip.modconv_prepack_wino(w)
    _repeat(lambda: [sis_hip.modconv2d(x, wpk, s, ds, 3, noise, nw, bias, fuse_act=True, wino_u=u),
                     sis_hip.modconv2d(x, wpk, s, ds, 3, wino_u=u), sis_hip.modconv2d(x, wpk, s, ds, 3)], n=4 if h == 256 else 8)
    if h <= 64:
        _repeat(lambda: [sis_hip.modconv2d_up(x, wpk, s, ds, padded_rows=True)[..., :2 * h + 1]])


@pytest.mark.parametrize("b,cin,cout,h", [(4, 64, 64, 32), (16, 128, 128, 32), (2, 256, 512, 16)])
def test_plain_convolution_family_repeats_bitwise(device, b, cin, cout, h):
    """fp32 3x3 (Winograd forward, adjoint data gradient, Winograd-domain weight gradient) and 1x1 (forward, data gradient,
    split-K weight gradient) through the autograd functions the segmentation networks use."""
    from networks.hip_conv import HipConv2d
    gen = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(b, cin, h, h, generator=gen).to(device).requires_grad_(True)
    for k in (3, 1):
        conv = HipConv2d(cin, cout, k, padding=k // 2, bias=False).to(device)
        gy = torch.randn(b, cout, h, h, generator=gen).to(device)

        def fwd_bwd():
            x.grad = None
            conv.weight.grad = None
            y = conv(x)
            y.backward(gy)
            return [y.detach(), x.grad, conv.weight.grad]
        _repeat(fwd_bwd, n=6)


def test_vit_kernels_repeat_bitwise(device):
    import sis_hip as S
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(2048, 768, generator=gen).bfloat16().to(device)
    x = torch.randn(2048, 3072, generator=gen).bfloat16().to(device)
    _repeat(lambda: [S.gemm_bf16(g, x, S.GEMM_TN, S.EPI_F32, splits=4, tile=4), S.gemm_bf16(g, x, S.GEMM_TN, S.EPI_F32, splits=8)])
    qkv = torch.randn(2, 320, 2304, generator=gen).bfloat16().to(device)
    d_ctx = torch.randn(2, 320, 768, generator=gen).bfloat16().to(device)
    ctx, lse = S.attention_fwd(qkv, 12)
    _repeat(lambda: [S.attention_bwd(d_ctx, qkv, ctx, lse, 12), *S.attention_fwd(qkv, 12)])
