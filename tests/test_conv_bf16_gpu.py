"""GPU parity: the bf16 MFMA convolutions (csrc/conv_bf16.hip) through the C ABI against fp32 ``F.conv2d`` on the same
bf16-rounded operands.  Stated tolerance: the result is rounded to bf16 once (2^-9 relative) after an fp32 accumulation
whose association differs from the library's -> |err| <= 1e-2 * max|ref| element-wise."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # batch, cin, cout, h, w, k, stride, bias
    (2, 64, 64, 32, 32, 3, 1, False),      # MT 64, tile 32 wide
    (2, 128, 256, 64, 64, 3, 1, False),    # MT 128, tile 64 wide
    (1, 256, 128, 24, 40, 3, 1, False),    # ragged tiles in both directions
    (2, 64, 16, 64, 64, 3, 1, False),      # MT 32
    (1, 16, 3, 64, 64, 3, 1, True),        # segmentation head: 3 output channels + bias
    (2, 192, 64, 48, 48, 3, 1, False),     # concatenated skip (192 channels)
    (1, 768, 512, 32, 32, 3, 1, False),    # decoder conv_more
    (2, 256, 1024, 16, 16, 1, 1, False),   # pointwise, flat pixel row
    (1, 1024, 768, 32, 32, 1, 1, True),    # patch embedding (bias)
    (2, 64, 256, 20, 28, 1, 1, False),     # pointwise, ragged tail tile
    (1, 64, 64, 127, 127, 1, 1, False),    # odd planes: unaligned staging path
    (1, 64, 64, 127, 127, 3, 1, False),
    (2, 128, 128, 63, 63, 3, 2, False),    # stride 2 (block2 / block3 entry)
    (1, 256, 512, 63, 63, 1, 2, False),
    (1, 128, 128, 127, 127, 3, 2, False),
]


def _ref(x, w, b, k, s):
    """fp32 convolution of the bf16-rounded operands.  Small layers: on the CPU (an oracle independent of this device and
    its libraries); layers above ~2 GFLOP: the device's fp32 library convolution (still independent of the kernels under test)."""
    flops = 2.0 * x.shape[0] * w.shape[0] * w.shape[1] * k * k * x.shape[2] * x.shape[3] / (s * s)
    if flops <= 2e9:
        out = F.conv2d(x.float().cpu(), w.float().cpu(), None if b is None else b.cpu(), stride=s, padding=k // 2)
        return out.to(x.device)
    return F.conv2d(x.float(), w.float(), b, stride=s, padding=k // 2)


@pytest.mark.parametrize("batch,cin,cout,h,w,k,stride,bias", CASES)
def test_conv_bf16_forward(device, batch, cin, cout, h, w, k, stride, bias):
    import sis_hip
    assert sis_hip.conv_bf16_supported(cin, cout, h, w, k, stride)
    gen = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(batch, cin, h, w, generator=gen).to(device).bfloat16()
    wt = (torch.randn(cout, cin, k, k, generator=gen) / (cin * k * k) ** 0.5).to(device)
    b = torch.randn(cout, generator=gen).to(device) if bias else None
    for weight in (wt, wt.bfloat16()):  # fp32 master weights and the bf16 output of the weight standardisation
        y = sis_hip.conv_bf16(x, sis_hip.conv_bf16_pack(weight, h, w, stride), cout, k, stride, b)
        ref = _ref(x, wt.bfloat16(), b, k, stride)
        assert y.dtype == torch.bfloat16 and y.shape == ref.shape
        err = (y.float() - ref).abs().max().item()
        assert err <= 1e-2 * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize("batch,cin,cout,h,w,k", [(2, 64, 128, 32, 32, 3), (1, 192, 64, 48, 48, 3), (2, 256, 64, 16, 16, 1),
                                                  (1, 64, 256, 127, 127, 1), (1, 128, 128, 24, 40, 3)])
def test_conv_bf16_data_gradient(device, batch, cin, cout, h, w, k):
    """dL/dx = the same kernel on adjoint-packed weights, against autograd of the fp32 convolution."""
    import sis_hip
    gen = torch.Generator().manual_seed(cin * 3 + cout)
    wt = (torch.randn(cout, cin, k, k, generator=gen) / (cout * k * k) ** 0.5).to(device).bfloat16()
    gy = torch.randn(batch, cout, h, w, generator=gen).to(device).bfloat16()
    x = torch.zeros(batch, cin, h, w, device=device, requires_grad=True)
    F.conv2d(x, wt.float(), padding=k // 2).backward(gy.float())
    assert sis_hip.conv_bf16_supported(cout, cin, h, w, k, 1)
    gx = sis_hip.conv_bf16(gy, sis_hip.conv_bf16_pack(wt, h, w, 1, adjoint=True), cin, k, 1)
    err = (gx.float() - x.grad).abs().max().item()
    assert err <= 1e-2 * x.grad.abs().max().item(), (err, x.grad.abs().max().item())


@pytest.mark.parametrize("batch,cin,cout,h,w", [(2, 64, 128, 32, 32), (2, 128, 64, 64, 64), (1, 192, 64, 48, 48), (3, 256, 256, 16, 16),
                                                (1, 64, 64, 127, 127), (2, 96, 160, 24, 40), (8, 64, 64, 128, 128), (2, 64, 64, 32, 32),
                                                (2, 64, 16, 64, 64), (2, 16, 16, 48, 80), (1, 16, 3, 64, 200), (2, 128, 24, 40, 40),
                                                (1, 64, 16, 16, 256), (1, 16, 16, 8, 256), (1, 16, 3, 8, 512), (1, 48, 16, 12, 136)])   # (wide maps: the double-width strips of the one-block layers)
def test_conv_bf16_weight_gradient(device, batch, cin, cout, h, w):
    """dL/dw on the pixel-contraction kernel (csrc/conv_bf16_wgrad.hip) against autograd of the fp32 convolution on the
    same bf16-rounded tensors; fp32 result: |err| <= 2e-3 * max|ref| (fp32 accumulation in a different order), bf16
    result: 1e-2."""
    import sis_hip
    gen = torch.Generator().manual_seed(cin + 7 * cout + h)
    x = torch.randn(batch, cin, h, w, generator=gen).to(device).bfloat16()
    gy = torch.randn(batch, cout, h, w, generator=gen).to(device).bfloat16()
    wt = torch.zeros(cout, cin, 3, 3, device=device, requires_grad=True)
    F.conv2d(x.float(), wt, padding=1).backward(gy.float())
    assert sis_hip.conv_bf16_wgrad_supported(batch, cin, cout, h, w)
    for dtype, tol in ((torch.float32, 2e-3), (torch.bfloat16, 1e-2)):
        dw = sis_hip.conv_bf16_wgrad(x, gy, dtype)
        assert dw.dtype == dtype and dw.shape == wt.shape
        err = (dw.float() - wt.grad).abs().max().item()
        assert err <= tol * wt.grad.abs().max().item(), (dtype, err, wt.grad.abs().max().item())
    again = sis_hip.conv_bf16_wgrad(x, gy, torch.float32)
    assert torch.equal(again, sis_hip.conv_bf16_wgrad(x, gy, torch.float32))  # ordered slab reduction: bitwise reproducible


@pytest.mark.parametrize("batch,cin,cout,h,w", [(8, 64, 256, 32, 32), (2, 256, 64, 127, 127), (1, 64, 64, 127, 127), (3, 1024, 256, 16, 16),
                                                (2, 128, 512, 64, 64), (2, 72, 40, 20, 28), (1, 256, 1024, 32, 32), (2, 512, 128, 9, 7),
                                                (8, 256, 64, 127, 127)])
def test_conv1x1_bf16_weight_gradient(device, batch, cin, cout, h, w):
    """dL/dw of the 1x1 layers (conv1x1_wgrad_bf16_kernel: all four wave layouts, aligned and odd planes, channel counts that
    are not tile multiples, planes shorter than a stage) against the float64 contraction of the same bf16-rounded tensors on
    the CPU; fp32 result |err| <= 2e-3 * max|ref| (fp32 accumulation), bf16 result 1e-2; bitwise repeatable."""
    import sis_hip
    gen = torch.Generator().manual_seed(cin + 3 * cout + h)
    x = torch.randn(batch, cin, h, w, generator=gen).to(device).bfloat16()
    gy = torch.randn(batch, cout, h, w, generator=gen).to(device).bfloat16()
    ref = torch.einsum("bop,bip->oi", gy.cpu().double().flatten(2), x.cpu().double().flatten(2)).float().to(device)
    assert sis_hip.conv1x1_bf16_wgrad_supported(batch, cin, cout, h * w)
    for dtype, tol in ((torch.float32, 2e-3), (torch.bfloat16, 1e-2)):
        dw = sis_hip.conv1x1_bf16_wgrad(x, gy, dtype)
        assert dw.dtype == dtype and dw.shape == (cout, cin, 1, 1)
        err = (dw.float().view(cout, cin) - ref).abs().max().item()
        assert err <= tol * ref.abs().max().item(), (dtype, err, ref.abs().max().item())
    assert torch.equal(sis_hip.conv1x1_bf16_wgrad(x, gy), sis_hip.conv1x1_bf16_wgrad(x, gy))


def test_conv_bf16_autograd_function(device):
    """The autograd wrapper the networks call: forward, data and weight gradients, bias gradient in one graph."""
    from networks.hip_conv import conv_bf16, conv_bf16_applicable
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 32, 32, generator=gen).to(device).bfloat16().requires_grad_(True)
    w = (torch.randn(128, 64, 3, 3, generator=gen) / 24).to(device).requires_grad_(True)
    b = torch.randn(128, generator=gen).to(device).requires_grad_(True)
    assert conv_bf16_applicable(x, w, (1, 1), (1, 1), (1, 1), 1)
    y = conv_bf16(x, w, b, 1)
    gy = torch.randn(y.shape, generator=gen).to(device).bfloat16()
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().bfloat16().float().requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    F.conv2d(xr, wr, br, padding=1).backward(gy.float())
    for got, ref, tol in ((x.grad, xr.grad, 1e-2), (w.grad, wr.grad, 2e-3), (b.grad, br.grad, 1e-3)):
        assert (got.float() - ref).abs().max().item() <= tol * ref.abs().max().item()
    assert w.grad.dtype == torch.float32 and x.grad.dtype == torch.bfloat16


@pytest.mark.parametrize("cin,cout,h,w,k", [(64, 128, 32, 32, 3), (128, 128, 31, 31, 3), (256, 512, 32, 32, 1), (64, 256, 63, 63, 1)])
def test_stride_2_backward_on_the_stride_1_kernels(device, cin, cout, h, w, k):
    """Stride-2 layers of TransUNet's ResNetV2 (3x3 in the bottlenecks, 1x1 projection shortcuts; odd map sizes occur):
    forward on the strided kernel, both gradients on the stride-1 kernels (zero-stuffed dL/dy, or sampled input)."""
    from networks.hip_conv import conv_bf16, conv_bf16_applicable
    gen = torch.Generator().manual_seed(cin + h + k)
    x = torch.randn(2, cin, h, w, generator=gen).to(device).bfloat16().requires_grad_(True)
    wt = (torch.randn(cout, cin, k, k, generator=gen) * (cin * k * k) ** -0.5).to(device).requires_grad_(True)
    assert conv_bf16_applicable(x, wt, (2, 2), (k // 2, k // 2), (1, 1), 1)
    y = conv_bf16(x, wt, None, 2)
    gy = torch.randn(y.shape, generator=gen).to(device).bfloat16()
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = wt.detach().bfloat16().float().requires_grad_(True)
    ref = F.conv2d(xr, wr, None, stride=2, padding=k // 2)
    assert ref.shape == y.shape
    ref.backward(gy.float())
    assert (y.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()
    # (1x1: the per-sample products dy_b x_b^T leave the library GEMM in bf16 before the fp32 sum over the batch: 2^-8 relative)
    for got, want, tol in ((x.grad, xr.grad, 1e-2), (wt.grad, wr.grad, 2e-3 if k == 3 else 5e-3)):
        assert got.shape == want.shape
        assert (got.float() - want).abs().max().item() <= tol * want.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,h,w", [(16, 3, 64, 64), (16, 9, 40, 56), (32, 2, 32, 32)])
def test_narrow_output_layer_backward(device, cin, cout, h, w):
    """The segmentation head (3x3, Cout = number of classes): its gradients contract over Cout, which the kernels take in chunks
    of 16, so dL/dy and the filters get zero channels appended; the gradients must not change (reference:
    stylegan_code_finder/networks/trans_u_net/vit_seg_modeling.py:324-330, SegmentationHead)."""
    import sis_hip
    from networks.hip_conv import conv_bf16, conv_bf16_applicable
    gen = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(2, cin, h, w, generator=gen).to(device).bfloat16().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=gen) * (cin * 9) ** -0.5).to(device).requires_grad_(True)
    bias = torch.randn(cout, generator=gen).to(device).requires_grad_(True)
    assert conv_bf16_applicable(x, wt, (1, 1), (1, 1), (1, 1), 1)
    assert not sis_hip.conv_bf16_supported(cout, cin, h, w, 3, 1)
    y = conv_bf16(x, wt, bias, 1)
    gy = torch.randn(y.shape, generator=gen).to(device).bfloat16()
    y.backward(gy)
    xr = x.detach().float().requires_grad_(True)
    wr = wt.detach().bfloat16().float().requires_grad_(True)
    br = bias.detach().clone().requires_grad_(True)
    ref = F.conv2d(xr, wr, br, padding=1)
    ref.backward(gy.float())
    for got, want, tol in ((x.grad, xr.grad, 1e-2), (wt.grad, wr.grad, 2e-3), (bias.grad, br.grad, 1e-3)):
        assert got.shape == want.shape
        assert (got.float() - want).abs().max().item() <= tol * want.abs().max().item()


def test_plain_pack_bank_writes_the_images_of_the_single_layer_pack(device):
    """``WeightStdPackBank(standardize=False)`` (the decoder's and the head's weights, one launch per forward): every layer's
    forward / adjoint image is bit for bit what ``conv_bf16_pack`` writes for that layer alone; layers without an adjoint plan
    (3 output channels) get none."""
    import sis_hip
    gen = torch.Generator().manual_seed(3)
    shapes = [(512, 768, 3), (256, 1024, 3), (64, 64, 3), (16, 64, 3), (3, 16, 3), (96, 128, 1)]
    weights = [torch.randn(co, ci, k, k, generator=gen).to(device) for co, ci, k in shapes]
    assert all(sis_hip.WeightStdPackBank.supported(w, 1) for w in weights)
    bank = sis_hip.WeightStdPackBank(weights, [1] * len(weights), 0.0, standardize=False)
    bank.refresh()
    assert all(t is None for t in bank.w_hat) and all(t is None for t in bank.invstd)
    for w, packed, adjoint in zip(weights, bank.packed, bank.adjoint):
        assert torch.equal(packed, sis_hip.conv_bf16_pack(w, 64, 64, 1))
        if w.shape[0] % (16 if w.shape[2] == 3 else 64) == 0:   # the adjoint contracts over Cout in chunks of 16 (3x3) / 64 (1x1)
            assert adjoint is not None and torch.equal(adjoint, sis_hip.conv_bf16_pack(w, 64, 64, 1, adjoint=True))
        else:
            assert adjoint is None
    weights[2].mul_(2.0)
    bank.refresh()   # same buffers, new contents
    assert torch.equal(bank.packed[2], sis_hip.conv_bf16_pack(weights[2], 64, 64, 1))


@pytest.mark.parametrize("kind,jobs,batch,cin,cout,h,w", [(3, 8, 8, 256, 256, 32, 32), (3, 3, 2, 128, 128, 64, 64), (3, 20, 1, 64, 64, 16, 16),
                                                          (3, 2, 1, 64, 64, 127, 127), (1, 8, 8, 1024, 256, 32, 32), (1, 5, 2, 256, 1024, 32, 32),
                                                          (1, 2, 1, 64, 256, 127, 127), (1, 18, 2, 72, 40, 20, 28)])
def test_batched_weight_gradients_of_one_shape(device, kind, jobs, batch, cin, cout, h, w):
    """Several layers of ONE shape through one tile launch + one reduction launch (sis_conv_bf16_wgrad_multi /
    sis_conv1x1_bf16_wgrad_multi, what ``sis_hip.flush_deferred`` runs for the trunk's repeated bottleneck units): every layer's
    dW against its own single-layer call (same products; the joint tile plan cuts a layer into other units, so the fp32 partial
    sums associate differently: 2e-3 of the largest entry for fp32 results, 1e-2 for bf16 ones), more layers than one launch
    takes (16), unaligned planes, and bitwise repeatability of the batched form."""
    import sis_hip
    gen = torch.Generator().manual_seed(kind * 1000 + jobs * 10 + cin)
    xs = [torch.randn(batch, cin, h, w, generator=gen).to(device).bfloat16() for _ in range(jobs)]
    gys = [torch.randn(batch, cout, h, w, generator=gen).to(device).bfloat16() for _ in range(jobs)]
    single = sis_hip.conv_bf16_wgrad if kind == 3 else sis_hip.conv1x1_bf16_wgrad
    dims = (batch, cin, cout, h, w) if kind == 3 else (batch, cin, cout, h * w)
    for dtype, tol in ((torch.float32, 2e-3), (torch.bfloat16, 1e-2)):
        ref = [single(x, gy, dtype) for x, gy in zip(xs, gys)]
        runs = []
        for _ in range(2):
            dws = [torch.empty_like(r) for r in ref]
            for x, gy, dw in zip(xs, gys, dws):
                sis_hip._defer_conv_wgrad(kind, x, gy, dw, dims)
            assert sis_hip.deferred_pending() == jobs
            sis_hip.flush_deferred()
            assert sis_hip.deferred_pending() == 0
            runs.append(dws)
        for j in range(jobs):
            scale = float(ref[j].float().abs().max())
            assert float((runs[0][j].float() - ref[j].float()).abs().max()) <= tol * scale, (j, dtype)
            assert torch.equal(runs[0][j], runs[1][j])


@pytest.mark.parametrize("image_dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("batch,h,w", [(2, 64, 64), (1, 224, 224), (2, 50, 70), (1, 37, 129), (3, 128, 96)])
def test_stem_conv_7x7_stride_2_on_the_image(device, batch, h, w, image_dtype):
    """csrc/stem_conv.hip: ResNetV2's root convolution (3 -> 64, 7x7, stride 2, padding 3) forward and weight gradient on the matrix
    cores against F.conv2d / autograd in fp32 on the same bf16-rounded tensors: output within bf16 rounding (1e-2 of the largest
    entry), dW 2e-3 (fp32) / 1e-2 (bf16); output widths that are no multiple of the 32-pixel tile or the 16-pixel K-step, odd
    sizes (unaligned dL/dy rows), float32 images converted while they are staged; bitwise repeatable."""
    import sis_hip
    gen = torch.Generator().manual_seed(h * 3 + w)
    x = torch.randn(batch, 3, h, w, generator=gen).to(device).to(image_dtype)
    wt = (torch.randn(64, 3, 7, 7, generator=gen) / 147 ** 0.5).to(device).bfloat16()
    assert sis_hip.stem_conv_supported(x, wt, 2, 3)
    wr = wt.float().requires_grad_(True)
    ref = F.conv2d(x.bfloat16().float(), wr, stride=2, padding=3)
    y = sis_hip.stem_conv_fwd(x, wt)
    assert y.dtype == torch.bfloat16 and y.shape == ref.shape
    assert float((y.float() - ref.detach()).abs().max()) <= 1e-2 * float(ref.detach().abs().max())
    gy = torch.randn(*ref.shape, generator=gen).to(device).bfloat16()
    ref.backward(gy.float())
    for dtype, tol in ((torch.float32, 2e-3), (torch.bfloat16, 1e-2)):
        dw = sis_hip.stem_conv_wgrad(x, gy, dtype)
        assert dw.dtype == dtype and dw.shape == wt.shape
        assert float((dw.float() - wr.grad).abs().max()) <= tol * float(wr.grad.abs().max()), dtype
    assert torch.equal(sis_hip.stem_conv_wgrad(x, gy), sis_hip.stem_conv_wgrad(x, gy))
    assert not sis_hip.stem_conv_supported(torch.empty(1, 4, 64, 64, device=device), wt, 2, 3)
    assert not sis_hip.stem_conv_supported(x, wt, 1, 3)
