"""GPU parity: the bf16 GEMM with fused epilogues (csrc/gemm_bf16.hip) through the C ABI against fp32 CPU arithmetic on the same
bf16-rounded operands (the oracle for the ViT encoder's Linear layers is plain ``x @ W.T + b`` etc.,
networks/trans_u_net/vit_seg_modeling.py:60-67,76-96,104-122,181-189).

Stated tolerances: fp32 accumulation in a different association than the CPU's, then ONE rounding to bf16 (2^-9 relative) for
the bf16 outputs -> |err| <= 1e-2 * max|ref|; fp32 outputs (residual stream, weight gradients) -> |err| <= 2e-4 * max|ref|
(K up to 8192 products of bf16 values accumulated in fp32)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16_TOL, F32_TOL = 1e-2, 2e-4


def _rand(shape, gen, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).bfloat16()


def _close(got, ref, tol):
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item(), (err, ref.abs().max().item())


def _gelu(x):
    return 0.5 * x * (1 + torch.erf(x / math.sqrt(2)))


def _gelu_grad(x):
    return 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


NT_CASES = [(256, 256, 128), (128, 384, 64), (392, 768, 768), (1000, 2304, 768), (512, 768, 3072), (130, 132, 192)]


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("m,n,k", NT_CASES)
def test_gemm_nt_epilogues(device, m, n, k, tile):
    import sis_hip as S
    gen = torch.Generator().manual_seed(m * 7 + n * 3 + k + tile)
    x, w = _rand((m, k), gen), _rand((n, k), gen, k ** -0.5)
    bias = torch.randn(n, generator=gen)
    resid = torch.randn(m, n, generator=gen)
    xd, wd, bd, rd = x.to(device), w.to(device), bias.to(device), resid.to(device)
    y = x.float() @ w.float().t()
    _close(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_NONE, tile=tile), y, BF16_TOL)
    _close(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS, bias=bd, tile=tile), y + bias, BF16_TOL)
    out, dact = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=bd, tile=tile)
    _close(out, _gelu(y + bias), BF16_TOL)
    _close(dact, _gelu_grad(y + bias), BF16_TOL)               # the second output: d out / d (pre-activation), what EPI_GELU_BWD multiplies by
    res = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=bd, resid=rd, tile=tile)
    assert res.dtype == torch.float32
    _close(res, resid + y + bias, F32_TOL * 4)
    _close(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_F32, tile=tile), y, F32_TOL)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (392, 768, 2304), (640, 3072, 768), (200, 136, 64)])
def test_gemm_nn_data_gradient(device, m, n, k, tile):
    """dx = g W (W [k = out features][n = in features], read K-major through the transposing LDS reads)."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(m + n + k + tile)
    g, w = _rand((m, k), gen), _rand((k, n), gen, k ** -0.5)
    pre = _rand((m, n), gen)
    gd, wd, pd = g.to(device), w.to(device), pre.to(device)
    y = g.float() @ w.float()
    _close(S.gemm_bf16(gd, wd, S.GEMM_NN, S.EPI_NONE, tile=tile), y, BF16_TOL)
    _close(S.gemm_bf16(gd, wd, S.GEMM_NN, S.EPI_GELU_BWD, pre=pd, tile=tile), y * pre.float(), BF16_TOL)   # (pre: the stored derivative factor)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("m,n,k,splits", [(256, 256, 128, 1), (768, 768, 392, 1), (2304, 768, 2048, 4), (768, 3072, 1024, 8),
                                          (136, 264, 200, 2), (768, 768, 8192, 16),
                                          # contraction lengths whose K steps 8 (4) slices cannot cover without an empty slice
                                          # (12 / 13 / 20 / 28 steps of 64; 5 steps): the slice count steps down
                                          (768, 768, 768, 8), (768, 768, 784, 8), (768, 768, 1280, 8), (2304, 768, 1792, 8),
                                          (768, 768, 320, 4), (768, 768, 1100, 16)])
def test_gemm_tn_weight_gradient(device, m, n, k, splits, tile):
    """dW[m = out][n = in] = sum_tokens g[token][m] x[token][n]: both operands K-major, fp32 result, split over the tokens;
    the token count need not be a multiple of the 64-deep K step (rows past the end read as zeros)."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(m + n + k + splits + tile)
    g, x = _rand((k, m), gen), _rand((k, n), gen)
    ref = g.float().t() @ x.float()
    got = S.gemm_bf16(g.to(device), x.to(device), S.GEMM_TN, S.EPI_F32, splits=splits, tile=tile)
    assert got.dtype == torch.float32
    _close(got, ref, F32_TOL)
    again = S.gemm_bf16(g.to(device), x.to(device), S.GEMM_TN, S.EPI_F32, splits=splits, tile=tile)
    assert torch.equal(got, again), "split-K partial sums are added in slab order: bitwise repeatable"


@pytest.mark.parametrize("tile", [0, 4, 6])
@pytest.mark.parametrize("m,n,k,splits", [(2304, 768, 8192, 4), (768, 768, 8192, 8), (3072, 768, 2048, 4), (136, 264, 200, 2), (264, 72, 1000, 2),
                                          (768, 768, 784, 8)])
def test_gemm_weight_and_bias_gradient_in_one_pair_of_launches(device, m, n, k, splits, tile):
    """sis_gemm_bf16_wgrad_bias: the weight gradient is bitwise the plain split-K TN run, the bias gradient (column sums of the
    gradient, computed by extra workgroups of the same launches) bitwise sis_column_sum -- same summation orders -- incl. output
    widths that are not multiples of 256 / of the tile and row counts that are not multiples of anything."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(m + n + k + splits + tile)
    g, x = _rand((k, m), gen).to(device), _rand((k, n), gen).to(device)
    dw, db = S.gemm_bf16_wgrad_bias(g, x, splits, tile)
    assert torch.equal(dw, S.gemm_bf16(g, x, S.GEMM_TN, S.EPI_F32, splits=splits, tile=tile))
    assert torch.equal(db, S.column_sum(g))
    _close(db, g.float().sum(0).cpu(), F32_TOL)
    dw2, db2 = S.gemm_bf16_wgrad_bias(g, x, splits, tile)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)


def test_gemm_strided_views(device):
    """Operands are taken as row-strided views: q / k / v column blocks of the fused projection, no copies."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(5)
    qkv, w = _rand((512, 2304), gen), _rand((768, 768), gen, 768 ** -0.5)
    qd = qkv.to(device)
    for c0 in (0, 768, 1536):
        _close(S.gemm_bf16(qd[:, c0:c0 + 768], w.to(device), S.GEMM_NT), qkv[:, c0:c0 + 768].float() @ w.float().t(), BF16_TOL)
        ref = qkv[:, c0:c0 + 768].float().t() @ qkv[:, :768].float()
        _close(S.gemm_bf16(qd[:, c0:c0 + 768], qd[:, :768], S.GEMM_TN, S.EPI_F32, splits=2), ref, F32_TOL)


def test_gemm_dropout_stream(device):
    """Dropout in the epilogues: the dropped fraction is p, survivors are scaled by 1 / (1 - p), forward and backward of a
    site see the same mask (nothing is stored), two sites and two steps (seed words) see different ones."""
    import sis_hip as S
    m, n, k, p = 1024, 768, 128, 0.1
    gen = torch.Generator().manual_seed(11)
    x, w = _rand((m, k), gen), _rand((n, k), gen, k ** -0.5)
    xd, wd = x.to(device), w.to(device)
    zero_b, zero_r = torch.zeros(n, device=device), torch.zeros(m, n, device=device)
    seed = torch.tensor([12345], dtype=torch.int64, device=device)
    plain = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=zero_b, resid=zero_r)
    drop = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=zero_b, resid=zero_r, seed=seed, site=3, drop_p=p)
    assert (plain == 0).sum().item() == 0
    mask = drop != 0
    frac = 1 - mask.float().mean().item()
    assert abs(frac - p) < 5e-3, frac
    thr = int(p * 65536 + 0.5)            # the probability is realised to 2^-16 (csrc/vit_common.h)
    scale = 65536 / (65536 - thr)
    assert torch.allclose(drop[mask], plain[mask] * scale, rtol=1e-6, atol=0)
    # the backward of the same site: bf16(g * factor) with the identical mask
    g = torch.randn(m, n, generator=gen).to(device)
    gb = S.dropout_bwd_cast(g, seed, 3, p)
    assert torch.equal(gb == 0, ~mask | (g.bfloat16() == 0))
    _close(gb[mask], (g[mask] * scale).cpu(), BF16_TOL)
    other_site = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=zero_b, resid=zero_r, seed=seed, site=4, drop_p=p) != 0
    S.dropout_advance(seed)
    assert seed.item() != 12345
    other_step = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=zero_b, resid=zero_r, seed=seed, site=3, drop_p=p) != 0
    for other in (other_site, other_step):
        agree = (other == mask).float().mean().item()
        assert abs(agree - (p * p + (1 - p) ** 2)) < 1e-2, agree    # independent masks agree on p^2 + (1-p)^2 of the elements
    # fc1 forward (GELU + dropout) and the fc2 data gradient's epilogue (dropout * gelu') share a site and a mask
    seed2 = torch.tensor([777], dtype=torch.int64, device=device)
    out, dact = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=zero_b, seed=seed2, site=9, drop_p=p)
    out0, dact0 = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=zero_b)
    gy, w2 = _rand((m, k), gen).to(device), _rand((k, n), gen, k ** -0.5).to(device)
    dpre = S.gemm_bf16(gy, w2, S.GEMM_NN, S.EPI_GELU_BWD, pre=dact)
    dpre0 = S.gemm_bf16(gy, w2, S.GEMM_NN, S.EPI_GELU_BWD, pre=dact0)
    fwd_dropped = (out == 0) & (out0 != 0)
    fac_dropped = (dact == 0) & (dact0 != 0)
    bwd_dropped = (dpre == 0) & (dpre0 != 0)
    assert fwd_dropped.float().mean().item() > 0.08
    assert torch.equal(fwd_dropped & (dact0 != 0), fac_dropped & (out0 != 0))       # one mask for the activation and its derivative factor ...
    assert torch.equal(fac_dropped & (dpre0 != 0), bwd_dropped & (dact0 != 0))      # ... which is the mask of the data gradient
    keep = ~fac_dropped & (dact0 != 0)
    _close(dact[keep], (dact0[keep].float() * scale).cpu(), BF16_TOL)


def test_gemm_full_size_qkv(device):
    """configs[4] size: 8 192 tokens x (768 -> 2304), every tile of the grid against the CPU."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(99)
    x, w, b = _rand((8192, 768), gen), _rand((2304, 768), gen, 768 ** -0.5), torch.randn(2304, generator=gen)
    torch.set_num_threads(16)
    ref = x.float() @ w.float().t() + b
    for tile in range(9):
        _close(S.gemm_bf16(x.to(device), w.to(device), S.GEMM_NT, S.EPI_BIAS, bias=b.to(device), tile=tile), ref, BF16_TOL)


@pytest.mark.parametrize("batch,cin,cout,hw", [(2, 128, 256, 1024), (8, 512, 128, 4096), (3, 256, 1024, 1024), (4, 64, 72, 256)])
def test_gemm_batched_as_pointwise_convolution(device, batch, cin, cout, hw):
    """The three GEMMs of a bf16 1x1 convolution on NCHW tensors, one launch each (sis_gemm_bf16_batched): forward (NN, the
    weight shared by the images), data gradient (TN: the same weight tensor read K-major), weight gradient (NT summed over the
    images), against fp32 CPU arithmetic on the bf16-rounded operands (F.conv2d and its autograd: the reference's
    vit_seg_modeling_resnet_skip.py:30-37 conv1x1)."""
    import torch.nn.functional as F
    import sis_hip as S
    gen = torch.Generator().manual_seed(batch + cin + cout)
    x = torch.randn(batch, cin, hw, generator=gen).bfloat16()
    w = (torch.randn(cout, cin, generator=gen) * cin ** -0.5).bfloat16()
    gy = torch.randn(batch, cout, hw, generator=gen).bfloat16()
    xr = x.float().view(batch, cin, hw, 1).requires_grad_(True)
    wr = w.float().view(cout, cin, 1, 1).requires_grad_(True)
    yr = F.conv2d(xr, wr)
    yr.backward(gy.float().view(batch, cout, hw, 1))
    xd, wd, gd = x.to(device), w.to(device), gy.to(device)
    y = S.gemm_bf16_batched(wd, xd, S.GEMM_NN)
    assert tuple(y.shape) == (batch, cout, hw)
    _close(y, yr.detach().view(batch, cout, hw), BF16_TOL)
    dx = S.gemm_bf16_batched(wd, gd, S.GEMM_TN)
    _close(dx, xr.grad.view(batch, cin, hw), BF16_TOL)
    if batch in (1, 2, 4) or batch % 8 == 0:
        dw = S.gemm_bf16_batched(gd, xd, S.GEMM_NT, S.EPI_F32, sum_over_batches=True)
        assert dw.dtype == torch.float32 and tuple(dw.shape) == (cout, cin)
        _close(dw, wr.grad.view(cout, cin), F32_TOL)
        assert torch.equal(dw, S.gemm_bf16_batched(gd, xd, S.GEMM_NT, S.EPI_F32, sum_over_batches=True))


# ---- the 256-row tiles (csrc/gemm256_bf16.hip): NT layout, every bf16 / fp32-residual epilogue, whole and ragged tiles
G256_CASES = [(256, 96, 128), (512, 288, 768), (1024, 768, 768), (768, 2304, 768), (512, 3072, 768), (1024, 768, 3072),
              (300, 100, 192), (8192, 768, 128), (260, 580, 2304), (64, 96, 256)]


@pytest.mark.parametrize("tile", [9, 10, 11])
@pytest.mark.parametrize("m,n,k", G256_CASES)
def test_gemm256_nt_epilogues(device, m, n, k, tile):
    """One 8-wave workgroup per 256 x 96 / 192 / 288 output tile, fragments double-buffered in registers over four LDS stages:
    the same results as the 128-wide tiles (and the fp32 CPU product) for every epilogue the encoder uses, including ragged
    row / column tiles (rows beyond M and N read as zeros through the buffer range check) and the shortest K loop (4 steps)."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(m * 7 + n * 3 + k + tile)
    x, w = _rand((m, k), gen), _rand((n, k), gen, k ** -0.5)
    bias = torch.randn(n, generator=gen)
    resid = torch.randn(m, n, generator=gen)
    pre_in = _rand((m, n), gen)
    xd, wd, bd, rd, pd = x.to(device), w.to(device), bias.to(device), resid.to(device), pre_in.to(device)
    y = x.float() @ w.float().t()
    _close(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_NONE, tile=tile), y, BF16_TOL)
    _close(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS, bias=bd, tile=tile), y + bias, BF16_TOL)
    out, dact = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=bd, tile=tile)
    _close(out, _gelu(y + bias), BF16_TOL)
    _close(dact, _gelu_grad(y + bias), BF16_TOL)
    res = S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=bd, resid=rd, tile=tile)
    assert res.dtype == torch.float32
    _close(res, resid + y + bias, F32_TOL * 4)
    _close(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_GELU_BWD, pre=pd, tile=tile), y * pre_in.float(), BF16_TOL)
    # bitwise the 128-wide kernel's result: both accumulate the K steps in order in fp32 and round once
    assert torch.equal(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS, bias=bd, tile=tile), S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS, bias=bd, tile=0))
    if n % 12 == 0:   # three bias segments (query | key | value)
        seg = n // 3
        b3 = (bd[:seg].contiguous(), bd[seg:2 * seg].contiguous(), bd[2 * seg:].contiguous())
        _close(S.gemm_bf16(xd, wd, S.GEMM_NT, S.EPI_BIAS, bias=b3, tile=tile), y + bias, BF16_TOL)


def test_gemm256_dropout_masks_match_the_128_wide_kernel(device):
    """The dropout stream is a function of (seed, site, element index): the 256-row tiles must draw the masks the other kernels
    (and the backward's recomputation) draw."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(9)
    m, n, k = 512, 768, 768
    x, w = _rand((m, k), gen).to(device), _rand((n, k), gen, k ** -0.5).to(device)
    bias, resid = torch.randn(n, generator=gen).to(device), torch.randn(m, n, generator=gen).to(device)
    seed = S.dropout_seed(device)
    a = S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=bias, resid=resid, seed=seed, site=5, drop_p=0.1, tile=9)
    b = S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS_DROP_RESID, bias=bias, resid=resid, seed=seed, site=5, drop_p=0.1, tile=0)
    assert torch.equal(a, b)
    o1, p1 = S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=bias, seed=seed, site=6, drop_p=0.1, tile=9)
    o0, p0 = S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_BIAS_GELU_DROP, bias=bias, seed=seed, site=6, drop_p=0.1, tile=0)
    assert torch.equal(o1, o0) and torch.equal(p1, p0)
    dropped = (o1 == 0).float().mean().item()
    assert 0.08 < dropped < 0.13


def test_gemm256_rejects_what_it_does_not_serve(device):
    import sis_hip as S
    x, w = torch.zeros(256, 96, dtype=torch.bfloat16, device=device), torch.zeros(96, 96, dtype=torch.bfloat16, device=device)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        S.gemm_bf16(x, w, S.GEMM_NT, S.EPI_NONE, tile=9)          # k = 96: not a multiple of 64 (the layout's own rule)
    with pytest.raises(RuntimeError, match="256 rows"):
        S.gemm_bf16(x[:, :64].contiguous(), w[:, :64].contiguous(), S.GEMM_NT, S.EPI_NONE, tile=10)   # k = 64 < 128: fewer than 4 K steps
    with pytest.raises(RuntimeError, match="256 rows"):
        S.gemm_bf16(torch.zeros(128, 256, dtype=torch.bfloat16, device=device), torch.zeros(256, 96, dtype=torch.bfloat16, device=device),
                    S.GEMM_NN, S.EPI_NONE, tile=11)               # the NN layout stays on the 128-wide tiles
    assert S.gemm_tile_256(8192, 2304, 768) == S.TILE_256X288 and S.gemm_tile_256(2048, 2304, 768) is None
    if S._GEMM256_WIDTHS == (288, 192):   # the shipped set: 256 x 96, slower than the 128-wide tiles on 768-wide outputs, stays off
        assert S.gemm_tile_256(8192, 3072, 768) == S.TILE_256X192 and S.gemm_tile_256(8192, 768, 3072) is None


def test_transpose_bank(device):
    import sis_hip as S
    gen = torch.Generator().manual_seed(3)
    mats = [_rand(s, gen).to(device) for s in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (70, 130), (64, 64))]
    bank = S.TransposeBank(mats)
    bank.refresh()
    for src, dst in zip(mats, bank.out):
        assert torch.equal(dst, src.t().contiguous())
    mats[1].mul_(2)
    bank.refresh()
    assert torch.equal(bank.out[1], mats[1].t().contiguous()) and bank.current(mats)


def test_dropout_quads_are_pairwise_independent(device):
    """the four uniforms of a quad must be independent -- the joint drop frequency of every element pair of a quad is
    p^2 (here p = 0.3: 0.09 +- 3 sigma of 2 M quads), including the pairs (0, 2) and (1, 3) whose second member used to be a
    function of the first."""
    import sis_hip as S
    m, n, p = 8192, 1024, 0.3
    g = torch.ones(m, n, device=device)
    seed = torch.tensor([2024], dtype=torch.int64, device=device)
    dropped = (S.dropout_bwd_cast(g, seed, 7, p) == 0).view(-1, 4).float()
    marg = dropped.mean(0)
    assert (marg - p).abs().max().item() < 2e-3, marg
    quads = dropped.shape[0]
    tol = 4 * (p * p * (1 - p * p) / quads) ** 0.5
    for i in range(4):
        for j in range(i + 1, 4):
            joint = (dropped[:, i] * dropped[:, j]).mean().item()
            assert abs(joint - p * p) < tol, (i, j, joint)


@pytest.mark.parametrize("shape", [(8, 768, 1024), (2, 1024, 768), (3, 70, 33), (1, 1, 5), (2, 64, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_swap_last2_is_the_contiguous_transpose(device, shape, dtype):
    """sis_transpose_batched (csrc/vit_elementwise.hip): [B, R, C] -> [B, C, R], bitwise ``x.transpose(-1, -2).contiguous()``, and the
    gradient is the same operation (TransUNet: patch embeddings -> tokens, tokens -> the decoder's feature map)."""
    import sis_hip
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))).to(device).to(dtype).requires_grad_(True)
    y = sis_hip.swap_last2(x)
    assert y.is_contiguous() and torch.equal(y, x.detach().transpose(-1, -2).contiguous())
    g = torch.randn_like(y)
    y.backward(g)
    assert torch.equal(x.grad, g.transpose(-1, -2).contiguous())
    if min(shape[1:]) > 1:   # (a transposed view with a size-1 axis is still contiguous)
        with pytest.raises(RuntimeError, match="contiguous 3-d tensor"):
            sis_hip.swap_last2(x.detach().transpose(-1, -2))
