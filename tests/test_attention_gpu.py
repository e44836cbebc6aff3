"""GPU parity: fused attention forward / backward (csrc/attention_bf16.hip) through the C ABI against the reference's own
formulation evaluated in fp32 on the CPU from the same bf16-rounded q / k / v
(networks/trans_u_net/vit_seg_modeling.py:76-96: scores = q k^T / sqrt(64), softmax, probs @ v, heads merged).

Stated tolerance: probabilities (and dS) are rounded to bf16 before their second product and the outputs are bf16:
|err| <= 2e-2 * max|ref| element-wise; the log-sum-exp is fp32 arithmetic on fp32 MFMA sums: 1e-3 absolute."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-2


def _reference(qkv, heads, d_ctx=None):
    b, n, _ = qkv.shape
    x = qkv.float().clone().requires_grad_(d_ctx is not None)
    q, k, v = (t.view(b, n, heads, 64).permute(0, 2, 1, 3) for t in x.split(heads * 64, dim=-1))
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(64)
    probs = torch.softmax(scores, dim=-1)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(b, n, heads * 64)
    lse = torch.logsumexp(scores, dim=-1)
    if d_ctx is None:
        return ctx, lse, None
    ctx.backward(d_ctx.float())
    return ctx.detach(), lse.detach(), x.grad


def _close(got, ref, tol=TOL):
    err = (got.float().cpu() - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item(), (err, ref.abs().max().item())


@pytest.mark.parametrize("b,n,heads,spread", [(1, 128, 1, 1.0), (2, 256, 3, 1.0), (1, 1024, 2, 1.0), (2, 196, 2, 1.0), (1, 70, 1, 1.0),
                                              (1, 320, 12, 3.0)])
def test_attention_forward_backward(device, b, n, heads, spread):
    import sis_hip as S
    gen = torch.Generator().manual_seed(b * 1000 + n + heads)
    qkv = (torch.randn(b, n, 3 * heads * 64, generator=gen) * spread).bfloat16()
    d_ctx = torch.randn(b, n, heads * 64, generator=gen).bfloat16()
    ref_ctx, ref_lse, ref_grad = _reference(qkv, heads, d_ctx)
    qd = qkv.to(device)
    ctx, lse = S.attention_fwd(qd, heads)
    assert ctx.dtype == torch.bfloat16 and tuple(ctx.shape) == (b, n, heads * 64)
    _close(ctx, ref_ctx)
    assert (lse.cpu() - ref_lse).abs().max().item() <= 1e-3 * max(1.0, ref_lse.abs().max().item())
    d_qkv = S.attention_bwd(d_ctx.to(device), qd, ctx, lse, heads)
    assert d_qkv.dtype == torch.bfloat16 and d_qkv.shape == qd.shape
    hd = heads * 64
    for name, sl in (("dq", slice(0, hd)), ("dk", slice(hd, 2 * hd)), ("dv", slice(2 * hd, 3 * hd))):
        err = (d_qkv[..., sl].float().cpu() - ref_grad[..., sl]).abs().max().item()
        assert err <= TOL * ref_grad[..., sl].abs().max().item(), (name, err, ref_grad[..., sl].abs().max().item())
    again = S.attention_bwd(d_ctx.to(device), qd, ctx, lse, heads)
    assert torch.equal(d_qkv, again), "no atomics: the backward is bitwise repeatable"


def test_attention_max_jump(device):
    """Online softmax: a key far above the running maximum arrives in a late tile (the rescale path), and one query row
    whose scores are all far below zero (no underflow to a zero normaliser)."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(3)
    b, n, heads = 1, 512, 1
    qkv = torch.randn(b, n, 192, generator=gen)
    qkv[0, 400, 64:128] = qkv[0, 17, 0:64] * 6.0      # key 400 (7th tile) lines up with query 17
    qkv[0, 33, 0:64] *= 20.0                          # query 33: huge score spread
    qkv = qkv.bfloat16()
    ref_ctx, ref_lse, _ = _reference(qkv, heads)
    ctx, lse = S.attention_fwd(qkv.to(device), heads)
    _close(ctx, ref_ctx)
    assert torch.isfinite(lse).all()
    assert (lse.cpu() - ref_lse).abs().max().item() <= 1e-3 * ref_lse.abs().max().item()


def test_attention_full_size(device):
    """configs[4] size (8 images x 12 heads x 1024 tokens): two heads of two images against the CPU."""
    import sis_hip as S
    gen = torch.Generator().manual_seed(8)
    qkv = torch.randn(8, 1024, 2304, generator=gen).bfloat16()
    d_ctx = torch.randn(8, 1024, 768, generator=gen).bfloat16()
    qd = qkv.to(device)
    ctx, lse = S.attention_fwd(qd, 12)
    d_qkv = S.attention_bwd(d_ctx.to(device), qd, ctx, lse, 12)
    for img, head in ((0, 0), (7, 11), (3, 5)):
        cols = [slice(k * 768 + head * 64, k * 768 + head * 64 + 64) for k in range(3)]
        sub = torch.cat([qkv[img:img + 1, :, c] for c in cols], dim=-1)
        ref_ctx, ref_lse, ref_grad = _reference(sub, 1, d_ctx[img:img + 1, :, head * 64:head * 64 + 64])
        _close(ctx[img:img + 1, :, head * 64:head * 64 + 64], ref_ctx)
        for k, c in enumerate(cols):
            _close(d_qkv[img:img + 1, :, c], ref_grad[..., k * 64:k * 64 + 64])
