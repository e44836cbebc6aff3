"""GPU parity at the level of one transformer block: the fused autograd function (networks/trans_u_net/vit_encoder.py::
_FusedBlockFn: LayerNorm -> bf16 GEMM epilogues -> fused attention -> ... and its hand-written backward) against the SAME
module evaluated module-by-module in fp64 on the CPU (the reference's composition, vit_seg_modeling.py:53-122,171-190).

There is no ReLU in the block (GELU is smooth), so unlike the whole-network comparison nothing flips: the deviation is the
rounding of the bf16 operands and outputs.  Error model: every Linear output is one bf16 rounding (2^-9 relative) of an fp32
sum, the residual stream stays fp32; a block chains 4 such roundings in the forward and 8 in the backward.  Stated tolerance:
output 1e-2 * max|ref|, input gradient 2e-2 relative L2, every parameter gradient 2.5e-2 relative L2 (measured: see the
assertion messages of a failing run; typical 3e-3 / 6e-3 / 4e-3 .. 1e-2)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _config(dropout):
    from networks.trans_u_net.vit_seg_configs import get_r50_b16_config
    cfg = get_r50_b16_config()
    cfg.transformer["dropout_rate"] = dropout
    return cfg


# (3, 256) / (4, 196): 768 / 784 tokens = 12 / 13 K steps of the weight-gradient GEMMs, which 8 split-K slices cannot cover
# without an empty slice -- gemm_impl steps the slice count down instead of failing
@pytest.mark.parametrize("batch,tokens", [(2, 256), (1, 196), (3, 70), (3, 256), (4, 196)])
def test_fused_block_matches_the_module_composition(device, batch, tokens):
    from networks.trans_u_net import vit_encoder as V
    torch.manual_seed(batch * 100 + tokens)
    block = V.Block(_config(0.0), vis=False)
    with torch.no_grad():   # non-trivial norms / biases (the default init leaves biases at ~0 and norms at identity)
        for name, p in block.named_parameters():
            if name.endswith("bias"):
                p.normal_(0, 0.05)
            elif "norm" in name:
                p.add_(0.1 * torch.randn_like(p))
    ref = copy.deepcopy(block).double()
    x = torch.randn(batch, tokens, 768)
    gy = torch.randn(batch, tokens, 768)
    xr = x.double().requires_grad_(True)
    yr, _ = ref(xr)
    yr.backward(gy.double())
    block = block.to(device).train()
    xd = x.to(device).requires_grad_(True)
    assert block._fused_ok(xd) is False   # outside autocast the module path runs
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        assert block._fused_ok(xd)
        y, weights = block(xd)
    assert weights is None and y.dtype == torch.float32
    y.backward(gy.to(device))
    err = (y.detach().cpu().double() - yr.detach()).abs().max().item()
    assert err <= 1e-2 * yr.abs().max().item(), ("output", err, yr.abs().max().item())
    rel = ((xd.grad.cpu().double() - xr.grad).norm() / xr.grad.norm()).item()
    assert rel <= 2e-2, ("input gradient", rel)
    got = dict(block.named_parameters())
    for name, p in ref.named_parameters():
        g = got[name].grad
        assert g is not None and g.dtype == torch.float32, name
        denom = p.grad.norm()
        if name == "attn.key.bias":   # exactly zero in exact arithmetic (softmax ignores a constant added to every key's score):
            denom = dict(ref.named_parameters())["attn.query.bias"].grad.norm()   # measured against the query bias' scale
        rel = ((g.cpu().double() - p.grad).norm() / (denom + 1e-30)).item()
        assert rel <= 2.5e-2, (name, rel)


def test_fused_block_dropout_is_a_function_of_the_seed_word(device):
    """Training-mode dropout (p = 0.1 on the MLP sites): two forwards under one seed word agree bitwise, advancing the word
    changes the masks, and the backward uses the forward's masks (finite-difference-free check: the gradient w.r.t. the
    block input of sum(y) equals what the module composition gives when fed the SAME masks -- here: zero-probability sites
    reproduce the p = 0 result, and with p = 0.1 the dropped fraction of the MLP branch is ~10 %)."""
    import sis_hip
    from networks.trans_u_net import vit_encoder as V
    torch.manual_seed(5)
    block = V.Block(_config(0.1), vis=False).to(device).train()
    x = torch.randn(2, 128, 768, device=device)
    seed = sis_hip.dropout_seed(device)
    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
        y0, _ = block(x)
        y1, _ = block(x)
        assert torch.equal(y0, y1)
        sis_hip.dropout_advance(seed)
        y2, _ = block(x)
        assert not torch.equal(y0, y2)
        block.eval()
        ye, _ = block(x)
    assert torch.isfinite(y0).all() and (y0 - ye).abs().max().item() > 0


def test_wgrad_plan_and_sites():
    from networks.trans_u_net import vit_encoder as V
    assert V._wgrad_plan(2304, 768) == (4, 0) and V._wgrad_plan(768, 768) == (8, 0)
    assert V._wgrad_plan(3072, 768) == (4, 4) and V._wgrad_plan(768, 3072) == (4, 4)


def test_two_forwards_then_two_backwards_use_their_own_dropout_masks(device):
    """every training forward of the encoder snapshots the dropout seed word it read, so a backward that runs after
    ANOTHER forward (gradient accumulation: forward, forward, backward, backward) still recomputes its own masks: the
    gradients equal those of forward-backward, forward-backward on the same two seed words."""
    import sis_hip
    from networks.trans_u_net import vit_encoder as V
    cfg = _config(0.1)
    cfg.transformer["num_layers"] = 2
    torch.manual_seed(7)
    enc = V.Encoder(cfg, vis=False).to(device).train()
    xa, xb = torch.randn(2, 128, 768, device=device), torch.randn(2, 128, 768, device=device)
    word = sis_hip.dropout_seed(device)

    def grads(interleaved):
        word.fill_(4242)
        enc.zero_grad(set_to_none=True)
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            if interleaved:
                ya, yb = enc(xa)[0], enc(xb)[0]
                ya.square().mean().backward()
                yb.square().mean().backward()
            else:
                enc(xa)[0].square().mean().backward()
                enc(xb)[0].square().mean().backward()
        return [p.grad.clone() for p in enc.parameters()]

    for u, v in zip(grads(False), grads(True)):
        assert torch.equal(u, v)


def test_deferred_layer_norm_reductions_are_the_undeferred_ones(device, monkeypatch):
    """SIS_DEFER_REDUCES: inside a backward the encoder's LayerNorm parameter-gradient rows are summed by ONE batched launch at
    the end of the backward (sis_hip.flush_deferred, an autograd-engine callback) instead of one launch per norm.  Same kernels
    body, same order: every gradient is bitwise the undeferred one; the queue is empty when backward() returns; jobs were in fact
    queued; and a direct call of the binding outside a backward is never deferred."""
    import sis_hip
    from networks.trans_u_net import vit_encoder as V
    cfg = _config(0.0)
    cfg.transformer["num_layers"] = 3
    torch.manual_seed(11)
    enc = V.Encoder(cfg, vis=False).to(device).train()
    x = torch.randn(2, 128, 768, device=device)

    monkeypatch.setattr(sis_hip, "_DEFER_WGRAD", False)   # (the batched weight gradients have a test of their own below)

    def grads(defer):
        monkeypatch.setattr(sis_hip, "_DEFER", defer)
        enc.zero_grad(set_to_none=True)
        queued = []
        flush = sis_hip.flush_deferred
        monkeypatch.setattr(sis_hip, "flush_deferred", lambda: (queued.append(sis_hip.deferred_pending()), flush())[1])
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            enc(x)[0].square().mean().backward()
        monkeypatch.setattr(sis_hip, "flush_deferred", flush)
        assert sis_hip.deferred_pending() == 0
        return {n: p.grad.clone() for n, p in enc.named_parameters()}, max(queued, default=0)

    plain, queued_off = grads(False)
    deferred, queued_on = grads(True)
    assert queued_off == 0 and queued_on >= 6, (queued_off, queued_on)   # two norms per block (+ the encoder's final norm undeferred or not)
    for name in plain:
        assert torch.equal(plain[name], deferred[name]), name
    # outside a backward: complete results at once
    n = 768
    xs, g = torch.randn(64, n, device=device), torch.randn(64, n, device=device)
    gamma, beta = torch.randn(n, device=device), torch.randn(n, device=device)
    _, mean, rstd = sis_hip.layer_norm_fwd(xs, gamma, beta, 1e-6, torch.float32)
    _, dg, db, _ = sis_hip.layer_norm_bwd_fused(g, xs, mean, rstd, gamma, defer=True)
    assert sis_hip.deferred_pending() == 0
    _, dg0, db0 = sis_hip.layer_norm_bwd(g, xs, mean, rstd, gamma)
    assert torch.equal(dg, dg0) and torch.equal(db, db0)


def test_deferred_weight_gradients_match_the_undeferred_ones(device, monkeypatch):
    """SIS_DEFER_WGRAD: inside a backward the weight / bias gradients of the blocks' Linear layers are queued and multiplied by ONE
    pointer-table launch per Linear shape at the end of the backward (sis_hip.defer_wgrad_bias / flush_deferred: every problem
    contracts all its tokens per tile instead of split-K slabs).  Same products, another order of the fp32 sums: 1e-5 of the
    largest entry; nothing is left queued; a second backward ON TOP of existing gradients (accumulation) takes the undeferred
    path and adds up; under a blocker (a consumer that reads gradients inside the backward) nothing is queued at all."""
    import sis_hip
    from networks.trans_u_net import vit_encoder as V
    cfg = _config(0.0)
    cfg.transformer["num_layers"] = 3
    torch.manual_seed(12)
    enc = V.Encoder(cfg, vis=False).to(device).train()
    x = torch.randn(2, 256, 768, device=device)

    def backward_once(zero=True):
        if zero:
            enc.zero_grad(set_to_none=True)
        queued = []
        flush = sis_hip.flush_deferred
        monkeypatch.setattr(sis_hip, "flush_deferred", lambda: (queued.append(sum(len(v) for v in sis_hip._deferred["wgrad"].values())), flush())[1])
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            enc(x)[0].square().mean().backward()
        monkeypatch.setattr(sis_hip, "flush_deferred", flush)
        assert sis_hip.deferred_pending() == 0
        return {n: p.grad.clone() for n, p in enc.named_parameters()}, max(queued, default=0)

    monkeypatch.setattr(sis_hip, "_DEFER_WGRAD", False)
    plain, queued_off = backward_once()
    monkeypatch.setattr(sis_hip, "_DEFER_WGRAD", True)
    deferred, queued_on = backward_once()
    assert queued_off == 0 and queued_on == 12, (queued_off, queued_on)   # four Linear layers per block, three blocks
    for name in plain:
        scale = float(plain[name].abs().max())
        assert float((plain[name] - deferred[name]).abs().max()) <= 1e-5 * scale + 1e-12, name
    twice, queued_acc = backward_once(zero=False)   # gradients in place: autograd adds on the spot, so nothing may be deferred
    assert queued_acc == 0
    for name in plain:
        scale = float(plain[name].abs().max())
        assert float((twice[name] - 2 * deferred[name]).abs().max()) <= 2e-5 * scale + 1e-12, name
    sis_hip.block_wgrad_deferral(enc)
    try:
        _, queued_blocked = backward_once()
    finally:
        sis_hip.block_wgrad_deferral(enc, False)
    assert queued_blocked == 0
    # a tensor hook reads the gradient INSIDE the backward: that Linear layer is multiplied on the spot, the others stay queued
    name, hooked = next((n, p) for n, p in enc.named_parameters() if n.endswith("ffn.fc1.weight"))
    seen = []
    handle = hooked.register_hook(lambda g: seen.append(g.clone()))
    try:
        with_hook, queued_hook = backward_once()
    finally:
        handle.remove()
    assert queued_hook == 11 and len(seen) == 1
    assert torch.equal(seen[0], with_hook[name])
    assert float((with_hook[name] - plain[name]).abs().max()) <= 1e-5 * float(plain[name].abs().max())


def test_block_output_gradient_cast_rides_in_the_next_blocks_layer_norm_backward(device, monkeypatch):
    """SIS_FUSE_BLOCK_CAST: block i's LayerNorm-1 backward writes, next to the input gradient g, bf16(g * dropout factor of block
    i - 1's fc2 site) -- what block i - 1's backward starts from -- instead of a cast launch at the top of that backward.  Same
    values, same dropout mask: every gradient bitwise equal to the unfused form (dropout 0.1 on the MLP sites, one seed word)."""
    import sis_hip
    from networks.trans_u_net import vit_encoder as V
    cfg = _config(0.1)
    cfg.transformer["num_layers"] = 3
    torch.manual_seed(21)
    enc = V.Encoder(cfg, vis=False).to(device).train()
    x = torch.randn(2, 128, 768, device=device, requires_grad=True)
    word = sis_hip.dropout_seed(device)

    def grads(fused):
        monkeypatch.setattr(V, "_FUSE_BLOCK_CAST", fused)
        word.fill_(777)
        enc.zero_grad(set_to_none=True)
        x.grad = None
        casts = []
        real = sis_hip.dropout_bwd_cast
        monkeypatch.setattr(sis_hip, "dropout_bwd_cast", lambda *a, **k: (casts.append(1), real(*a, **k))[1])
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            enc(x)[0].square().mean().backward()
        monkeypatch.setattr(sis_hip, "dropout_bwd_cast", real)
        assert not V._NEXT_BLOCK_CAST
        return [p.grad.clone() for p in enc.parameters()] + [x.grad.clone()], len(casts)

    plain, n_plain = grads(False)
    fused, n_fused = grads(True)
    assert n_plain == 3 and n_fused == 1, (n_plain, n_fused)   # only the last block (its gradient comes from the final norm) casts by itself
    for u, v in zip(plain, fused):
        assert torch.equal(u, v)
