"""GPU parity, part 3: the segmentation-training kernels against stock torch on the host (fp64 / fp32).

* sis_upsample_ce_fwd/bwd == F.interpolate(bilinear, align_corners=True) -> log_softmax -> NLL(ignore) -> mean
  (networks/ema_net/network.py:305-311, :319-327 of the reference), values and gradients;
* sis_sgd_momentum == torch.optim.SGD(momentum, weight_decay) over several steps, three groups;
* sis_ema_update == the in-place mu update of updater/segmentation_updater.py:56-66.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_loss(logits, labels, size, ignore):
    pred = F.interpolate(logits, size=size, mode="bilinear", align_corners=True)
    nll = F.nll_loss(F.log_softmax(pred, dim=1), labels, ignore_index=ignore, reduction="none")
    return nll.mean(dim=2).mean(dim=1)


@pytest.mark.parametrize("b,c,h,w,H,W", [(2, 3, 32, 32, 256, 256), (3, 5, 7, 9, 40, 33), (1, 21, 16, 16, 16, 16),
                                         (2, 2, 1, 1, 8, 8), (2, 4, 33, 33, 257, 257)])
def test_upsample_ce_forward_backward(device, b, c, h, w, H, W):
    import sis_hip
    gen = torch.Generator().manual_seed(b + c + h + H)
    logits = torch.randn(b, c, h, w, generator=gen) * 2
    labels = torch.randint(0, c, (b, H, W), generator=gen)
    labels[torch.rand(b, H, W, generator=gen) < 0.1] = 255
    gl = torch.randn(b, generator=gen)
    x64 = logits.double().requires_grad_(True)
    ref = _ref_loss(x64, labels, (H, W), 255)
    (gref,) = torch.autograd.grad(ref, x64, gl.double())
    loss = sis_hip.upsample_ce_fwd(logits.to(device), labels.to(device), (H, W), 255)
    assert torch.allclose(loss.cpu().double(), ref.detach(), rtol=2e-5, atol=1e-6)
    gx = sis_hip.upsample_ce_bwd(gl.to(device), logits.to(device), labels.to(device), (H, W), 255)
    scale = gref.abs().max().item()
    assert (gx.cpu().double() - gref).abs().max().item() <= 2e-5 * scale + 1e-9
    # determinism: no atomics anywhere
    assert torch.equal(loss, sis_hip.upsample_ce_fwd(logits.to(device), labels.to(device), (H, W), 255))
    assert torch.equal(gx, sis_hip.upsample_ce_bwd(gl.to(device), logits.to(device), labels.to(device), (H, W), 255))


def test_upsample_ce_autograd_through_network_tail(device):
    from networks.ema_net.network import _UpsampleCrossEntropy
    gen = torch.Generator().manual_seed(3)
    logits = torch.randn(2, 3, 32, 32, generator=gen)
    labels = torch.randint(0, 3, (2, 256, 256), generator=gen)
    xr = logits.clone().requires_grad_(True)
    _ref_loss(xr, labels, (256, 256), 255).mean().backward()
    xd = logits.to(device).requires_grad_(True)
    _UpsampleCrossEntropy.apply(xd, labels.to(device), (256, 256), 255).mean().backward()
    assert torch.allclose(xd.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-9)


def test_fused_sgd_matches_torch_sgd(device):
    from training.fused_sgd import FusedSGD
    gen = torch.Generator().manual_seed(5)
    shapes = [(64, 3, 3, 3), (70000,), (128,), (3, 256, 1, 1), (1,), (200000,)]
    ref_p = [torch.randn(*s, generator=gen).requires_grad_(True) for s in shapes]
    dev_p = [p.detach().clone().to(device).requires_grad_(True) for p in ref_p]
    groups = lambda ps: [{"params": ps[:2], "lr": 0.009, "weight_decay": 1e-4},
                         {"params": ps[2:4], "lr": 0.009, "weight_decay": 0},
                         {"params": ps[4:], "lr": 0.018, "weight_decay": 0.0}]
    ref_opt = torch.optim.SGD(groups(ref_p), momentum=0.9)
    dev_opt = FusedSGD(groups(dev_p), momentum=0.9)
    for step in range(4):
        for pr, pd in zip(ref_p, dev_p):
            g = torch.randn(pr.shape, generator=gen)
            pr.grad = g.clone()
            if pd.grad is None:
                pd.grad = g.to(device)
            else:
                pd.grad.copy_(g)
        if step == 2:
            for opt in (ref_opt, dev_opt):
                for gr in opt.param_groups:
                    gr["lr"] *= 0.5
        ref_opt.step()
        dev_opt.step()
        for pr, pd in zip(ref_p, dev_p):
            assert torch.allclose(pd.detach().cpu(), pr.detach(), rtol=1e-6, atol=1e-7), step
    sd = dev_opt.state_dict()
    assert all("momentum_buffer" in s for s in sd["state"].values())
    # a parameter without gradient is skipped, like torch.optim.SGD
    extra = torch.zeros(5, device=device, requires_grad=True)
    opt = FusedSGD([extra, dev_p[0]], lr=0.1, momentum=0.9)
    opt.step()
    assert torch.equal(extra.detach().cpu(), torch.zeros(5))


def test_ema_update(device):
    import sis_hip
    gen = torch.Generator().manual_seed(9)
    mu = torch.randn(1, 512, 64, generator=gen)
    mub = torch.randn(6, 512, 64, generator=gen)
    ref = mu.clone()
    ref *= 0.9
    ref += mub.mean(dim=0, keepdim=True) * (1 - 0.9)
    d = mu.to(device)
    sis_hip.ema_update(d, mub.to(device), 0.9)
    assert torch.allclose(d.cpu(), ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("b,c,h,w", [(2, 64, 128, 128), (16, 8, 4, 4), (3, 20, 10, 6), (2, 512, 32, 32)])
@pytest.mark.parametrize("relu,use_res", [(True, True), (True, False), (False, False), (False, True)])
def test_fused_batch_norm_act(device, b, c, h, w, relu, use_res):
    """csrc/bn_ops.hip vs F.batch_norm (+ add + relu) in fp64: output, running statistics, every gradient."""
    from networks.ema_net.network import SynchronizedBatchNorm2d
    gen = torch.Generator().manual_seed(b + c + h)
    x = torch.randn(b, c, h, w, generator=gen) * 2 + 0.5
    res = torch.randn(b, c, h, w, generator=gen)
    gy = torch.randn(b, c, h, w, generator=gen)
    gamma, beta = 1 + 0.1 * torch.randn(c, generator=gen), 0.1 * torch.randn(c, generator=gen)
    xr, rr = x.double().requires_grad_(True), res.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    ref = F.batch_norm(xr, rm, rv, gr, br, True, 3e-4, 1e-5)
    if use_res:
        ref = ref + rr
    if relu:
        ref = F.relu(ref)
    grads_ref = torch.autograd.grad(ref, [xr, gr, br] + ([rr] if use_res else []), gy.double())
    bn = SynchronizedBatchNorm2d(c, momentum=3e-4).to(device).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    xd, rd = x.to(device).requires_grad_(True), res.to(device).requires_grad_(True)
    y = bn(xd, residual=rd if use_res else None, relu=relu)
    assert torch.allclose(y.detach().cpu().double(), ref.detach(), rtol=1e-4, atol=1e-5)
    assert torch.allclose(bn.running_mean.cpu().double(), rm, rtol=1e-4, atol=1e-7)
    assert torch.allclose(bn.running_var.cpu().double(), rv, rtol=1e-5)
    assert int(bn.num_batches_tracked) == 0
    grads = torch.autograd.grad(y, [xd, bn.weight, bn.bias] + ([rd] if use_res else []), gy.to(device))
    for got, want in zip(grads, grads_ref):
        scale = want.abs().max().item() + 1e-12
        assert (got.cpu().double() - want).abs().max().item() <= 2e-4 * scale
    bn.eval()
    with torch.no_grad():
        ye = bn(x.to(device), relu=relu)
        re = F.batch_norm(x.double(), rm, rv, gamma.double(), beta.double(), False, 3e-4, 1e-5)
        assert torch.allclose(ye.cpu().double(), F.relu(re) if relu else re, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("b,c,h,w,relu,use_res", [(16, 64, 32, 32, True, True), (4, 37, 16, 16, True, False), (16, 24, 32, 32, False, True),
                                                  (1, 8, 64, 64, False, False), (7, 16, 32, 32, True, True)])
def test_batch_norm_single_pass_equals_three_launches(device, b, c, h, w, relu, use_res, monkeypatch):
    """Channels whose batch * H * W values fit one workgroup (csrc/bn_ops.hip, bn_fused_*): statistics + apply in one launch and
    the backward in one launch: the forward gives BITWISE the results of the three-launch form (sign mask and running
    statistics included), the backward agrees to an ulp; batch sizes that leave the slice partly empty too."""
    import sis_hip
    gen = torch.Generator().manual_seed(b * 5 + c + h)
    x = (torch.randn(b, c, h, w, generator=gen) * 2 + 0.5).to(device)
    res = torch.randn(b, c, h, w, generator=gen).to(device) if use_res else None
    dy = torch.randn(b, c, h, w, generator=gen).to(device)
    gamma, beta = (1 + 0.1 * torch.randn(c, generator=gen)).to(device), (0.1 * torch.randn(c, generator=gen)).to(device)

    def run(single):
        monkeypatch.setenv("SIS_BN_SINGLE_PASS", "1" if single else "0")
        rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)
        if single:
            assert sis_hip.bn_fused_supported(x)
            y, mean, invstd, mask = sis_hip.bn_fused_fwd(x, res, gamma, beta, rm, rv, 1e-5, 3e-4, relu, want_mask=relu)
        else:
            assert not sis_hip.bn_fused_supported(x)
            mean, invstd = sis_hip.bn_stats(x, rm, rv, 1e-5, 3e-4)
            out = sis_hip.bn_act_fwd(x, res, mean, invstd, gamma, beta, relu, want_mask=relu)
            y, mask = out if relu else (out, None)
        grads = sis_hip.bn_act_bwd(dy, None if mask is not None else y, x, mean, invstd, gamma, relu, use_res, mask=mask)
        return [y, mean, invstd, rm, rv] + ([mask] if mask is not None else []) + [t for t in grads if t is not None]

    one, three = run(True), run(False)
    assert len(one) == len(three)
    n_fwd = 5 + (1 if relu else 0)
    for u, v in zip(one[:n_fwd], three[:n_fwd]):
        assert torch.equal(u, v)
    # backward: the same sums in the same order, but the compiler contracts the multiply-adds of the two kernels differently:
    # 1 ulp on the channel sums (measured 8e-8 relative), nothing else
    for u, v in zip(one[n_fwd:], three[n_fwd:]):
        assert (u - v).abs().max().item() <= 4e-7 * v.abs().max().item()
    assert torch.equal(one[n_fwd:][-1], run(True)[n_fwd:][-1])   # and repeatable


@pytest.mark.parametrize("b,c,h,w,relu,use_res", [(16, 64, 64, 64, True, True), (8, 32, 64, 64, True, False), (5, 16, 64, 64, False, True),
                                                  (2, 8, 128, 128, True, False), (16, 256, 64, 64, True, True)])
def test_batch_norm_wide_single_pass(device, b, c, h, w, relu, use_res, monkeypatch):
    """Channels of 16 385 ... 65 536 values (EMANet's 64 x 64 layers at batch 16; csrc/bn_ops.hip, bn_wide_*: one workgroup of 1 024
    threads per channel, the backward reads x a second time out of L2) against the three-launch form: statistics within 1e-6, outputs
    within 2e-6 of the largest value, gradients 2e-5, the ReLU gate identical except within rounding of zero; bitwise repeatable."""
    import sis_hip
    gen = torch.Generator().manual_seed(b * 3 + c + h)
    x = (torch.randn(b, c, h, w, generator=gen) * 2 + 0.5).to(device)
    res = torch.randn(b, c, h, w, generator=gen).to(device) if use_res else None
    dy = torch.randn(b, c, h, w, generator=gen).to(device)
    gamma, beta = (1 + 0.1 * torch.randn(c, generator=gen)).to(device), (0.1 * torch.randn(c, generator=gen)).to(device)

    def run(single):
        monkeypatch.setenv("SIS_BN_SINGLE_PASS", "1" if single else "0")
        rm, rv = torch.zeros(c, device=device), torch.ones(c, device=device)
        if single:
            assert sis_hip.bn_fused_supported(x)
            y, mean, invstd, mask = sis_hip.bn_fused_fwd(x, res, gamma, beta, rm, rv, 1e-5, 3e-4, relu, want_mask=relu)
        else:
            assert not sis_hip.bn_fused_supported(x)
            mean, invstd = sis_hip.bn_stats(x, rm, rv, 1e-5, 3e-4)
            out = sis_hip.bn_act_fwd(x, res, mean, invstd, gamma, beta, relu, want_mask=relu)
            y, mask = out if relu else (out, None)
        grads = sis_hip.bn_act_bwd(dy, None if mask is not None else y, x, mean, invstd, gamma, relu, use_res, mask=mask)
        return dict(y=y, mean=mean, invstd=invstd, rm=rm, rv=rv, mask=mask, grads=[t for t in grads if t is not None])

    one, three, again = run(True), run(False), run(True)
    for k in ("mean", "invstd", "rm", "rv"):
        assert (one[k] - three[k]).abs().max().item() <= 1e-6 * three[k].abs().max().item() + 1e-7, k
    scale = three["y"].abs().max().item()
    assert (one["y"] - three["y"]).abs().max().item() <= 2e-6 * scale
    if relu:
        flipped = sum(bin(int(v) & 0xFFFFFFFFFFFFFFFF).count("1") for v in (one["mask"] ^ three["mask"])[one["mask"] != three["mask"]].cpu().tolist())
        assert flipped <= 8, flipped
    for u, v in zip(one["grads"], three["grads"]):
        assert (u - v).abs().max().item() <= 2e-5 * v.abs().max().item()
    for k in ("y", "mean", "invstd", "rm", "rv"):
        assert torch.equal(one[k], again[k]), k
    for u, v in zip(one["grads"], again["grads"]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("b,c,h,w,use_res", [(2, 8, 16, 16, False), (3, 5, 12, 20, True), (16, 64, 32, 32, True), (1, 3, 6, 6, False)])
def test_batch_norm_relu_sign_mask_equals_the_output_gate(device, b, c, h, w, use_res):
    """The 1-bit-per-element ReLU gate written by the forward apply pass gives bit-for-bit the backward that reads y itself
    (csrc/bn_ops.hip: same arithmetic, only the source of [y > 0] differs), ragged last ballot group included."""
    import sis_hip
    gen = torch.Generator().manual_seed(b * 7 + c + h)
    x = (torch.randn(b, c, h, w, generator=gen) * 2 + 0.5).to(device)
    res = torch.randn(b, c, h, w, generator=gen).to(device) if use_res else None
    dy = torch.randn(b, c, h, w, generator=gen).to(device)
    gamma, beta = (1 + 0.1 * torch.randn(c, generator=gen)).to(device), (0.1 * torch.randn(c, generator=gen)).to(device)
    mean, invstd = sis_hip.bn_stats(x, None, None, 1e-5, 3e-4)
    y0 = sis_hip.bn_act_fwd(x, res, mean, invstd, gamma, beta, True)
    y1, mask = sis_hip.bn_act_fwd(x, res, mean, invstd, gamma, beta, True, want_mask=True)
    assert torch.equal(y0, y1) and mask.numel() == (((b * c * h * w) // 4 + 63) // 64) * 4
    want = sis_hip.bn_act_bwd(dy, y0, x, mean, invstd, gamma, True, use_res)
    got = sis_hip.bn_act_bwd(dy, None, x, mean, invstd, gamma, True, use_res, mask=mask)
    for g, w_ in zip(got, want):
        assert (g is None and w_ is None) or torch.equal(g, w_)
