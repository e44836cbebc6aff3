"""GPU parity: the fused TransUNet objective (csrc/loss_ops.hip) against the reference formulation on the CPU in fp32:
0.5 * nn.CrossEntropyLoss + 0.5 * DiceLoss(softmax=True)  (updater/segmentation_updater.py:95-102,
networks/trans_u_net/utils.py:7-42), value and gradient w.r.t. the logits.

Stated tolerance: fp32 arithmetic with a different summation order -> 2e-5 relative on the three loss values, 1e-5 * max|grad|
on the fp32 gradient; a bf16 gradient is that rounded once (2^-8 relative)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _reference(logits, labels, classes):
    from networks.trans_u_net.utils import DiceLoss
    z = logits.float().clone().requires_grad_(True)
    ce = nn.CrossEntropyLoss()(z, labels)
    dice = DiceLoss(classes)(z, labels, softmax=True)
    loss = 0.5 * ce + 0.5 * dice
    loss.backward()
    return loss.detach(), ce.detach(), dice.detach(), z.grad


@pytest.mark.parametrize("b,c,h,w,dtype", [(2, 3, 64, 64, torch.float32), (8, 3, 128, 128, torch.bfloat16), (1, 2, 32, 36, torch.float32),
                                           (3, 8, 40, 40, torch.bfloat16), (2, 5, 512, 512, torch.float32)])
def test_ce_dice_forward_backward(device, b, c, h, w, dtype):
    import sis_hip as S
    gen = torch.Generator().manual_seed(b + c + h)
    logits = (torch.randn(b, c, h, w, generator=gen) * 3).to(dtype)
    labels = torch.randint(0, c, (b, h, w), generator=gen)
    loss, ce, dice, grad = _reference(logits, labels, c)
    zd, ld = logits.to(device), labels.to(device)
    assert S.ce_dice_supported(zd, ld)
    out, stats = S.ce_dice_fwd(zd, ld)
    for got, want in zip(out.cpu().tolist(), (loss.item(), ce.item(), dice.item())):
        assert abs(got - want) <= 2e-5 * abs(want), (got, want)
    g = S.ce_dice_bwd(torch.tensor(1.0, device=device), zd, ld, stats)
    assert g.dtype == dtype and g.shape == zd.shape
    tol = 1e-5 if dtype == torch.float32 else 2 ** -8
    assert (g.float().cpu() - grad).abs().max().item() <= tol * grad.abs().max().item()
    g2 = S.ce_dice_bwd(torch.tensor(0.5, device=device), zd, ld, stats)
    assert torch.allclose(g2.float(), 0.5 * g.float(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-6, atol=1e-12)
    out2, _ = S.ce_dice_fwd(zd, ld)
    assert torch.equal(out, out2), "partials are added in workgroup order: bitwise repeatable"


def test_ce_dice_through_the_updater_function(device):
    """The autograd function the TransUNet updater calls: gradient flows to the logits, the observed parts do not."""
    from updater.segmentation_updater import _CeDiceFn
    gen = torch.Generator().manual_seed(4)
    logits = torch.randn(2, 3, 32, 32, generator=gen).to(device).requires_grad_(True)
    labels = torch.randint(0, 3, (2, 32, 32), generator=gen).to(device)
    loss, parts = _CeDiceFn.apply(logits, labels)
    assert not parts.requires_grad and loss.requires_grad
    loss.backward()
    want = _reference(logits.detach().cpu(), labels.cpu(), 3)
    assert abs(loss.item() - want[0].item()) < 2e-5 * want[0].item()
    assert (logits.grad.cpu() - want[3]).abs().max().item() <= 1e-5 * want[3].abs().max().item()
