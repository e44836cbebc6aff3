"""GPU parity, part 5: the TransUNet training step (product modules + TransUNetUpdater + FusedSGD on MI355X)
against the golden outputs of the reference (tests/golden/trans_u_net_step.npz)."""
import os

import numpy as np
import pytest
import torch

from oracle import ema_net_ref as E
from oracle import trans_u_net_ref as T

pytestmark = pytest.mark.gpu


def _net(device, size, classes, wseed):
    from networks.trans_u_net.vit_seg_modeling import VIT_CONFIGS, VisionTransformer
    cfg = VIT_CONFIGS["R50-ViT-B_16"].copy()
    cfg.n_classes, cfg.n_skip = classes, 3
    cfg.patches.grid = (size // 16, size // 16)
    cfg.transformer.dropout_rate = 0.0
    net = VisionTransformer(cfg, img_size=size, num_classes=classes)
    assert list(net.state_dict().keys()) == [k for k, _ in T.state_dict_schema(size, classes)]
    net.load_state_dict(T.seeded_state_dict(size, classes, seed=wseed), strict=True)
    return net.to(device).train()


def test_trans_u_net_two_iterations_vs_golden(device, golden_dir):
    from training.fused_sgd import FusedSGD
    from training.loop import get_current_reporter
    from updater.segmentation_updater import TransUNetUpdater
    g = np.load(os.path.join(golden_dir, "trans_u_net_step.npz"))
    size, classes, wseed, bseed, batch = g["cfg"].tolist()
    net = _net(device, size, classes, wseed)
    b0 = E.seeded_batch(batch, size, classes, seed=bseed)
    # instrumented first forward / backward
    pred = net(b0["images"].to(device))
    np.testing.assert_allclose(pred.detach()[:, :, ::8, ::8].cpu().numpy(), g["logits_slice"], rtol=5e-3, atol=2e-3)
    decided = g["margin"].astype(np.float32) > 1e-2
    assert (pred.argmax(1).cpu().numpy().astype(np.uint8)[decided] == g["labels"][decided]).all() and decided.mean() > 0.9
    gt = b0["segmented"].squeeze(1).to(device)
    from networks.trans_u_net.utils import DiceLoss
    ce = torch.nn.functional.cross_entropy(pred, gt)
    dice = DiceLoss(classes)(pred, gt, softmax=True)
    loss = 0.5 * ce + 0.5 * dice
    np.testing.assert_allclose([loss.item(), ce.item(), dice.item()], g["losses"], rtol=2e-4)
    loss.backward()
    grads = dict(net.named_parameters())
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        np.testing.assert_allclose(grads[str(name)].grad.double().norm().item(), ref, rtol=2e-2, atol=1e-8, err_msg=str(name))
    np.testing.assert_allclose(net.segmentation_head[0].weight.grad.cpu().numpy(), g["grad_head"], rtol=1e-2,
                               atol=2e-3 * np.abs(g["grad_head"]).max())
    net.zero_grad(set_to_none=True)
    net.load_state_dict(T.seeded_state_dict(size, classes, seed=wseed), strict=True)
    opt = FusedSGD(list(net.parameters()), lr=1e-4, momentum=0.9, weight_decay=1e-4)
    upd = TransUNetUpdater(num_classes=classes, iterators={"images": [E.seeded_batch(batch, size, classes, seed=bseed + i)
                                                                      for i in range(2)]},
                           networks={"segmentation": net}, optimizers={"main": opt}, device=device)
    upd.update()
    obs = get_current_reporter().scalars()
    np.testing.assert_allclose([obs["loss/combined"], obs["loss/CE"], obs["loss/Dice"]], g["losses"], rtol=2e-4)
    upd.update()
    np.testing.assert_allclose(get_current_reporter().scalars()["loss/combined"], g["loss_1"], rtol=2e-2)
    sd = net.state_dict()
    init = T.seeded_state_dict(size, classes, seed=wseed)
    for k in ("segmentation_head.0.weight", "segmentation_head.0.bias", "decoder.blocks.3.conv2.1.weight",
              "transformer.encoder.encoder_norm.weight"):
        ref = g["delta_" + k]
        np.testing.assert_allclose((sd[k].cpu() - init[k]).numpy(), ref, rtol=1e-1, atol=8e-2 * np.abs(ref).max(), err_msg=k)


def test_trans_u_net_512_shapes(device):
    """BASELINE.json configs[4] geometry: 512^2 input -> 32x32 tokens -> 512^2 logits; grayscale input is tiled."""
    net = _net(device, 512, 3, 5).eval()
    with torch.no_grad():
        out = net(torch.randn(1, 1, 512, 512, device=device))
        assert tuple(out.shape) == (1, 3, 512, 512) and torch.isfinite(out).all()
        assert tuple(net.predict_classes(torch.randn(1, 3, 512, 512, device=device)).shape) == (1, 1, 512, 512)
