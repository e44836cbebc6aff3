"""Generates tests/golden/ema_net_step.npz (and trans_u_net_step.npz) by running the UNMODIFIED reference
segmentation networks (imported by file path, oracle/load_reference.py) through one training iteration in
the order of updater/segmentation_updater.py:47-73 / :83-106 with torch.optim.SGD built as
training_builder/ema_net_train_builder.py:27-48 / trans_u_net_train_builder.py:39-40.

Run in the build container, in its own process (the reference's top-level ``networks`` package name clashes
with the product's).  Weights / batches are re-derived anywhere from the seeded numpy streams in
oracle/ema_net_ref.py / oracle/trans_u_net_ref.py; only outputs are stored.  Dropout layers are set to p = 0
(device RNG streams can never match across vendors, SURVEY.md §7 "hard parts").
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import load_reference  # noqa: E402
from oracle import ema_net_ref as E  # noqa: E402


def make_ema_net(ema, ema_utils, residual_scale=1.0, lr=2e-5, out_name="ema_net_step.npz"):
    """``residual_scale`` 0.1 / lr 0.009 (the shipped config's learning rate) -> ema_net_step_conditioned.npz: the
    well-conditioned fixture (oracle/ema_net_ref.py::seeded_state_dict) on which gradients are pinned at 1e-3."""
    torch.manual_seed(0)
    net = ema.EMANet(3, 50, use_pretrained_resnet=False)
    schema = E.state_dict_schema(50, 3)
    assert [k for k, _ in schema] == list(net.state_dict().keys())
    for (k, s), v in zip(schema, net.state_dict().values()):
        assert tuple(s) == tuple(v.shape), k
    net.load_state_dict(E.seeded_state_dict(50, 3, seed=7, residual_scale=residual_scale), strict=True)
    net.fc1[1].p = 0.0
    net.train()
    # wd / momenta of configs/segmenter/stylegan2_ema_net_segmenter.yaml:17-26; lr 2e-5 instead of 0.009: on this
    # randomly initialised net the first-step stem gradients are ~1000x the weights, so at lr 0.009 the second
    # iteration is chaotic (1e-7 perturbations change its loss by tens of percent) and pins nothing
    wd, mom, em_mom = 1e-4, 0.9, 0.9
    opt = torch.optim.SGD([
        {"params": ema_utils.get_params(net, key="1x"), "lr": lr, "weight_decay": wd},
        {"params": ema_utils.get_params(net, key="1y"), "lr": lr, "weight_decay": 0},
        {"params": ema_utils.get_params(net, key="2x"), "lr": 2 * lr, "weight_decay": 0.0}], momentum=mom)
    out = {"cfg": np.array([50, 3, 7, 8, 2, 256])}
    # label maps: forward without labels on the initial state (batch-statistics BN: with running stats at their
    # init values an eval-mode forward of a random net overflows to 1e4-sized logits and says nothing)
    with torch.no_grad():
        batch = E.seeded_batch(2, 256, 3, seed=10)
        pred = net(batch["images"])
        out["pred_slice"] = pred[:, :, ::16, ::16].numpy()
        out["pred_labels"] = net.predict_classes(batch["images"]).numpy().astype(np.uint8)
        top2 = pred.topk(2, dim=1).values
        out["pred_margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    net.load_state_dict(E.seeded_state_dict(50, 3, seed=7, residual_scale=residual_scale), strict=True)  # undo the running-stat updates
    for it in range(2):
        batch = E.seeded_batch(2, 256, 3, seed=8 + it)
        loss, mu = net(batch["images"], torch.squeeze(batch["segmented"], dim=1))
        with torch.no_grad():
            mu_mean = mu.mean(dim=0, keepdim=True)
            net.emau.mu *= em_mom
            net.emau.mu += mu_mean * (1 - em_mom)
        total = loss.mean()
        opt.zero_grad()
        total.backward()
        if it == 0:
            out["loss"] = loss.detach().numpy()
            out["mu_slice"] = mu.detach()[:, ::32, ::8].numpy()
            out["mu_abs_sum"] = mu.detach().abs().sum().double().numpy()
            names = [n for n, p in net.named_parameters()]
            out["grad_norms"] = np.array([-1.0 if p.grad is None else p.grad.double().norm().item()
                                          for _, p in net.named_parameters()])  # -1: no gradient reaches it
            out["grad_names"] = np.array(names)
            out["grad_fc2_weight"] = net.fc2.weight.grad.numpy()
            out["grad_stem0_slice"] = net.extractor[0][0].weight.grad[::8].numpy()
            out["grad_l4"] = net.extractor[7][2].conv3.weight.grad[::64, ::16].numpy().copy()
        opt.step()
        out[f"loss_mean_{it}"] = total.detach().numpy()
    sd = net.state_dict()
    out["after_names"] = np.array(list(sd.keys()))
    out["after_abs_sums"] = np.array([v.double().abs().sum().item() for v in sd.values()])
    out["after_emau_mu_slice"] = sd["emau.mu"][0, ::32, ::8].numpy()
    init = E.seeded_state_dict(50, 3, seed=7, residual_scale=residual_scale)
    for k in ("fc2.weight", "fc2.bias", "fc1.0.bn.weight"):  # two-step parameter deltas
        out["delta_" + k] = (sd[k] - init[k]).numpy()
    out["delta_layer4_conv3_slice"] = (sd["extractor.7.2.conv3.weight"] - init["extractor.7.2.conv3.weight"])[::64, ::16].numpy().copy()
    out["delta_stem0_slice"] = (sd["extractor.0.0.weight"] - init["extractor.0.0.weight"])[::8].numpy().copy()
    out["after_bn_running_var_fc0"] = sd["fc0.bn.running_var"].numpy()
    out["grad_layer4_conv3_slice"] = np.zeros(1) if "grad_l4" not in out else out.pop("grad_l4")
    np.savez_compressed(os.path.join(HERE, out_name), **out)


def make_trans_u_net(vit, tu_utils):
    from oracle import trans_u_net_ref as T
    cfg = vit.VIT_CONFIGS["R50-ViT-B_16"]
    cfg.n_classes, cfg.n_skip = 3, 3
    cfg.patches.grid = (14, 14)
    cfg.transformer.dropout_rate = 0.0  # stochastic layers off (embedding / MLP dropout 0.1 in the shipped config)
    net = vit.VisionTransformer(cfg, img_size=224, num_classes=3)
    schema = T.state_dict_schema(224, 3)
    assert [k for k, _ in schema] == list(net.state_dict().keys())
    for (k, s), v in zip(schema, net.state_dict().values()):
        assert tuple(s) == tuple(v.shape), k
    net.load_state_dict(T.seeded_state_dict(224, 3, seed=17), strict=True)
    net.train()
    # lr / momentum / weight decay of configs/segmenter/stylegan2_trans_u_net_segmenter.yaml:17-19 except lr
    # (1e-4 instead of 0.01 for the same conditioning reason as the EMANet fixture)
    opt = torch.optim.SGD(net.parameters(), lr=1e-4, momentum=0.9, weight_decay=1e-4)
    ce_loss, dice = torch.nn.CrossEntropyLoss(), tu_utils.DiceLoss(3)
    out = {"cfg": np.array([224, 3, 17, 18, 2])}
    for it in range(2):
        batch = E.seeded_batch(2, 224, 3, seed=18 + it)
        opt.zero_grad()
        pred = net(batch["images"])
        gt = torch.squeeze(batch["segmented"], dim=1)
        loss_ce = ce_loss(pred, gt.long())
        loss_dice = dice(pred, gt, softmax=True)
        loss = 0.5 * loss_ce + 0.5 * loss_dice
        loss.backward()
        if it == 0:
            out["losses"] = np.array([loss.item(), loss_ce.item(), loss_dice.item()])
            out["logits_slice"] = pred.detach()[:, :, ::8, ::8].numpy()
            out["labels"] = pred.detach().argmax(1).numpy().astype(np.uint8)
            top2 = pred.detach().topk(2, dim=1).values
            out["margin"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
            out["grad_names"] = np.array([n for n, _ in net.named_parameters()])
            out["grad_norms"] = np.array([p.grad.double().norm().item() for _, p in net.named_parameters()])
            out["grad_head"] = net.segmentation_head[0].weight.grad.numpy()
        opt.step()
        out[f"loss_{it}"] = np.array(loss.item())
    sd = net.state_dict()
    init = T.seeded_state_dict(224, 3, seed=17)
    for k in ("segmentation_head.0.weight", "segmentation_head.0.bias", "decoder.blocks.3.conv2.1.weight",
              "transformer.encoder.encoder_norm.weight"):
        out["delta_" + k] = (sd[k] - init[k]).numpy()
    np.savez_compressed(os.path.join(HERE, "trans_u_net_step.npz"), **out)


if __name__ == "__main__":
    assert load_reference.reference_available()
    ema, vit, tu_utils, ema_utils = load_reference.load_reference_segmenters()
    torch.set_num_threads(8)
    make_ema_net(ema, ema_utils)
    make_ema_net(ema, ema_utils, residual_scale=0.1, lr=0.009, out_name="ema_net_step_conditioned.npz")
    make_trans_u_net(vit, tu_utils)
    for f in sorted(os.listdir(HERE)):
        if f.endswith("_step.npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")
