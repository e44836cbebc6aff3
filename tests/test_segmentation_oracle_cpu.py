"""CPU suite, part 3: the segmentation-training oracle is pinned against golden outputs of the reference
EMANet run through one-and-a-half updater iterations (tests/golden/make_golden_seg.py)."""
import os

import numpy as np
import torch

from oracle import ema_net_ref as E


def test_ema_net_schema_counts():
    schema = E.state_dict_schema(50, 3)
    assert len(schema) == 353
    n_params = sum(int(np.prod(s)) for n, s in schema if "running_" not in n and not n.endswith("tracked") and n != "emau.mu")
    assert n_params == 34776771
    sd = E.seeded_state_dict(50, 3, seed=1)
    g1x, g1y, g2x = E.param_groups(sd)
    assert (len(g1x), len(g1y), len(g2x)) == (60, 58, 60)


def test_ema_net_train_step_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ema_net_step.npz"))
    n_layers, classes, wseed, bseed, batch, size = g["cfg"].tolist()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sd = E.seeded_state_dict(n_layers, classes, seed=wseed)
    bufs = {}
    total, loss, mu, grads = E.train_step(sd, bufs, E.seeded_batch(batch, size, classes, seed=bseed), lr=2e-5)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=2e-5)
    np.testing.assert_allclose(total.numpy(), g["loss_mean_0"], rtol=2e-5)
    np.testing.assert_allclose(mu[:, ::32, ::8].numpy(), g["mu_slice"], rtol=1e-4, atol=1e-6)
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        if ref < 0:
            assert grads[str(name)] is None
        else:
            np.testing.assert_allclose(grads[str(name)].double().norm().item(), ref, rtol=2e-3, err_msg=str(name))
    np.testing.assert_allclose(grads["fc2.weight"].numpy(), g["grad_fc2_weight"], rtol=1e-3, atol=1e-6)
    total1, _, _, _ = E.train_step(sd, bufs, E.seeded_batch(batch, size, classes, seed=bseed + 1), lr=2e-5)
    np.testing.assert_allclose(total1.numpy(), g["loss_mean_1"], rtol=1e-4)
    for name, ref in zip(g["after_names"], g["after_abs_sums"]):
        np.testing.assert_allclose(sd[str(name)].double().abs().sum().item(), ref, rtol=1e-5, err_msg=str(name))
    np.testing.assert_allclose(sd["emau.mu"][0, ::32, ::8].numpy(), g["after_emau_mu_slice"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(sd["fc0.bn.running_var"].numpy(), g["after_bn_running_var_fc0"], rtol=1e-4)
    init = E.seeded_state_dict(n_layers, classes, seed=wseed)
    for k in ("fc2.weight", "fc2.bias", "fc1.0.bn.weight"):
        ref = g["delta_" + k]
        np.testing.assert_allclose((sd[k] - init[k]).numpy(), ref, rtol=1e-2, atol=1e-3 * np.abs(ref).max(), err_msg=k)


def test_ema_net_label_maps_vs_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ema_net_step.npz"))
    n_layers, classes, wseed, bseed, batch, size = g["cfg"].tolist()
    sd = E.seeded_state_dict(n_layers, classes, seed=wseed)
    with torch.no_grad():
        pred, _ = E.forward(sd, E.seeded_batch(batch, size, classes, seed=10)["images"], training=True)
    np.testing.assert_allclose(pred[:, :, ::16, ::16].numpy(), g["pred_slice"], rtol=1e-3, atol=1e-4)
    labels = pred.argmax(1, keepdim=True).numpy().astype(np.uint8)
    margin = g["pred_margin"].astype(np.float32)
    decided = margin > 1e-3  # argmax must be bit-exact wherever the reference's top-2 margin is not a rounding tie
    assert (labels[:, 0][decided] == g["pred_labels"][:, 0][decided]).all()
    assert decided.mean() > 0.98


def test_trans_u_net_schema_counts():
    from oracle import trans_u_net_ref as T
    schema = T.state_dict_schema(224, 3)
    assert len(schema) == 409
    n_params = sum(int(np.prod(s)) for n, s in schema if "running_" not in n and not n.endswith("tracked"))
    assert n_params == 105276211  # R50-ViT-B_16 @224, 3 classes (105.9 M at 512^2 incl. larger pos-emb)


def test_trans_u_net_train_step_vs_golden(golden_dir):
    from oracle import trans_u_net_ref as T
    g = np.load(os.path.join(golden_dir, "trans_u_net_step.npz"))
    size, classes, wseed, bseed, batch = g["cfg"].tolist()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sd = T.seeded_state_dict(size, classes, seed=wseed)
    init = {k: v.clone() for k, v in sd.items()}
    bufs = {}
    loss, ce, dice, grads, logits = T.train_step(sd, bufs, E.seeded_batch(batch, size, classes, seed=bseed), lr=1e-4)
    np.testing.assert_allclose([loss.item(), ce.item(), dice.item()], g["losses"], rtol=2e-5)
    np.testing.assert_allclose(logits[:, :, ::8, ::8].numpy(), g["logits_slice"], rtol=1e-3, atol=1e-4)
    decided = g["margin"].astype(np.float32) > 1e-3
    assert (logits.argmax(1).numpy().astype(np.uint8)[decided] == g["labels"][decided]).all() and decided.mean() > 0.98
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        np.testing.assert_allclose(grads[str(name)].double().norm().item(), ref, rtol=2e-3, atol=1e-9, err_msg=str(name))
    np.testing.assert_allclose(grads["segmentation_head.0.weight"].numpy(), g["grad_head"], rtol=1e-3,
                               atol=1e-4 * np.abs(g["grad_head"]).max())
    loss1, *_ = T.train_step(sd, bufs, E.seeded_batch(batch, size, classes, seed=bseed + 1), lr=1e-4)
    np.testing.assert_allclose(loss1.item(), g["loss_1"], rtol=1e-3)
    for k in ("segmentation_head.0.weight", "segmentation_head.0.bias", "decoder.blocks.3.conv2.1.weight",
              "transformer.encoder.encoder_norm.weight"):
        ref = g["delta_" + k]
        np.testing.assert_allclose((sd[k] - init[k]).numpy(), ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max(), err_msg=k)


def test_ema_net_conditioned_fixture_vs_golden(golden_dir):
    """The well-conditioned fixture (bn3 x 0.1 -- oracle/ema_net_ref.py::seeded_state_dict) at the SHIPPED learning rate
    0.009: the oracle against the reference's own two iterations (tests/golden/ema_net_step_conditioned.npz)."""
    g = np.load(os.path.join(golden_dir, "ema_net_step_conditioned.npz"))
    n_layers, classes, wseed, bseed, batch, size = g["cfg"].tolist()
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    sd = E.seeded_state_dict(n_layers, classes, seed=wseed, residual_scale=0.1)
    bufs = {}
    total, loss, mu, grads = E.train_step(sd, bufs, E.seeded_batch(batch, size, classes, seed=bseed), lr=0.009)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=2e-5)
    for name, ref in zip(g["grad_names"], g["grad_norms"]):
        if ref >= 0:
            np.testing.assert_allclose(grads[str(name)].double().norm().item(), ref, rtol=1e-3, err_msg=str(name))
    np.testing.assert_allclose(grads["fc2.weight"].numpy(), g["grad_fc2_weight"], rtol=1e-3, atol=1e-6)
    total1, _, _, _ = E.train_step(sd, bufs, E.seeded_batch(batch, size, classes, seed=bseed + 1), lr=0.009)
    np.testing.assert_allclose(total1.numpy(), g["loss_mean_1"], rtol=1e-4)
    init = E.seeded_state_dict(n_layers, classes, seed=wseed, residual_scale=0.1)
    for k in ("fc2.weight", "fc2.bias", "fc1.0.bn.weight"):
        ref = g["delta_" + k]
        np.testing.assert_allclose((sd[k] - init[k]).numpy(), ref, rtol=1e-2, atol=1e-3 * np.abs(ref).max(), err_msg=k)
