"""GPU: the training entry point end to end (reference: train.py:69-147) -- ``train.main`` on synthetic batches for a
few iterations across an epoch boundary: config file -> builder -> updater -> per-iteration LR schedule -> snapshot."""
import math
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "synthesis-in-style_amd")


def _run(tmp_path, name, overrides, max_iter):
    import train
    from training.loop import get_current_reporter
    cfg = yaml.safe_load(open(os.path.join(SRC, "configs", "segmenter", name)))
    cfg.update(overrides)
    path = tmp_path / name
    path.write_text(yaml.safe_dump(cfg))
    args = train.parse_args([str(path), "--synthetic", "--max-iter", str(max_iter), "-l", str(tmp_path / "logs")])
    get_current_reporter().observations.clear()
    train.main(0, args, 1)
    return get_current_reporter().scalars()


def test_train_main_ema_net_crosses_an_epoch_boundary(device, tmp_path):
    # 2 iterations per epoch, 5 iterations: the finite loader is restarted twice; iteration 3 captures the step graph
    obs = _run(tmp_path, "ema_net_resnet50_256.yaml",
               dict(batch_size=2, image_size=64, iterations_per_epoch=2, epochs=3, snapshot_save_iter=4, log_iter=1), 5)
    assert math.isfinite(obs["loss/softmax"]) and obs["loss/softmax"] > 0
    ckpt = torch.load(tmp_path / "logs" / "000004.pt", map_location="cpu")
    assert set(ckpt) == {"segmentation_network", "main"} and len(ckpt["segmentation_network"]) == 353
    assert "momentum_buffer" in next(iter(ckpt["main"]["state"].values()))


def test_train_main_trans_u_net_bf16(device, tmp_path):
    obs = _run(tmp_path, "trans_u_net_r50_vit_b16_512.yaml",
               dict(batch_size=2, image_size=64, iterations_per_epoch=2, epochs=2, log_iter=1), 4)
    assert all(math.isfinite(obs[k]) for k in ("loss/combined", "loss/CE", "loss/Dice"))
    assert 0 < obs["loss/Dice"] < 1
