"""CPU suite, part 6: the training entry path (reference: train.py:39-147, utils/clamped_cosine.py:8-19,
training_builder/{ema_net,trans_u_net}_train_builder.py, networks/__init__.py:22-41,415-423, vit_seg_modeling.py:401-448).

Nothing here launches a kernel: schedules, config -> builder -> optimizer wiring, the .npz weight import against a
fixture produced by the reference's own ``load_from``, the epoch-boundary restart of a finite loader, and the
generator factories under their reference names."""
import argparse
import ast
import importlib.util
import math
import os
import sys

import numpy as np
import pytest
import torch
import yaml
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "synthesis-in-style_amd")
REF = "/root/reference/stylegan_code_finder"


def _sgd(lrs):
    return torch.optim.SGD([{"params": [nn.Parameter(torch.zeros(1))], "lr": lr} for lr in lrs], lr=0.1, momentum=0.9)


def test_clamped_cosine_closed_form():
    from utils.clamped_cosine import ClampedCosineAnnealingLR
    opt = _sgd([0.009, 0.018])
    sched = ClampedCosineAnnealingLR(opt, 20, eta_min=1e-8)
    assert [g["lr"] for g in opt.param_groups] == [0.009, 0.018]  # t = 0 after construction
    for t in range(1, 30):
        sched.step()
        for base, group in zip((0.009, 0.018), opt.param_groups):
            want = 1e-8 + (base - 1e-8) * (1 + math.cos(math.pi * t / 20)) / 2 if t <= 20 else 1e-8
            assert group["lr"] == pytest.approx(want, rel=1e-12, abs=0)
    assert opt.param_groups[0]["lr"] == 1e-8


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_clamped_cosine_equals_the_reference_class():
    """The reference class (utils/clamped_cosine.py:8-19) lifted with ``ast`` (its module imports the absent
    ``pytorch_training``) on top of torch's own CosineAnnealingLR (recursive form) vs the closed form used here."""
    from torch.optim.lr_scheduler import CosineAnnealingLR
    from utils.clamped_cosine import ClampedCosineAnnealingLR
    path = os.path.join(REF, "utils", "clamped_cosine.py")
    node = next(n for n in ast.parse(open(path).read()).body if isinstance(n, ast.ClassDef) and n.name == "ClampedCosineAnnealingLR")

    class Base(CosineAnnealingLR):  # the reference forwards ``verbose`` (torch 1.9); torch >= 2.7 dropped the argument
        def __init__(self, optimizer, T_max, eta_min=0, last_epoch=-1, verbose=False):
            super().__init__(optimizer, T_max, eta_min, last_epoch)

    scope = {"CosineAnnealingLR": Base}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), scope)
    a, b = _sgd([0.009, 0.018]), _sgd([0.009, 0.018])
    ref = scope["ClampedCosineAnnealingLR"](a, 15, eta_min=1e-8)
    mine = ClampedCosineAnnealingLR(b, 15, eta_min=1e-8)
    for _ in range(25):
        a.step(), ref.step(), mine.step()
        for ga, gb in zip(a.param_groups, b.param_groups):
            assert gb["lr"] == pytest.approx(ga["lr"], rel=1e-9, abs=1e-15)


def test_get_scheduler_end_iteration_and_warm_restarts():
    import train
    opt = {"main": _sgd([0.01])}
    assert train.get_scheduler({"cosine_max_update_epoch": 2, "end_lr": 1e-8, "epochs": 4}, 1000, opt)["main"].T_max == 2000
    assert train.get_scheduler({"cosine_max_update_iter": 77, "epochs": 4}, 1000, {"main": _sgd([0.01])})["main"].T_max == 77
    assert train.get_scheduler({"epochs": 4}, 1000, {"main": _sgd([0.01])})["main"].T_max == 4
    warm = train.get_scheduler({"epochs": 4, "warm_restarts": True, "end_lr": 1e-6}, 10, {"main": _sgd([0.01])})["main"]
    assert isinstance(warm, torch.optim.lr_scheduler.CosineAnnealingWarmRestarts) and warm.T_0 == 4 and warm.eta_min == 1e-6


def _config(name):
    cfg = yaml.safe_load(open(os.path.join(SRC, "configs", "segmenter", name)))
    cfg["fine_tune"] = None
    return cfg


def test_ema_net_train_builder_groups_and_updater_wiring():
    from training.fused_sgd import FusedSGD
    from training_builder.train_builder_selection import get_train_builder_class
    from training_builder.ema_net_train_builder import EMANetTrainBuilder
    from updater.segmentation_updater import EMANetUpdater
    from utils.synthetic_data import SyntheticSegmentationLoader
    cfg = _config("ema_net_resnet50_256.yaml")
    assert get_train_builder_class(cfg) is EMANetTrainBuilder
    loader = SyntheticSegmentationLoader(2, 32, 3, num_batches=3, distinct=1)
    builder = EMANetTrainBuilder(cfg, loader, None, rank=0, world_size=1)
    opt = builder.get_optimizers()["main"]
    assert opt is builder.get_optimizers()["main"] and isinstance(opt, FusedSGD)  # ONE optimizer: the scheduler's is the trainer's
    groups = opt.param_groups
    assert [len(g["params"]) for g in groups] == [60, 58, 60]  # conv weights / BN scales / biases (ema_net/utils.py:7-21)
    assert [g["lr"] for g in groups] == [0.009, 0.009, 0.018]
    assert [g["weight_decay"] for g in groups] == [1e-4, 0.0, 0.0]
    assert all(g["momentum"] == 0.9 for g in groups)
    net = builder.get_network()
    n_learnable = sum(1 for _ in net.parameters())
    assert sum(len(g["params"]) for g in groups) == n_learnable
    updater = builder.get_updater()
    assert isinstance(updater, EMANetUpdater) and updater.em_mom == 0.9
    assert updater.networks["segmentation"] is net and updater.optimizers["main"] is opt
    assert builder.find_unused_params is True


def test_trans_u_net_train_builder_config_and_optimizer():
    from training_builder.train_builder_selection import get_train_builder_class
    from training_builder.trans_u_net_train_builder import TransUNetTrainBuilder
    from networks.trans_u_net.vit_seg_modeling import VIT_CONFIGS
    from updater.segmentation_updater import TransUNetUpdater
    cfg = _config("trans_u_net_r50_vit_b16_512.yaml")
    assert get_train_builder_class(cfg) is TransUNetTrainBuilder
    grid_before = tuple(VIT_CONFIGS["R50-ViT-B_16"].patches.grid)
    builder = TransUNetTrainBuilder(cfg, [], None, rank=0, world_size=1)
    net = builder.get_network()
    assert tuple(VIT_CONFIGS["R50-ViT-B_16"].patches.grid) == grid_before  # the shared table is not mutated
    assert net.config.patches.grid == (32, 32) and net.config.n_classes == 3 and net.config.n_skip == 3
    assert net.transformer.embeddings.position_embeddings.shape == (1, 1024, 768)
    assert len(net.state_dict()) == 409
    opt = builder.get_optimizers()["main"]
    assert len(opt.param_groups) == 1 and len(opt.param_groups[0]["params"]) == sum(1 for _ in net.parameters())
    g = opt.param_groups[0]
    assert (g["lr"], g["momentum"], g["weight_decay"]) == (0.01, 0.9, 1e-4)
    updater = builder.get_updater()
    assert isinstance(updater, TransUNetUpdater) and updater.amp_dtype == torch.bfloat16
    assert updater.dice_loss.n_classes == 3


def test_npz_import_matches_the_references_load_from(golden_dir):
    spec = importlib.util.spec_from_file_location("make_golden_npz_import", os.path.join(golden_dir, "make_golden_npz_import.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    from networks.trans_u_net import vit_seg_configs
    from networks.trans_u_net.vit_seg_modeling import VisionTransformer
    torch.manual_seed(0)
    net = VisionTransformer(mk.shrink(vit_seg_configs.get_r50_b16_config()), img_size=mk.IMG, num_classes=3)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_from(mk.synthetic_checkpoint())
    g = np.load(os.path.join(golden_dir, "npz_import.npz"))
    names, sums = mk.checksums(net.state_dict())
    assert names == g["names"].tolist()
    touched = np.array([not torch.equal(before[k], v) for k, v in net.state_dict().items()])
    assert np.array_equal(touched, g["touched"])  # decoder / head keep their initialisation, like the reference's
    np.testing.assert_allclose(sums[touched], g["sums"][touched], rtol=1e-12, atol=1e-9)


class _FakeCudaUpdater:
    pass


def test_updater_restarts_a_finite_loader_at_the_epoch_boundary():
    """``train.py`` runs epochs * iterations_per_epoch updates over a loader that yields
    iterations_per_epoch batches; ``next_batch`` re-creates the iterator like the reference's per-epoch trainer."""
    from training.loop import Updater
    from utils.synthetic_data import SyntheticSegmentationLoader
    loader = SyntheticSegmentationLoader(2, 8, 3, num_batches=3, distinct=3)
    seen = []

    class Probe(Updater):
        def update_core(self):
            seen.append(self.next_batch('images')['images'][0, 0, 0, 0].item())

    up = Probe({'images': loader}, {}, {}, device='cpu')
    for _ in range(8):
        up.update()
    assert up.iteration == 8 and seen[:3] == seen[3:6] and seen[:2] == seen[6:8] and len(set(seen[:3])) == 3
    with pytest.raises(RuntimeError, match="yields no batches"):
        Probe({'images': []}, {}, {}, device='cpu').update()


def test_generator_factories_under_the_reference_names(tmp_path):
    import networks
    from networks.stylegan2.model import Generator
    g = networks.get_stylegan2_generator(16, 32, n_mlp=2, channel_multiplier=1)
    assert isinstance(g, Generator) and g.size == 16 and g.style_dim == 32
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"g_ema": g.state_dict(), "g": {}}, ckpt)
    g2 = networks.get_stylegan2_generator(16, 32, n_mlp=2, channel_multiplier=1, init_ckpt=str(ckpt))
    assert all(torch.equal(a, b) for a, b in zip(g.state_dict().values(), g2.state_dict().values()))
    with pytest.raises(RuntimeError):  # strict: a generator of another width does not load
        networks.get_stylegan2_generator(16, 32, n_mlp=3, channel_multiplier=1, init_ckpt=str(ckpt))
    cfg = {"stylegan_variant": 2, "image_size": 16, "latent_size": 32, "input_dim": 3, "n_mlp": 2, "channel_multiplier": 1}
    auto = networks.load_autoencoder_or_generator(argparse.Namespace(device="cpu", checkpoint=str(ckpt)), cfg)
    assert isinstance(auto.decoder, Generator)
    assert all(torch.equal(a, b) for a, b in zip(g.state_dict().values(), auto.decoder.state_dict().values()))
    with pytest.raises(NotImplementedError):
        networks.load_autoencoder_or_generator(argparse.Namespace(device="cpu", checkpoint=str(ckpt)),
                                               {**cfg, "stylegan_checkpoint": "x"})
    plain = tmp_path / "plain.pt"  # load_weights without a key takes the dict as the state_dict (networks/__init__.py:26-28)
    torch.save(g.state_dict(), plain)
    networks.load_weights(g2, plain, key="g_ema")


def test_bench_gpus_flag_launches_ranks_or_rejects_a_mismatch():
    """`python bench.py --gpus N` must start N ranks itself (no launcher in the driver's command) and fail loudly when it
    cannot: here, without a HIP device, every rank raises, and the launcher's exit code comes back non-zero; a WORLD_SIZE
    that disagrees with --gpus is refused before anything runs."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=dict(env, WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--workload", "synthesis"],
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "local_rank: 1" in r.stderr  # two ranks were started, both failed loudly


def test_bench_line_stays_compact_and_carries_the_training_legs(tmp_path, capsys, monkeypatch):
    """bench.py's `--workload all` line (host logic only, no GPU): the driver's parsed record keeps `config`, `roofline` and
    `cpu_baseline` as flat objects and the END of stdout, so the training legs must appear (1) as flat ``seg_<leg>_*`` keys in
    those objects, (2) as the compact nested `seg_train` object at the very end of the line; per-kernel tables go to the
    detail file and the whole line stays below 6 KB."""
    import argparse
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    table = {f"kernel_{i}<128,256,{i}>": {"launches": i, "ms": 0.1 * i, "nominal_tflops": 1.0, "gbs_algorithmic": 2.0,
                                          "bound": "mfma", "frac_of_peak": 0.5} for i in range(120)}

    def leg(value, dtype, cpu):
        return {"value": value, "ms_per_step": 27.0, "dtype": dtype, "steps": 20, "warmup": 5, "scaling": "weak",
                "config": {"workload": "w" * 150, "batch_per_gpu": 16, "image_size": 256, "hip_graph": True},
                "roofline": {"bound": "mfma", "achieved": 86.0, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.55, "algorithmic_frac": 0.85,
                             "dominant_own_kernel": "conv1x1_f32_kernel<128,256,16>", "traffic": 1.0e8,
                             "own_kernels_eager_iteration": dict(table)},
                "cpu_baseline": cpu and {"value": 1.5, "unit": "images/s", "cores": 16, "kind": "port", "batch": 4, "sample": "s" * 200},
                "library_calls_per_step": {"fallback": {}, "intended": {"stem": 2}},
                "library_ms_per_step": {"own_ms": 26.5, "library_ms": 0.34, "top_library_kernels_ms": {"x" * 60: 0.3}},
                "data_parallel_rehearsal": {"graph_ms_per_step": 27.2, "eager_ms_per_step": 27.3, "direct_rccl": True}}

    result = {"metric": bench.METRIC, "value": 2100.0, "unit": "images/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 15.2,
              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
              "config": {"workload": "synthesis", "batch_per_gpu": 32},
              "roofline": {"kernel": "modconv_wino2_kernel", "bound": "mfma", "achieved": 118.0, "peak": 157.3, "unit": "TFLOP/s",
                           "frac": 0.75, "traffic": 1.7e9, "kernels": dict(table)},
              "cpu_baseline": {"value": 2.2, "unit": "images/s", "cores": 16, "kind": "port", "sample": "b4"}}
    seg = {"emanet": leg(600.0, "f32", True), "transunet_bf16": leg(370.0, "bf16", True), "transunet_f32": leg(94.0, "f32", False)}
    bench.attach_seg_train(result, seg)
    bench.emit(result, argparse.Namespace(workload="all"))
    line = capsys.readouterr().out.strip()
    assert "\n" not in line and len(line) < 6144, len(line)
    r = json.loads(line)
    assert list(r)[-1] == "seg_train"                      # the tail of stdout is the compact summary
    assert r["value"] == 2100.0 and "kernels" not in r["roofline"]
    assert r["config"]["seg_emanet_images_per_s"] == 600.0 and r["config"]["seg_transunet_bf16_images_per_s"] == 370.0
    assert r["roofline"]["seg_transunet_bf16_frac"] == 0.55 and r["roofline"]["seg_emanet_dominant_own_kernel"].startswith("conv1x1")
    assert r["cpu_baseline"]["seg_emanet_value"] == 1.5 and "seg_transunet_f32_value" not in r["cpu_baseline"]
    for obj in (r["config"], r["roofline"], r["cpu_baseline"]):   # flat: nothing the driver's parser would drop
        assert not any(isinstance(v, (dict, list)) for v in obj.values())
    s = r["seg_train"]["transunet_bf16"]
    assert s["images_per_s"] == 370.0 and s["roofline"]["frac"] == 0.55 and s["cpu_baseline"]["value"] == 1.5
    assert s["library_ms_per_step"] == 0.34 and s["dp_rehearsal_graph_ms_per_step"] == 27.2 and s["hip_graph"] is True
    detail = json.load(open(tmp_path / "gpurun_out" / "bench_detail_all.json"))
    assert len(detail["seg_train"]["emanet"]["roofline"]["own_kernels_eager_iteration"]) == 120
    assert len(detail["roofline_tables"]["kernels"]) == 120
