"""Every run-time switch of DESIGN.md §0.4 in its NON-default position: one small iteration of the workload the switch belongs to
must give the loss and the parameter gradients of the default position (the two positions are two implementations of the same
arithmetic: own kernel vs library operator, fused vs unfused launch, one stream vs two).  Tolerances: fp32 workloads 2e-4 on the
loss / 2 % on every gradient norm; TransUNet under bf16 autocast 2e-3 / 12 % (positions that change WHERE values are rounded to
bf16 -- encoder blocks module by module -- move the small GroupNorm gradients of the trunk's first units by 5-6 %).  The default positions themselves are
what the parity tests against the oracle run on; this file is what keeps the A/B legs of `profiles/` runnable.

Switches read by the C library once per process (`static const … getenv`) cannot be flipped inside a test process; they are
flipped in a child process that reruns one parity test of the kernel they belong to (`test_c_side_switch_in_a_child_process`)."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import ema_net_ref as E
from oracle import stylegan2_ref as R
from oracle import trans_u_net_ref as T

pytestmark = pytest.mark.gpu

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ema_net_step(device):
    from networks.ema_net.network import EMANet
    from training.fused_sgd import FusedSGD
    from training.loop import get_current_reporter
    from updater.segmentation_updater import EMANetUpdater
    net = EMANet(3, 50, use_pretrained_resnet=False)
    net.load_state_dict(E.seeded_state_dict(50, 3, seed=41), strict=True)
    net.fc1[1].p = 0.0
    net = net.to(device).train()
    opt = FusedSGD(list(net.parameters()), lr=1e-6, momentum=0.9, weight_decay=1e-4)
    upd = EMANetUpdater(em_mom=0.9, iterators={"images": [E.seeded_batch(2, 128, 3, seed=42)]}, networks={"segmentation": net},
                        optimizers={"main": opt}, device=device, hip_graph=False)
    upd.update()
    torch.cuda.synchronize()
    return get_current_reporter().scalars()["loss/softmax"], {n: p.grad.double().norm().item() for n, p in net.named_parameters()
                                                               if p.grad is not None}


def _trans_u_net_step(device):
    from networks.trans_u_net.vit_seg_modeling import VIT_CONFIGS, VisionTransformer
    from training.fused_sgd import FusedSGD
    from training.loop import get_current_reporter
    from updater.segmentation_updater import TransUNetUpdater
    size, classes = 224, 3
    cfg = VIT_CONFIGS["R50-ViT-B_16"].copy()
    cfg.n_classes, cfg.n_skip = classes, 3
    cfg.patches.grid = (size // 16, size // 16)
    cfg.transformer.dropout_rate = 0.0
    net = VisionTransformer(cfg, img_size=size, num_classes=classes)
    sd = T.seeded_state_dict(size, classes, seed=43)
    for k in sd:
        if ".gn3." in k:        # the conditioned regime of test_trans_u_net_gpu.py::_vit_like
            sd[k] = sd[k] * 0.1
    net.load_state_dict(sd, strict=True)
    net = net.to(device).train()
    opt = FusedSGD(list(net.parameters()), lr=1e-6, momentum=0.9, weight_decay=1e-4)
    upd = TransUNetUpdater(num_classes=classes, amp="bf16", iterators={"images": [E.seeded_batch(2, size, classes, seed=44)]},
                           networks={"segmentation": net}, optimizers={"main": opt}, device=device, hip_graph=False)
    upd.update()
    torch.cuda.synchronize()
    return get_current_reporter().scalars()["loss/combined"], {n: p.grad.double().norm().item() for n, p in net.named_parameters()
                                                                if p.grad is not None}


def _generator_images(device):
    from networks.stylegan2.model import Generator
    g = Generator(64, 64, 2, channel_multiplier=2)
    g.load_state_dict(R.seeded_state_dict(64, 64, 2, 2, seed=45), strict=True)
    g = g.to(device).eval()
    z, noise = R.seeded_inputs(64, 4, 64, seed=46)
    with torch.no_grad():
        img, _ = g([z.to(device)], noise=[n.to(device) for n in noise])
    torch.cuda.synchronize()
    return float(img.double().abs().mean()), {"image": img.double().norm().item(),
                                              "image_first": img[0].double().norm().item(), "image_last": img[-1].double().norm().item()}


_WORKLOADS = {"ema_net": (_ema_net_step, 2e-4, 2e-2), "trans_u_net": (_trans_u_net_step, 2e-3, 0.12), "generator": (_generator_images, 1e-5, 1e-5)}
_DEFAULT = {}


def _default(workload, device):
    if workload not in _DEFAULT:
        _DEFAULT[workload] = _WORKLOADS[workload][0](device)
    return _DEFAULT[workload]


def _agree(workload, got, ref):
    _, loss_rtol, grad_rtol = _WORKLOADS[workload]
    np.testing.assert_allclose(got[0], ref[0], rtol=loss_rtol)
    assert got[1].keys() == ref[1].keys()
    floor = 1e-5 * max(ref[1].values())   # gradients that are zero in exact arithmetic (an attention key bias: softmax is shift invariant)
    for name, norm in ref[1].items():
        assert abs(got[1][name] - norm) <= grad_rtol * norm + floor, (name, got[1][name], norm)


# (module, attribute, non-default value, workload): module-level switches, read from the environment at import
_MODULE_SWITCHES = [
    ("networks.hip_conv", "_F32_POINTWISE", False, "ema_net"),            # SIS_F32_POINTWISE
    ("networks.hip_conv", "_FUSE_SKIP_GRAD", False, "ema_net"),           # SIS_FUSE_SKIP_GRAD
    ("networks.hip_conv", "_HALF_DIL_OWN", False, "ema_net"),             # SIS_HALF_DIL_OWN
    ("networks.hip_conv", "_STRIDE2_OWN", False, "ema_net"),              # SIS_STRIDE2_OWN
    ("networks.ema_net.network", "_HIP_EMAU", False, "ema_net"),          # SIS_HIP_EMAU
    ("networks.ema_net.network", "_RELU_MASK", False, "ema_net"),         # SIS_BN_RELU_MASK
    ("networks.ema_net.network", "_SUB_IMAGE_UNITS", False, "ema_net"),   # SIS_SUB_IMAGE_UNITS
    ("networks.ema_net.network", "_WINO_BANK", False, "ema_net"),         # SIS_WINO_BANK
    ("networks.hip_conv", "_BF16_CONV", False, "trans_u_net"),            # SIS_BF16_CONV
    ("networks.hip_conv", "_PW_WGRAD_OWN", False, "trans_u_net"),         # SIS_PW_WGRAD_OWN
    ("networks.hip_conv", "_STRIDE2_OWN", False, "trans_u_net"),
    ("networks.trans_u_net.cup_decoder", "_FUSE_UP_CAT", False, "trans_u_net"),                   # SIS_FUSE_UP_CAT
    ("networks.trans_u_net.cup_decoder", "_DECODER_BANK", False, "trans_u_net"),                  # SIS_DECODER_BANK
    ("networks.trans_u_net.vit_encoder", "_HIP_LN", False, "trans_u_net"),                        # SIS_HIP_LN
    ("networks.trans_u_net.vit_encoder", "_AMP_LINEAR", False, "trans_u_net"),                    # SIS_AMP_LINEAR
    ("networks.trans_u_net.vit_encoder", "_SHADOW", False, "trans_u_net"),                        # SIS_LINEAR_SHADOW
    ("networks.trans_u_net.vit_encoder", "_FUSED_BLOCK", False, "trans_u_net"),                   # SIS_FUSED_VIT
    ("networks.trans_u_net.vit_encoder", "_GEMM256", False, "trans_u_net"),                       # SIS_GEMM256
    ("networks.trans_u_net.vit_encoder", "_GEMM256_DGRAD", True, "trans_u_net"),                  # SIS_GEMM256_DGRAD
    ("networks.trans_u_net.vit_encoder", "_FUSE_BIAS_GRAD", False, "trans_u_net"),                # SIS_FUSE_BIAS_GRAD
    ("networks.trans_u_net.vit_encoder", "_FUSE_BLOCK_CAST", False, "trans_u_net"),               # SIS_FUSE_BLOCK_CAST
    ("networks.trans_u_net.vit_encoder", "_WGRAD_SIDE", 1, "trans_u_net"),                        # SIS_WGRAD_STREAM
    ("networks.trans_u_net.vit_encoder", "_WGRAD_SIDE", 2, "trans_u_net"),
    ("networks.trans_u_net.vit_seg_modeling_resnet_skip", "_GN_GATE_BITS", False, "trans_u_net"),  # SIS_GN_GATE_BITS
    ("networks.trans_u_net.vit_seg_modeling_resnet_skip", "_WS_BANK", False, "trans_u_net"),       # SIS_WS_BANK
    ("networks.trans_u_net.vit_seg_modeling_resnet_skip", "_SAMPLE_POINTWISE", False, "trans_u_net"),  # SIS_SAMPLE_POINTWISE_S2
    ("networks.trans_u_net.vit_seg_modeling_resnet_skip", "_STEM_OWN", False, "trans_u_net"),          # SIS_STEM_OWN
    ("networks.trans_u_net.vit_seg_modeling_resnet_skip", "_FUSE_RESIDUAL", False, "trans_u_net"),  # SIS_GN_RES
    ("networks.trans_u_net.vit_seg_modeling_resnet_skip", "_DUAL_STREAM", False, "trans_u_net"),   # SIS_GN_DUAL
    ("updater.segmentation_updater", "_FUSED_LOSS", False, "trans_u_net"),                        # SIS_FUSED_LOSS
    ("sis_hip", "_GN_FUSED_FINISH", False, "trans_u_net"),                                        # SIS_GN_FUSED_FINISH
    ("sis_hip", "_GEMM256_WIDTHS", (288, 192, 96), "trans_u_net"),                                # SIS_GEMM256_TILES
    ("sis_hip", "_UP_FIR", False, "generator"),                                                   # SIS_UP_FIR
    ("sis_hip", "_DEFER", False, "trans_u_net"),                                                  # SIS_DEFER_REDUCES
    ("sis_hip", "_DEFER_WGRAD", False, "trans_u_net"),                                            # SIS_DEFER_WGRAD
    ("sis_hip", "_DEFER_WGRAD", False, "ema_net"),
]


@pytest.mark.parametrize("module,attr,value,workload", _MODULE_SWITCHES,
                         ids=[f"{m.rsplit('.', 1)[-1]}.{a}={v}-{w}" for m, a, v, w in _MODULE_SWITCHES])
def test_module_switch_non_default_position(device, monkeypatch, module, attr, value, workload):
    ref = _default(workload, device)
    mod = importlib.import_module(module)
    assert getattr(mod, attr) != value, "the table lists NON-default positions"
    monkeypatch.setattr(mod, attr, value)
    _agree(workload, _WORKLOADS[workload][0](device), ref)


# switches read from the environment at call time (Python side)
_ENV_SWITCHES = [
    ("SIS_WINOGRAD", "0", "generator"),
    ("SIS_RGB_STREAM", "0", "generator"),
    ("SIS_BN_SINGLE_PASS", "0", "ema_net"),      # csrc/bn_ops.hip reads it per call
    ("SIS_BN_WIDE", "0", "ema_net"),             # (the shapes it steers: tests/test_seg_ops_gpu.py::test_batch_norm_wide_single_pass)
]


@pytest.mark.parametrize("name,value,workload", _ENV_SWITCHES, ids=[f"{n}={v}" for n, v, _ in _ENV_SWITCHES])
def test_environment_switch_non_default_position(device, monkeypatch, name, value, workload):
    ref = _default(workload, device)
    monkeypatch.setenv(name, value)
    _agree(workload, _WORKLOADS[workload][0](device), ref)


# switches the C library reads ONCE per process: (variable, value, parity test of the kernel it steers)
_C_SIDE = [
    ("SIS_WINO_PIPE", "0", "tests/test_generator_gpu.py::test_generator_vs_golden_small"),
    ("SIS_WINO_TPW", "4", "tests/test_generator_gpu.py::test_generator_vs_golden_small"),
    ("SIS_WINO_XCD_MB", "0", "tests/test_generator_gpu.py::test_generator_vs_golden_small"),
    ("SIS_UPFIR_WAVES", "8", "tests/test_generator_gpu.py::test_modconv_up_fir_vs_oracle"),   # the one-workgroup-per-CU tile ...
    ("SIS_UPFIR_PIPE", "0", "tests/test_generator_gpu.py::test_modconv_up_fir_vs_oracle"),    # ... and its unpipelined loop
    ("SIS_WGRAD_WAVES", "4", "tests/test_conv_bf16_gpu.py::test_conv_bf16_weight_gradient"),   # 64 x 64 tiles, two workgroups per CU, everywhere
    ("SIS_WGRAD_NARROW_KS", "0", "tests/test_conv_bf16_gpu.py::test_conv_bf16_weight_gradient"),   # (with SIS_WGRAD_WAVES=4 above: the narrow layers' single-width strips)
    ("SIS_GN_SINGLE_PASS", "0", "tests/test_upsample_gpu.py"),
    ("SIS_UP2_DIRECT", "0", "tests/test_upsample_gpu.py"),
    ("SIS_PW_KC", "32", "tests/test_conv1x1_f32_gpu.py"),
    ("SIS_BN_BWD512", "0", "tests/test_seg_ops_gpu.py"),
    ("SIS_KMEANS_FAST", "0", "tests/test_dataset_ops_gpu.py"),
]


# a second value of a variable above (its own child): the 4-wave fast-FIR workgroups with the loop that is not pipelined across the barrier
_C_SIDE_2 = [
    ("SIS_UPFIR_PIPE", "1", "tests/test_generator_gpu.py::test_modconv_up_fir_vs_oracle"),
]


@pytest.mark.parametrize("switches", [_C_SIDE, _C_SIDE_2], ids=["set1", "set2"])
def test_c_side_switch_in_a_child_process(device, switches):
    """One child process per switch would cost one GPU context each; the switches steer different kernels, so ONE child sets
    them all and reruns the parity tests of those kernels."""
    env = dict(os.environ)
    for name, value, _ in switches:
        env[name] = value
    targets = sorted({t for _, _, t in switches})
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-p", "no:cacheprovider", *targets], cwd=_REPO, env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
