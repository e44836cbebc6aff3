"""Generates tests/golden/npz_import.npz: what the UNMODIFIED reference ``VisionTransformer.load_from``
(networks/trans_u_net/vit_seg_modeling.py:401-448, PreActBottleneck.load_from / Block.load_from behind it) makes of a
synthetic ImageNet-21k style ``.npz`` checkpoint, on a shrunken hybrid configuration (R50 trunk with one unit per
stage + 2 ViT blocks of width 64, 64x64 input) so that the fixture stays small.  The checkpoint is re-derived anywhere
from a frozen numpy stream (``synthetic_checkpoint``); stored are two fp64 checksums per state_dict entry.

The position embedding of the checkpoint has a class token and a 3x3 grid, the model a 4x4 grid: the bilinear
``ndimage.zoom`` resize path of load_from is exercised.

Run in the build container, in its own process:  python tests/golden/make_golden_npz_import.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

IMG, HIDDEN, MLP, HEADS, LAYERS, UNITS = 64, 64, 128, 4, 2, (1, 1, 1)


def shrink(config):
    """Applied alike to the reference's and the product's ``get_r50_b16_config()``."""
    config.hidden_size = HIDDEN
    config.transformer.mlp_dim, config.transformer.num_heads, config.transformer.num_layers = MLP, HEADS, LAYERS
    config.transformer.dropout_rate = 0.0
    config.resnet.num_layers = UNITS
    config.patches.grid = (IMG // 16, IMG // 16)
    config.n_classes, config.n_skip = 3, 3
    return config


def synthetic_checkpoint(seed=5):
    """JAX-layout arrays (HWIO convolution kernels, [in, heads, dim] attention kernels) under the checkpoint's keys."""
    rs = np.random.RandomState(seed)
    w = {}

    def put(key, *shape):
        w[key] = rs.standard_normal(shape).astype(np.float32)

    put("embedding/kernel", 1, 1, 1024, HIDDEN)
    put("embedding/bias", HIDDEN)
    put("Transformer/encoder_norm/scale", HIDDEN)
    put("Transformer/encoder_norm/bias", HIDDEN)
    put("Transformer/posembed_input/pos_embedding", 1, 1 + 9, HIDDEN)
    hd = HIDDEN // HEADS
    for i in range(LAYERS):
        root = f"Transformer/encoderblock_{i}"
        for ln in ("LayerNorm_0", "LayerNorm_2"):
            put(f"{root}/{ln}/scale", HIDDEN)
            put(f"{root}/{ln}/bias", HIDDEN)
        for name in ("query", "key", "value"):
            put(f"{root}/MultiHeadDotProductAttention_1/{name}/kernel", HIDDEN, HEADS, hd)
            put(f"{root}/MultiHeadDotProductAttention_1/{name}/bias", HEADS, hd)
        put(f"{root}/MultiHeadDotProductAttention_1/out/kernel", HEADS, hd, HIDDEN)
        put(f"{root}/MultiHeadDotProductAttention_1/out/bias", HIDDEN)
        put(f"{root}/MlpBlock_3/Dense_0/kernel", HIDDEN, MLP)
        put(f"{root}/MlpBlock_3/Dense_0/bias", MLP)
        put(f"{root}/MlpBlock_3/Dense_1/kernel", MLP, HIDDEN)
        put(f"{root}/MlpBlock_3/Dense_1/bias", HIDDEN)
    put("conv_root/kernel", 7, 7, 3, 64)
    put("gn_root/scale", 64)
    put("gn_root/bias", 64)
    cin = 64
    for b, n_units in enumerate(UNITS):
        cout, cmid = 256 * 2 ** b, 64 * 2 ** b
        for u in range(1, n_units + 1):
            root = f"block{b + 1}/unit{u}"
            for i, (kk, a, c) in enumerate(((1, cin if u == 1 else cout, cmid), (3, cmid, cmid), (1, cmid, cout)), start=1):
                put(f"{root}/conv{i}/kernel", kk, kk, a, c)
                put(f"{root}/gn{i}/scale", 1, 1, 1, c)
                put(f"{root}/gn{i}/bias", 1, 1, 1, c)
            if u == 1:
                put(f"{root}/conv_proj/kernel", 1, 1, cin, cout)
                put(f"{root}/gn_proj/scale", 1, 1, 1, cout)
                put(f"{root}/gn_proj/bias", 1, 1, 1, cout)
        cin = cout
    return w


def checksums(state_dict):
    names = list(state_dict.keys())
    sums = np.array([[v.double().sum().item(), (v.double() * torch.arange(1, v.numel() + 1, dtype=torch.float64)
                                                .reshape(v.shape) / v.numel()).sum().item()] for v in state_dict.values()])
    return names, sums


if __name__ == "__main__":
    from oracle import load_reference
    _, vit, _, _ = load_reference.load_reference_segmenters()
    import importlib
    ref_cfg = importlib.import_module("networks.trans_u_net.vit_seg_configs")
    torch.manual_seed(0)
    net = vit.VisionTransformer(shrink(ref_cfg.get_r50_b16_config()), img_size=IMG, num_classes=3)
    before = {k: v.clone() for k, v in net.state_dict().items()}
    net.load_from(synthetic_checkpoint())
    after = net.state_dict()
    names, sums = checksums(after)
    touched = np.array([not torch.equal(before[k], after[k]) for k in names])
    np.savez_compressed(os.path.join(HERE, "npz_import.npz"), names=np.array(names), sums=sums, touched=touched)
    print(f"{len(names)} entries, {int(touched.sum())} written by load_from")
