"""Import of the Google ImageNet-21k ViT / R50+ViT ``.npz`` checkpoints into the TransUNet modules.

Same key mapping as the ``load_from`` methods of the reference (vit_seg_modeling.py:192-230, :401-448 and
vit_seg_modeling_resnet_skip.py:77-111), expressed as tables: JAX kernels are HWIO / [in, out] and are transposed to
torch's OIHW / [out, in]; position embeddings are resized bilinearly (``scipy.ndimage.zoom(order=1)``) when the token
grid differs, after dropping the class token.
"""
import logging
from os.path import join as pjoin

import numpy as np
import torch

logger = logging.getLogger(__name__)

_ATTN = {"query": "MultiHeadDotProductAttention_1/query", "key": "MultiHeadDotProductAttention_1/key",
         "value": "MultiHeadDotProductAttention_1/value", "out": "MultiHeadDotProductAttention_1/out"}
_MLP = {"fc1": "MlpBlock_3/Dense_0", "fc2": "MlpBlock_3/Dense_1"}
_NORM = {"attention_norm": "LayerNorm_0", "ffn_norm": "LayerNorm_2"}


def _t(array, conv=False):
    return torch.from_numpy(array.transpose([3, 2, 0, 1]) if conv else array)


@torch.no_grad()
def load_encoder_block(block, weights, n_block):
    root, hs = f"Transformer/encoderblock_{n_block}", block.hidden_size
    for name, key in _ATTN.items():
        lin = getattr(block.attn, name)
        lin.weight.copy_(_t(weights[pjoin(root, key, "kernel")]).view(hs, hs).t())
        lin.bias.copy_(_t(weights[pjoin(root, key, "bias")]).view(-1))
    for name, key in _MLP.items():
        lin = getattr(block.ffn, name)
        lin.weight.copy_(_t(weights[pjoin(root, key, "kernel")]).t())
        lin.bias.copy_(_t(weights[pjoin(root, key, "bias")]).t())
    for name, key in _NORM.items():
        ln = getattr(block, name)
        ln.weight.copy_(_t(weights[pjoin(root, key, "scale")]))
        ln.bias.copy_(_t(weights[pjoin(root, key, "bias")]))


@torch.no_grad()
def load_bottleneck(unit, weights, n_block, n_unit):
    def w(name, conv=False):
        return _t(weights[pjoin(n_block, n_unit, name)], conv=conv)

    for i in (1, 2, 3):
        getattr(unit, f"conv{i}").weight.copy_(w(f"conv{i}/kernel", conv=True))
        getattr(unit, f"gn{i}").weight.copy_(w(f"gn{i}/scale").view(-1))
        getattr(unit, f"gn{i}").bias.copy_(w(f"gn{i}/bias").view(-1))
    if hasattr(unit, "downsample"):
        unit.downsample.weight.copy_(w("conv_proj/kernel", conv=True))
        unit.gn_proj.weight.copy_(w("gn_proj/scale").view(-1))
        unit.gn_proj.bias.copy_(w("gn_proj/bias").view(-1))


def _resized_position_embeddings(posemb, target, classifier):
    if posemb.size() == target.size():
        return posemb
    if posemb.size(1) - 1 == target.size(1):
        return posemb[:, 1:]
    from scipy import ndimage
    grid = posemb[0, 1:] if classifier == "seg" else posemb[0]
    gs_old, gs_new = int(np.sqrt(len(grid))), int(np.sqrt(target.size(1)))
    logger.info("load_pretrained: grid-size from %s to %s", gs_old, gs_new)
    zoomed = ndimage.zoom(grid.reshape(gs_old, gs_old, -1).numpy(), (gs_new / gs_old, gs_new / gs_old, 1), order=1)
    return _t(zoomed.reshape(1, gs_new * gs_new, -1))


@torch.no_grad()
def load_vision_transformer(model, weights):
    emb, enc = model.transformer.embeddings, model.transformer.encoder
    emb.patch_embeddings.weight.copy_(_t(weights["embedding/kernel"], conv=True))
    emb.patch_embeddings.bias.copy_(_t(weights["embedding/bias"]))
    enc.encoder_norm.weight.copy_(_t(weights["Transformer/encoder_norm/scale"]))
    enc.encoder_norm.bias.copy_(_t(weights["Transformer/encoder_norm/bias"]))
    emb.position_embeddings.copy_(_resized_position_embeddings(
        _t(weights["Transformer/posembed_input/pos_embedding"]), emb.position_embeddings, model.classifier))
    for uname, unit in enc.layer.named_children():
        load_encoder_block(unit, weights, uname)
    if emb.hybrid:
        root = emb.hybrid_model.root
        root.conv.weight.copy_(_t(weights["conv_root/kernel"], conv=True))
        root.gn.weight.copy_(_t(weights["gn_root/scale"]).view(-1))
        root.gn.bias.copy_(_t(weights["gn_root/bias"]).view(-1))
        for bname, block in emb.hybrid_model.body.named_children():
            for uname, unit in block.named_children():
                load_bottleneck(unit, weights, bname, uname)
