"""TransUNet: (hybrid ResNetV2 +) ViT encoder, cascaded-upsampling CNN decoder, 3x3 segmentation head.

Drop-in for /root/reference/stylegan_code_finder/networks/trans_u_net/vit_seg_modeling.py: same class names
and constructor signatures (Attention :53, Mlp :100, Embeddings :125, Block :171, Encoder :233,
Transformer :253, Conv2dReLU :265, DecoderBlock :290, SegmentationHead :324, DecoderCup :332,
VisionTransformer :376, VIT_CONFIGS :456), the same parameter names (409 state_dict entries for R50-ViT-B_16,
checked against the reference's key list in the test-suite) and the ``load_from(npz)`` weight import.

Execution differences: attention runs through ``scaled_dot_product_attention`` (no [B,12,N,N] score tensor in
HBM; the reference's attention dropout rate is 0.0, so this is the same function) unless attention maps are
requested (``vis``); dense layers / convolutions / norms go to the ROCm libraries through ATen for now.
"""
import copy
import logging
import math
from os.path import join as pjoin

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Conv2d, Dropout, LayerNorm, Linear, Softmax
from torch.nn.modules.utils import _pair

from . import vit_seg_configs as configs
from .vit_seg_configs import ConfigDict
from .vit_seg_modeling_resnet_skip import ResNetV2, np2th
from ..base_segmenter import BaseSegmenter

logger = logging.getLogger(__name__)

_NPZ = {"q": "MultiHeadDotProductAttention_1/query", "k": "MultiHeadDotProductAttention_1/key",
        "v": "MultiHeadDotProductAttention_1/value", "o": "MultiHeadDotProductAttention_1/out",
        "fc0": "MlpBlock_3/Dense_0", "fc1": "MlpBlock_3/Dense_1", "ln_attn": "LayerNorm_0", "ln_mlp": "LayerNorm_2"}


def swish(x):
    return x * torch.sigmoid(x)


ACT2FN = {"gelu": F.gelu, "relu": F.relu, "swish": swish}


class Attention(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.num_attention_heads = config.transformer["num_heads"]
        self.attention_head_size = int(config.hidden_size / self.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = Linear(config.hidden_size, self.all_head_size)
        self.key = Linear(config.hidden_size, self.all_head_size)
        self.value = Linear(config.hidden_size, self.all_head_size)
        self.out = Linear(config.hidden_size, config.hidden_size)
        self.attn_dropout = Dropout(config.transformer["attention_dropout_rate"])
        self.proj_dropout = Dropout(config.transformer["attention_dropout_rate"])
        self.softmax = Softmax(dim=-1)

    def transpose_for_scores(self, x):
        b, n, _ = x.shape
        return x.view(b, n, self.num_attention_heads, self.attention_head_size).permute(0, 2, 1, 3)

    def forward(self, hidden_states):
        q = self.transpose_for_scores(self.query(hidden_states))
        k = self.transpose_for_scores(self.key(hidden_states))
        v = self.transpose_for_scores(self.value(hidden_states))
        weights = None
        if self.vis or (self.training and self.attn_dropout.p > 0):
            scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(self.attention_head_size)
            probs = self.softmax(scores)
            weights = probs if self.vis else None
            context = torch.matmul(self.attn_dropout(probs), v)
        else:
            context = F.scaled_dot_product_attention(q, k, v)
        b, _, n, _ = context.shape
        context = context.permute(0, 2, 1, 3).reshape(b, n, self.all_head_size)
        return self.proj_dropout(self.out(context)), weights


class Mlp(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.fc1 = Linear(config.hidden_size, config.transformer["mlp_dim"])
        self.fc2 = Linear(config.transformer["mlp_dim"], config.hidden_size)
        self.act_fn = ACT2FN["gelu"]
        self.dropout = Dropout(config.transformer["dropout_rate"])
        for fc in (self.fc1, self.fc2):
            nn.init.xavier_uniform_(fc.weight)
            nn.init.normal_(fc.bias, std=1e-6)

    def forward(self, x):
        return self.dropout(self.fc2(self.dropout(self.act_fn(self.fc1(x)))))


class Embeddings(nn.Module):
    """Patch + position embeddings; in hybrid mode the "patches" are 1x1 (or p x p) cells of the ResNetV2
    stride-16 feature map and the stem's intermediate maps are returned as decoder skips."""

    def __init__(self, config, img_size, in_channels=3):
        super().__init__()
        self.config = config
        img_size = _pair(img_size)
        if config.patches.get("grid") is not None:
            grid = config.patches["grid"]
            patch_size = (img_size[0] // 16 // grid[0], img_size[1] // 16 // grid[1])
            real = (patch_size[0] * 16, patch_size[1] * 16)
            n_patches = (img_size[0] // real[0]) * (img_size[1] // real[1])
            self.hybrid = True
        else:
            patch_size = _pair(config.patches["size"])
            n_patches = (img_size[0] // patch_size[0]) * (img_size[1] // patch_size[1])
            self.hybrid = False
        if self.hybrid:
            self.hybrid_model = ResNetV2(block_units=config.resnet.num_layers, width_factor=config.resnet.width_factor)
            in_channels = self.hybrid_model.width * 16
        self.patch_embeddings = Conv2d(in_channels=in_channels, out_channels=config.hidden_size,
                                       kernel_size=patch_size, stride=patch_size)
        self.position_embeddings = nn.Parameter(torch.zeros(1, n_patches, config.hidden_size))
        self.dropout = Dropout(config.transformer["dropout_rate"])

    def forward(self, x):
        features = None
        if self.hybrid:
            x, features = self.hybrid_model(x)
        x = self.patch_embeddings(x).flatten(2).transpose(-1, -2)  # [B, n_patches, hidden]
        return self.dropout(x + self.position_embeddings), features


class Block(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.hidden_size = config.hidden_size
        self.attention_norm = LayerNorm(config.hidden_size, eps=1e-6)
        self.ffn_norm = LayerNorm(config.hidden_size, eps=1e-6)
        self.ffn = Mlp(config)
        self.attn = Attention(config, vis)

    def forward(self, x):
        a, weights = self.attn(self.attention_norm(x))
        x = x + a
        return x + self.ffn(self.ffn_norm(x)), weights

    def load_from(self, weights, n_block):
        root = f"Transformer/encoderblock_{n_block}"
        hs = self.hidden_size

        def arr(key, leaf):
            return np2th(weights[pjoin(root, _NPZ[key], leaf)])

        with torch.no_grad():
            for key, lin in (("q", self.attn.query), ("k", self.attn.key), ("v", self.attn.value), ("o", self.attn.out)):
                lin.weight.copy_(arr(key, "kernel").view(hs, hs).t())
                lin.bias.copy_(arr(key, "bias").view(-1))
            for key, lin in (("fc0", self.ffn.fc1), ("fc1", self.ffn.fc2)):
                lin.weight.copy_(arr(key, "kernel").t())
                lin.bias.copy_(arr(key, "bias").t())
            for key, ln in (("ln_attn", self.attention_norm), ("ln_mlp", self.ffn_norm)):
                ln.weight.copy_(arr(key, "scale"))
                ln.bias.copy_(arr(key, "bias"))


class Encoder(nn.Module):
    def __init__(self, config, vis):
        super().__init__()
        self.vis = vis
        self.layer = nn.ModuleList()
        self.encoder_norm = LayerNorm(config.hidden_size, eps=1e-6)
        prototype = Block(config, vis)
        for _ in range(config.transformer["num_layers"]):
            self.layer.append(copy.deepcopy(prototype))

    def forward(self, hidden_states):
        attn_weights = []
        for block in self.layer:
            hidden_states, weights = block(hidden_states)
            if self.vis:
                attn_weights.append(weights)
        return self.encoder_norm(hidden_states), attn_weights


class Transformer(nn.Module):
    def __init__(self, config, img_size, vis):
        super().__init__()
        self.embeddings = Embeddings(config, img_size=img_size)
        self.encoder = Encoder(config, vis)

    def forward(self, input_ids):
        embedding_output, features = self.embeddings(input_ids)
        encoded, attn_weights = self.encoder(embedding_output)
        return encoded, attn_weights, features


class Conv2dReLU(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size, padding=0, stride=1, use_batchnorm=True):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding,
                                   bias=not use_batchnorm),
                         nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))


class DecoderBlock(nn.Module):
    def __init__(self, in_channels, out_channels, skip_channels=0, use_batchnorm=True):
        super().__init__()
        self.conv1 = Conv2dReLU(in_channels + skip_channels, out_channels, kernel_size=3, padding=1,
                                use_batchnorm=use_batchnorm)
        self.conv2 = Conv2dReLU(out_channels, out_channels, kernel_size=3, padding=1, use_batchnorm=use_batchnorm)
        self.up = nn.UpsamplingBilinear2d(scale_factor=2)

    def forward(self, x, skip=None):
        x = self.up(x)
        if skip is not None:
            x = torch.cat([x, skip], dim=1)
        return self.conv2(self.conv1(x))


class SegmentationHead(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel_size=3, upsampling=1):
        super().__init__(nn.Conv2d(in_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2),
                         nn.UpsamplingBilinear2d(scale_factor=upsampling) if upsampling > 1 else nn.Identity())


class DecoderCup(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        head_channels = 512
        self.conv_more = Conv2dReLU(config.hidden_size, head_channels, kernel_size=3, padding=1, use_batchnorm=True)
        decoder_channels = config.decoder_channels
        in_channels = [head_channels] + list(decoder_channels[:-1])
        if self.config.n_skip != 0:
            skip_channels = self.config.skip_channels
            for i in range(4 - self.config.n_skip):  # unused skips contribute no channels
                skip_channels[3 - i] = 0
        else:
            skip_channels = [0, 0, 0, 0]
        self.blocks = nn.ModuleList([DecoderBlock(i, o, s) for i, o, s in zip(in_channels, decoder_channels,
                                                                              skip_channels)])

    def forward(self, hidden_states, features=None):
        b, n_patch, hidden = hidden_states.size()
        h = w = int(np.sqrt(n_patch))
        x = self.conv_more(hidden_states.permute(0, 2, 1).contiguous().view(b, hidden, h, w))
        for i, block in enumerate(self.blocks):
            skip = features[i] if (features is not None and i < self.config.n_skip) else None
            x = block(x, skip=skip)
        return x


class VisionTransformer(BaseSegmenter):
    def __init__(self, config: ConfigDict, img_size: int = 224, num_classes: int = 21843, zero_head: bool = False,
                 vis: bool = False, background_class_id: int = 0, min_confidence: float = 0.0,
                 min_contour_area: int = 0):
        super().__init__(background_class_id, min_confidence, min_contour_area)
        self.num_classes = num_classes
        self.zero_head = zero_head
        self.classifier = config.classifier
        self.transformer = Transformer(config, img_size, vis)
        self.decoder = DecoderCup(config)
        self.segmentation_head = SegmentationHead(in_channels=config['decoder_channels'][-1],
                                                  out_channels=config['n_classes'], kernel_size=3)
        self.config = config

    def forward(self, x):
        if x.size(1) == 1:
            x = x.repeat(1, 3, 1, 1)
        x, _, features = self.transformer(x)
        return self.segmentation_head(self.decoder(x, features))

    def predict_classes(self, x: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.forward(x), dim=1, keepdim=True)

    def load_from(self, weights):
        """Imports a Google ViT / R50+ViT ``.npz`` checkpoint (position embeddings resized bilinearly when the
        token grid differs, as the reference does with scipy.ndimage.zoom(order=1))."""
        emb = self.transformer.embeddings
        with torch.no_grad():
            emb.patch_embeddings.weight.copy_(np2th(weights["embedding/kernel"], conv=True))
            emb.patch_embeddings.bias.copy_(np2th(weights["embedding/bias"]))
            self.transformer.encoder.encoder_norm.weight.copy_(np2th(weights["Transformer/encoder_norm/scale"]))
            self.transformer.encoder.encoder_norm.bias.copy_(np2th(weights["Transformer/encoder_norm/bias"]))
            posemb = np2th(weights["Transformer/posembed_input/pos_embedding"])
            target = emb.position_embeddings
            if posemb.size() == target.size():
                target.copy_(posemb)
            elif posemb.size(1) - 1 == target.size(1):
                target.copy_(posemb[:, 1:])
            else:
                from scipy import ndimage
                grid = posemb[0, 1:] if self.classifier == "seg" else posemb[0]
                gs_old, gs_new = int(np.sqrt(len(grid))), int(np.sqrt(target.size(1)))
                logger.info("load_pretrained: grid-size from %s to %s", gs_old, gs_new)
                grid = ndimage.zoom(grid.reshape(gs_old, gs_old, -1).numpy(), (gs_new / gs_old, gs_new / gs_old, 1), order=1)
                target.copy_(np2th(grid.reshape(1, gs_new * gs_new, -1)))
            for uname, unit in self.transformer.encoder.layer.named_children():
                unit.load_from(weights, n_block=uname)
            if emb.hybrid:
                root = emb.hybrid_model.root
                root.conv.weight.copy_(np2th(weights["conv_root/kernel"], conv=True))
                root.gn.weight.copy_(np2th(weights["gn_root/scale"]).view(-1))
                root.gn.bias.copy_(np2th(weights["gn_root/bias"]).view(-1))
                for bname, block in emb.hybrid_model.body.named_children():
                    for uname, unit in block.named_children():
                        unit.load_from(weights, n_block=bname, n_unit=uname)


VIT_CONFIGS = {
    'ViT-B_16': configs.get_b16_config(),
    'ViT-B_32': configs.get_b32_config(),
    'ViT-L_16': configs.get_l16_config(),
    'ViT-L_32': configs.get_l32_config(),
    'ViT-H_14': configs.get_h14_config(),
    'R50-ViT-B_16': configs.get_r50_b16_config(),
    'R50-ViT-L_16': configs.get_r50_l16_config(),
    'testing': configs.get_testing(),
}
