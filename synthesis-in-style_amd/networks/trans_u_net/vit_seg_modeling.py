"""TransUNet: (hybrid ResNetV2 +) ViT encoder, cascaded-upsampling CNN decoder, 3x3 segmentation head.

Drop-in module path for /root/reference/stylegan_code_finder/networks/trans_u_net/vit_seg_modeling.py: every public
class of that file is importable from here (Attention, Mlp, Embeddings, Block, Encoder, Transformer, Conv2dReLU,
DecoderBlock, SegmentationHead, DecoderCup, VisionTransformer, VIT_CONFIGS) with the same constructor signatures and
parameter names (409 state_dict entries for R50-ViT-B_16, checked against the reference's key list in the tests).
The code itself is organised by role: ``vit_encoder.py`` (embeddings + transformer), ``cup_decoder.py`` (decoder +
head), ``npz_import.py`` (ImageNet-21k ``.npz`` weight import), this file (the assembled segmenter + config table).
"""
import torch

from . import vit_seg_configs as configs
from .cup_decoder import Conv2dReLU, DecoderBlock, DecoderCup, SegmentationHead, decoder_step_state  # noqa: F401
from .vit_encoder import ACT2FN, Attention, Block, Embeddings, Encoder, Mlp, Transformer, swish  # noqa: F401
from .vit_seg_configs import ConfigDict
from .vit_seg_modeling_resnet_skip import np2th  # noqa: F401
from ..base_segmenter import BaseSegmenter


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
        with decoder_step_state(self):   # one launch packs every decoder / head weight, one bumps every batch-norm counter
            return self.segmentation_head(self.decoder(x, features))

    def predict_classes(self, x: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.forward(x), dim=1, keepdim=True)

    def load_from(self, weights):
        """Imports a Google ViT / R50+ViT ``.npz`` checkpoint (networks/trans_u_net/npz_import.py)."""
        from .npz_import import load_vision_transformer
        load_vision_transformer(self, weights)


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
