"""TransUNet declaration for the shared train builder (reference: training_builder/trans_u_net_train_builder.py:12-51):
ViT configuration by name with classes / skips / token grid taken from the training config, optional ``.npz``
import of the ImageNet-21k weights, plain SGD(lr, momentum, weight_decay) over every parameter."""
import numpy

from networks.trans_u_net.vit_seg_modeling import VIT_CONFIGS, VisionTransformer
from training_builder.base_train_builder import BaseTrainBuilder
from updater.segmentation_updater import TransUNetUpdater


class TransUNetTrainBuilder(BaseTrainBuilder):
    updater_class = TransUNetUpdater

    def build_network(self):
        cfg = self.config
        vit = VIT_CONFIGS[cfg['pretrained_model_name']].copy()  # never mutate the shared table
        vit.n_classes, vit.n_skip = cfg['num_classes'], cfg['num_skip_channels']
        tokens_per_side = cfg['image_size'] // cfg['vit_patch_size']
        if vit.patches.get('grid') is not None:
            vit.patches.grid = (tokens_per_side, tokens_per_side)
        network = VisionTransformer(vit, img_size=cfg['image_size'], num_classes=vit.n_classes)
        if cfg.get('fine_tune') is None and cfg.get('pretrained_path'):
            network.load_from(weights=numpy.load(cfg['pretrained_path']))
        return network

    def optimizer_defaults(self):
        return {k: self.config[k] for k in ('lr', 'momentum', 'weight_decay')}

    def updater_options(self):
        return {'num_classes': self.config['num_classes'], 'amp': self.config.get('amp')}
