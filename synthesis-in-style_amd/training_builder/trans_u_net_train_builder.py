"""TransUNet builder (reference: training_builder/trans_u_net_train_builder.py:12-51): picks the ViT config,
sets classes / skips / token grid from the training config, SGD(lr, momentum, weight_decay) over all parameters."""
from typing import Dict

import numpy
from torch.optim import Optimizer

from networks.trans_u_net.vit_seg_modeling import VIT_CONFIGS, VisionTransformer
from training.fused_sgd import FusedSGD
from training_builder.base_train_builder import BaseSingleNetworkTrainBuilder, strip_parallel_module
from updater.segmentation_updater import TransUNetUpdater


class TransUNetTrainBuilder(BaseSingleNetworkTrainBuilder):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._initialize_segmentation_network()
        self.segmentation_network = self._prepare_segmentation_network(self.segmentation_network)
        self.optimizer_opts = {'lr': self.config['lr'], 'momentum': self.config['momentum'],
                               'weight_decay': self.config['weight_decay']}

    def _initialize_segmentation_network(self):
        transformer_config = VIT_CONFIGS[self.config['pretrained_model_name']].copy()
        transformer_config.n_classes = self.config['num_classes']
        transformer_config.n_skip = self.config['num_skip_channels']
        patch = self.config['vit_patch_size']
        if transformer_config.patches.get('grid') is not None:
            transformer_config.patches.grid = (self.config['image_size'] // patch, self.config['image_size'] // patch)
        network = VisionTransformer(transformer_config, img_size=self.config['image_size'],
                                    num_classes=transformer_config.n_classes)
        if self.config.get('fine_tune') is None and self.config.get('pretrained_path'):
            network.load_from(weights=numpy.load(self.config['pretrained_path']))
        self.segmentation_network = network

    def get_optimizers(self) -> Dict[str, Optimizer]:
        if self._optimizers is None:
            params = list(strip_parallel_module(self.segmentation_network).parameters())
            self._optimizers = {'main': FusedSGD(params, **self.optimizer_opts)}
        return self._optimizers

    def get_updater(self) -> TransUNetUpdater:
        return TransUNetUpdater(num_classes=self.config['num_classes'], amp=self.config.get('amp'),
                                iterators={'images': self.train_data_loader},
                                networks=self.get_networks_for_updater(), optimizers=self.get_optimizers(),
                                device=self.device(), copy_to_device=(self.world_size == 1))
