"""EMANet builder (reference: training_builder/ema_net_train_builder.py:12-59): network, the three-group SGD
(conv weights: lr + weight decay; BN scales: lr; all biases: 2*lr) and the updater."""
from typing import Dict

from torch.optim import Optimizer

from networks.ema_net.network import EMANet
from networks.ema_net.utils import get_params
from training.fused_sgd import FusedSGD
from training_builder.base_train_builder import BaseSingleNetworkTrainBuilder, strip_parallel_module
from updater.segmentation_updater import EMANetUpdater


class EMANetTrainBuilder(BaseSingleNetworkTrainBuilder):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._initialize_segmentation_network()
        self.find_unused_params = True  # emau.conv1 never receives a gradient (see networks/ema_net/network.py)
        self.segmentation_network = self._prepare_segmentation_network(self.segmentation_network)

    def _initialize_segmentation_network(self):
        use_pretrained_resnet = self.config.get('use_pretrained_resnet', self.config.get('fine_tune') is None)
        self.segmentation_network = EMANet(
            self.config['num_classes'], self.config['n_layers'], use_pretrained_resnet=use_pretrained_resnet,
            pretrained_path=self.config.get('pretrained_path') if use_pretrained_resnet else None)

    def get_optimizers(self) -> Dict[str, Optimizer]:
        if self._optimizers is None:
            net = strip_parallel_module(self.segmentation_network)
            lr, wd = self.config['lr'], self.config['weight_decay']
            self._optimizers = {'main': FusedSGD([
                {'params': list(get_params(net, key='1x')), 'lr': lr, 'weight_decay': wd},
                {'params': list(get_params(net, key='1y')), 'lr': lr, 'weight_decay': 0},
                {'params': list(get_params(net, key='2x')), 'lr': 2 * lr, 'weight_decay': 0.0},
            ], momentum=self.config['lr_mom'])}
        return self._optimizers

    def get_updater(self) -> EMANetUpdater:
        return EMANetUpdater(em_mom=self.config['em_mom'], iterators={'images': self.train_data_loader},
                             networks=self.get_networks_for_updater(), optimizers=self.get_optimizers(),
                             device=self.device(), copy_to_device=(self.world_size == 1))
