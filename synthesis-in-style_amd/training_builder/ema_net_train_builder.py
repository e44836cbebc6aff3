"""EMANet declaration for the shared train builder (reference: training_builder/ema_net_train_builder.py:12-59).

SGD with three groups, as networks/ema_net/utils.py:7-21 splits them: convolution weights (lr, weight decay),
batch-norm scales (lr, no decay), all biases (2*lr, no decay); momentum ``lr_mom``.  ``find_unused_params`` because
``emau.conv1`` never receives a gradient (its output only feeds the no-grad EM iterations)."""
from networks.ema_net.network import EMANet
from networks.ema_net.utils import get_params
from training_builder.base_train_builder import BaseTrainBuilder
from updater.segmentation_updater import EMANetUpdater


class EMANetTrainBuilder(BaseTrainBuilder):
    find_unused_params = True
    updater_class = EMANetUpdater

    def build_network(self):
        cfg = self.config
        pretrained = cfg.get('use_pretrained_resnet', cfg.get('fine_tune') is None)
        return EMANet(cfg['num_classes'], cfg['n_layers'], use_pretrained_resnet=pretrained,
                      pretrained_path=cfg.get('pretrained_path') if pretrained else None)

    def parameter_groups(self, network):
        lr = self.config['lr']
        plan = (('1x', lr, self.config['weight_decay']), ('1y', lr, 0.0), ('2x', 2 * lr, 0.0))
        return [{'params': list(get_params(network, key=key)), 'lr': group_lr, 'weight_decay': wd}
                for key, group_lr, wd in plan]

    def optimizer_defaults(self):
        return {'momentum': self.config['lr_mom']}

    def updater_options(self):
        return {'em_mom': self.config['em_mom']}
