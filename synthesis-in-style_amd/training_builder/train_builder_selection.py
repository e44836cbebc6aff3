"""Builder lookup by ``config['network']`` (reference: training_builder/train_builder_selection.py:7-18).
DocUFCN and PixelEnsemble are outside the hot path and are not provided."""
from training_builder.ema_net_train_builder import EMANetTrainBuilder
from training_builder.trans_u_net_train_builder import TransUNetTrainBuilder


def get_train_builder_class(config):
    builders = {'TransUNet': TransUNetTrainBuilder, 'EMANet': EMANetTrainBuilder}
    if config['network'] not in builders:
        raise NotImplementedError(f"network {config['network']!r}: only {sorted(builders)} are on the MI355X hot path")
    return builders[config['network']]
