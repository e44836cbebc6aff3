"""Model configurations of TransUNet (reference: networks/trans_u_net/vit_seg_configs.py:6-62).

The reference stores them in ``ml_collections.ConfigDict`` (a third-party attribute dict, pinned 0.1.0 in
requirements.txt); only attribute / item access, ``.get`` and assignment are used on the hot path, which the
small ``ConfigDict`` below provides without the dependency.
"""


class ConfigDict(dict):
    """dict with attribute access; nested dicts are wrapped on construction."""

    def __init__(self, initial=None, **kwargs):
        super().__init__()
        for k, v in dict(initial or {}, **kwargs).items():
            self[k] = v

    def __setitem__(self, key, value):
        super().__setitem__(key, ConfigDict(value) if isinstance(value, dict) and not isinstance(value, ConfigDict)
                            else value)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    __setattr__ = __setitem__

    def copy(self):
        return ConfigDict({k: (v.copy() if isinstance(v, ConfigDict) else (list(v) if isinstance(v, list) else v))
                           for k, v in self.items()})


def _vit(hidden, mlp, heads, layers, patch):
    return ConfigDict(
        patches={'size': (patch, patch)}, hidden_size=hidden,
        transformer={'mlp_dim': mlp, 'num_heads': heads, 'num_layers': layers, 'attention_dropout_rate': 0.0,
                     'dropout_rate': 0.1},
        classifier='seg', representation_size=None, resnet_pretrained_path=None, patch_size=patch,
        decoder_channels=(256, 128, 64, 16), n_classes=2, activation='softmax')


def get_b16_config():
    return _vit(768, 3072, 12, 12, 16)


def get_b32_config():
    return _vit(768, 3072, 12, 12, 32)


def get_l16_config():
    return _vit(1024, 4096, 16, 24, 16)


def get_l32_config():
    return _vit(1024, 4096, 16, 24, 32)


def get_h14_config():
    c = _vit(1280, 5120, 16, 32, 14)
    c.classifier = 'token'
    return c


def _hybrid(config, resnet_layers):
    """ResNetV2 stem in front of the transformer: 1x1 'patches' on a (S/16)^2 feature grid, three skips."""
    config.patches.grid = (16, 16)
    config.resnet = ConfigDict(num_layers=resnet_layers, width_factor=1)
    config.classifier = 'seg'
    config.decoder_channels = (256, 128, 64, 16)
    config.skip_channels = [512, 256, 64, 16]
    config.n_classes = 2
    config.n_skip = 3
    config.activation = 'softmax'
    return config


def get_r50_b16_config():
    return _hybrid(get_b16_config(), (3, 4, 9))


def get_r50_l16_config():
    return _hybrid(get_l16_config(), (3, 4, 9))


def get_testing():
    return ConfigDict(patches={'size': (16, 16)}, hidden_size=1,
                      transformer={'mlp_dim': 1, 'num_heads': 1, 'num_layers': 1, 'attention_dropout_rate': 0.0,
                                   'dropout_rate': 0.1}, classifier='token', representation_size=None)
