"""SGD parameter grouping of EMANet (reference: networks/ema_net/utils.py:7-21):
'1x' = convolution weights, '1y' = batch-norm scales, '2x' = every convolution / batch-norm bias."""
from torch import nn
from torch.nn.modules.batchnorm import _BatchNorm


def get_params(model, key):
    for module in model.modules():
        if key == '1x' and isinstance(module, nn.Conv2d):
            yield module.weight
        elif key == '1y' and isinstance(module, _BatchNorm) and module.weight is not None:
            yield module.weight
        elif key == '2x' and isinstance(module, (nn.Conv2d, _BatchNorm)) and module.bias is not None:
            yield module.bias
