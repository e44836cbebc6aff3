"""Same export list as the reference package (networks/stylegan2/op/__init__.py:1-2)."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d
