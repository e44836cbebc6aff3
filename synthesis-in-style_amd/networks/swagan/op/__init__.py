"""Operator surface of the SWAGAN model (reference: networks/swagan/op/__init__.py:1-2).  The reference keeps a
second copy of the two CUDA extensions here; its public ``upfirdn2d`` wrapper is the same as stylegan2's
(networks/swagan/op/upfirdn2d.py:143-148: only pad[0], pad[1] are used, on both axes), so one HIP library serves both."""
from networks.stylegan2.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d  # noqa: F401
