"""ORACLE (test infrastructure, not product): import the UNMODIFIED reference
modules by file path, in THIS container only.

``/root/reference`` does not exist on the GPU box, so nothing that runs there
(``-m gpu`` tests, ``smoke()``, ``bench.py``) may call into this file; it is used
by ``tests/golden/make_golden.py`` and by the CPU tests that pin the oracle, and
those tests skip when the reference tree is absent.

Recipe: ``networks/stylegan2/model.py`` has one relative import,
``from .op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d`` (model.py:12).
We register a synthetic package ``refpkg`` whose ``refpkg.op`` is
``oracle.ops_ref`` (the reference's own ops are CUDA-only), then exec model.py as
``refpkg.model``.  Everything else -- ModulatedConv2d, Blur pads, ToRGB, forward
control flow, truncation, noise handling, parameter init -- is the reference's
own code.
"""
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = "/root/reference/stylegan_code_finder"


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "networks", "stylegan2", "model.py"))


def load_reference_stylegan2():
    """Returns the reference ``networks.stylegan2.model`` module object."""
    if "refpkg.model" in sys.modules:
        return sys.modules["refpkg.model"]
    from oracle import ops_ref

    pkg = types.ModuleType("refpkg")
    pkg.__path__ = []
    sys.modules["refpkg"] = pkg
    sys.modules["refpkg.op"] = ops_ref
    path = os.path.join(REFERENCE_ROOT, "networks", "stylegan2", "model.py")
    spec = importlib.util.spec_from_file_location("refpkg.model", path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refpkg.model"] = mod
    spec.loader.exec_module(mod)
    return mod


def _stub_modules():
    """cv2 (only used by BaseSegmenter post-processing) and ml_collections (a config
    attribute-dict) are absent from this image; neither takes part in forward/backward."""
    if "cv2" not in sys.modules:
        try:
            import cv2  # noqa: F401
        except Exception:
            sys.modules["cv2"] = types.ModuleType("cv2")
    if "ml_collections" not in sys.modules:
        try:
            import ml_collections  # noqa: F401
        except Exception:
            m = types.ModuleType("ml_collections")

            class ConfigDict(dict):
                def __getattr__(self, k):
                    try:
                        return self[k]
                    except KeyError as e:
                        raise AttributeError(k) from e

                def __setattr__(self, k, v):
                    self[k] = v

            m.ConfigDict = ConfigDict
            sys.modules["ml_collections"] = m


def load_reference_segmenters():
    """Returns (ema_net.network, trans_u_net.vit_seg_modeling, trans_u_net.utils, ema_net.utils)
    of the reference, imported with a bare ``networks`` package (skipping networks/__init__.py,
    which needs skimage / pytorch_training)."""
    _stub_modules()
    if "networks" not in sys.modules or not getattr(sys.modules["networks"], "_sis_ref_stub", False):
        if "networks" in sys.modules:
            raise RuntimeError("a different top-level 'networks' package is already imported; "
                               "import the reference segmenters in a fresh process")
        for name in ("networks", "utils"):  # bare packages: skip their __init__.py (skimage, pytorch_training)
            if name in sys.modules and not getattr(sys.modules[name], "_sis_ref_stub", False):
                raise RuntimeError(f"a different top-level '{name}' package is already imported; "
                                   "import the reference segmenters in a fresh process")
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REFERENCE_ROOT, name)]
            pkg._sis_ref_stub = True
            sys.modules[name] = pkg
    import importlib

    ema = importlib.import_module("networks.ema_net.network")
    ema_utils = importlib.import_module("networks.ema_net.utils")
    vit = importlib.import_module("networks.trans_u_net.vit_seg_modeling")
    tu_utils = importlib.import_module("networks.trans_u_net.utils")
    return ema, vit, tu_utils, ema_utils


def load_reference_analysis_segmenter():
    """The reference's ``segmentation/analysis_segmenter.py`` module, importable here with: the bare ``networks`` /
    ``utils`` packages above, stand-in modules for what the file only *names* at import time and that need absent
    third-party packages -- ``torchvision.transforms`` (used in ``crop_and_batch_patches``: ToTensor + Normalize,
    restated in oracle/analysis_ref.py), ``training_builder.train_builder_selection`` (needs ``pytorch_training``) --
    and the reference's own ``visualization`` package as a bare package.  The patch-grid and max-assemble methods
    (`calculate_bboxes_for_patches`, `assemble_predictions`) are then the reference's own code."""
    load_reference_segmenters()
    import importlib
    if "torchvision" not in sys.modules:
        try:
            import torchvision  # noqa: F401
        except Exception:
            tv = types.ModuleType("torchvision")
            tv.transforms = types.ModuleType("torchvision.transforms")
            sys.modules["torchvision"] = tv
            sys.modules["torchvision.transforms"] = tv.transforms
    for name in ("segmentation", "visualization"):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [os.path.join(REFERENCE_ROOT, name)]
            pkg._sis_ref_stub = True
            sys.modules[name] = pkg
    if "training_builder" not in sys.modules:
        tb = types.ModuleType("training_builder")
        tb.__path__ = []
        sel = types.ModuleType("training_builder.train_builder_selection")
        sel.get_train_builder_class = lambda config: (_ for _ in ()).throw(RuntimeError("not available in the oracle"))
        sys.modules["training_builder"] = tb
        sys.modules["training_builder.train_builder_selection"] = sel
    return importlib.import_module("segmentation.analysis_segmenter")


def load_reference_swagan():
    """The reference ``networks/swagan/model.py`` module, loaded by file path into a synthetic package tree whose
    ``.op`` sub-packages are the oracle's CPU ops and whose ``..stylegan2.model`` is the reference's own stylegan2
    model (the only two relative imports of that file, model.py:11-12).  ``conv2d_gradfix`` is only used by the
    discriminator's ConvLayer path and is stood in by ``torch.nn.functional``."""
    if "refsw.swagan.model" in sys.modules:
        return sys.modules["refsw.swagan.model"]
    from oracle import ops_ref

    def package(name):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
        return pkg

    def load(name, *parts):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, *parts))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    for name in ("refsw", "refsw.stylegan2", "refsw.swagan"):
        package(name)
    sys.modules["refsw.stylegan2.op"] = ops_ref
    load("refsw.stylegan2.model", "networks", "stylegan2", "model.py")
    sw_op = types.ModuleType("refsw.swagan.op")
    sw_op.FusedLeakyReLU, sw_op.fused_leaky_relu = ops_ref.FusedLeakyReLU, ops_ref.fused_leaky_relu
    sw_op.upfirdn2d = ops_ref.upfirdn2d
    import torch.nn.functional as F
    sw_op.conv2d_gradfix = F
    sys.modules["refsw.swagan.op"] = sw_op
    return load("refsw.swagan.model", "networks", "swagan", "model.py")


def load_reference_upfirdn2d_native():
    """The reference's own pure-PyTorch statement of K2, ``upfirdn2d_native`` (networks/stylegan2/op/upfirdn2d.py:152-186),
    lifted out of its file with ``ast`` -- the module itself cannot be imported (it JIT-builds the CUDA extension at
    import, upfirdn2d.py:10-17) and the function is dead code there that uses ``F`` without importing it (SURVEY.md
    appendix A) -- and exec'd with ``torch`` / ``torch.nn.functional as F`` in scope.  The function body is the
    reference's text, unmodified."""
    import ast
    import torch
    import torch.nn.functional as F
    path = os.path.join(REFERENCE_ROOT, "networks", "stylegan2", "op", "upfirdn2d.py")
    with open(path) as f:
        source = f.read()
    node = next(n for n in ast.parse(source).body if isinstance(n, ast.FunctionDef) and n.name == "upfirdn2d_native")
    scope = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=[node], type_ignores=[]), path, "exec"), scope)
    return scope["upfirdn2d_native"]
