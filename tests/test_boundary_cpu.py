"""CPU suite, part 2: the drop-in boundary (no compute calls -- there is no GPU here).

* libsis_hip.so loads and exports every symbol include/sis_hip.h declares;
* the Python operator surface has the reference's names and rejects CPU tensors with RuntimeError
  (reference: fused_bias_act.cpp:13-14, upfirdn2d.cpp:15-16), i.e. there is no CPU fallback;
* Generator has the reference's 135-key state_dict schema and loads a g_ema checkpoint strictly;
* the product sources never import the oracle.
"""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sis_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sis_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import sis_hip
    assert os.path.exists(sis_hip.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(sis_hip.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 14
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sis_hip.h but not exported"
    assert sorted(sis_hip.exported_symbols()) == declared
    assert sis_hip.lib().sis_version() >= 1000


def test_out_size_helper_matches_reference_formula():
    import sis_hip
    L = sis_hip.lib()
    for (n, up, down, p0, p1, k) in [(9, 1, 1, 1, 1, 4), (4, 2, 1, 2, 1, 4), (8, 1, 2, 1, 1, 4), (5, 3, 2, 2, 3, 5)]:
        assert L.sis_upfirdn2d_out_size(n, up, down, p0, p1, k) == (n * up + p0 + p1 - k) // down + 1


def test_ops_reject_cpu_tensors():
    from networks.stylegan2.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        fused_leaky_relu(torch.randn(2, 3, 4, 4), torch.zeros(3))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        upfirdn2d(torch.randn(1, 2, 8, 8), torch.ones(4, 4) / 16, pad=(1, 1))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        FusedLeakyReLU(3)(torch.randn(2, 3))


def test_generator_schema_and_strict_load():
    from networks.stylegan2.model import Generator
    from oracle import stylegan2_ref as R
    for size, sdim, n_mlp, cm in [(16, 64, 3, 1), (256, 512, 8, 2)]:
        with torch.device("meta"):
            g = Generator(size, sdim, n_mlp, channel_multiplier=cm)
        schema = R.state_dict_schema(size, sdim, n_mlp, cm)
        sd = g.state_dict()
        assert list(sd.keys()) == [n for n, _ in schema]
        for n, shape in schema:
            assert tuple(sd[n].shape) == tuple(shape), n
    g = Generator(16, 64, 3, channel_multiplier=1)
    g.load_state_dict(R.seeded_state_dict(16, 64, 3, 1, seed=5), strict=True)
    assert g.n_latent == 6 and g.num_layers == 5 and g.log_size == 4 and g.size == 16
    assert [tuple(n.shape) for n in g.make_noise()] == [(1, 1, 4, 4), (1, 1, 8, 8), (1, 1, 8, 8), (1, 1, 16, 16),
                                                         (1, 1, 16, 16)]


def test_generator_on_cpu_raises_instead_of_falling_back():
    from networks.stylegan2.model import Generator
    g = Generator(16, 64, 2, channel_multiplier=1).eval()
    with torch.no_grad(), pytest.raises(RuntimeError):
        g([torch.randn(2, 64)])


def test_product_never_imports_oracle():
    src = os.path.join(ROOT, "synthesis-in-style_amd")
    for dirpath, _, files in os.walk(src):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)


def test_library_call_audit(monkeypatch):
    """sis_hip.library_call: the per-site counters bench.py reports (library_calls_per_step) and the strict position
    SIS_NO_LIBRARY_FALLBACK=1, in which a layer an own kernel should have taken raises instead of running on MIOpen / hipBLASLt;
    layers that are MEANT to stay on the libraries (the 3-channel stem) never raise."""
    import sis_hip
    sis_hip.library_calls(reset=True)
    sis_hip.library_call("unit.site")
    sis_hip.library_call("unit.site")
    sis_hip.library_call("unit.stem", intended=True)
    assert sis_hip.library_calls(reset=True) == {"fallback": {"unit.site": 2}, "intended": {"unit.stem": 1}}
    assert sis_hip.library_calls() == {"fallback": {}, "intended": {}}
    monkeypatch.setattr(sis_hip, "_LIBRARY_STRICT", True)
    sis_hip.library_call("unit.stem", intended=True)
    with pytest.raises(RuntimeError, match="unit.site fell back"):
        sis_hip.library_call("unit.site")
    sis_hip.library_calls(reset=True)


def test_every_kernel_of_the_sources_counts_as_own():
    """bench.py's library-time audit and tools/pmc_traffic.py tell own kernels from library ones by the ``__global__`` function names
    found in csrc/*.hip (``sis_hip.own_kernel_names``): the scan must see kernels declared with launch bounds, attributes
    (``amdgpu_waves_per_eu``) or ``static`` in any order -- a missed name would book that kernel's time as library time."""
    import re
    import sis_hip
    names = sis_hip.own_kernel_names()
    for expect in ("modconv_upfir_kernel", "modconv_wino2_kernel", "kmeans_mfma_kernel", "gemm_bf16_kernel", "gemm_colsum_finish_multi_kernel",
                   "wino_prepack_multi_kernel", "conv_wgrad_bf16_kernel", "conv1x1_wgrad_f32_kernel", "bn_wide_bwd_kernel"):
        assert expect in names, expect
    assert sis_hip.is_own_kernel("void (anonymous namespace)::modconv_upfir_kernel<1, 0, 4>((anonymous namespace)::UpFirParams)")
    assert not sis_hip.is_own_kernel("Cijk_Alik_Bljk_BBS_BH_Bias_HA_S_SAV_UserArgs_MT256x256x64")
    # every `__global__` in the sources is followed (eventually) by a function name the scan found
    import glob
    import os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "synthesis-in-style_amd", "csrc")
    for path in glob.glob(os.path.join(csrc, "*.hip")):
        text = open(path).read()
        for m in re.finditer(r"__global__[^;{]*?void\s+(?:__launch_bounds__\s*\([^)]*\)\s*)?([A-Za-z_]\w*)\s*\(", text):
            assert m.group(1) in names, (os.path.basename(path), m.group(1))
