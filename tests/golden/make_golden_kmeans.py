"""Generates tests/golden/kmeans_reference.npz from the UNMODIFIED reference ``FactorCatalog`` (build container only).

``segmentation/gan_local_edit/factor_catalog.py`` and ``ptutils.py`` are loaded by file path; the module's one other
import, ``spherical_kmeans`` (sklearn-0.24 private APIs, only used for fitting), is stood in by a holder of
``cluster_centers_``.  ``FactorCatalog.predict`` -> ``pairwise_distance`` (:47-62) then runs as written; its final
``cluster_ids.cuda()`` is made a no-op for the call (there is no GPU here).

Inputs are re-derived from frozen ``numpy.random.RandomState`` streams (``case_inputs``), only the label maps are stored.
Every second centre is a ~1e-7 relative perturbation of its predecessor, so that most pixels sit within a few ulp of
a tie between two centres: the label then depends on the association of the fp32 adds inside ``.sum(dim=-1)`` -- the
property the device kernel's documented summation order has to reproduce.

    python tests/golden/make_golden_kmeans.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/stylegan_code_finder"

CASES = [(2, 128, 32, 32, 24), (1, 512, 16, 16, 24), (1, 32, 5, 7, 40), (1, 1056, 4, 4, 6), (1, 20, 3, 3, 5)]


def case_inputs(index):
    b, c, h, w, k = CASES[index]
    rs = np.random.RandomState(1000 + index)
    x = rs.standard_normal((b, c, h, w)).astype(np.float32)
    centres = rs.standard_normal((k, c)).astype(np.float32)
    for j in range(1, k, 2):
        centres[j] = centres[j - 1] * (1 + 1e-7 * rs.standard_normal(c)).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(centres)


def load_reference_factor_catalog():
    for name in ("segmentation", "segmentation.gan_local_edit"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    stub = types.ModuleType("segmentation.gan_local_edit.spherical_kmeans")

    class MiniBatchSphericalKMeans:
        def __init__(self, n_clusters, random_state=0, **kwargs):
            self.cluster_centers_ = None

    stub.MiniBatchSphericalKMeans = MiniBatchSphericalKMeans
    sys.modules[stub.__name__] = stub
    for mod in ("ptutils", "factor_catalog"):
        name = f"segmentation.gan_local_edit.{mod}"
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, "segmentation", "gan_local_edit", f"{mod}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        setattr(sys.modules["segmentation.gan_local_edit"], mod, m)
        spec.loader.exec_module(m)
    return sys.modules["segmentation.gan_local_edit.factor_catalog"].FactorCatalog


def reference_labels(FactorCatalog, x, centres):
    cat = FactorCatalog(centres.shape[0])
    cat._factorization.cluster_centers_ = centres.numpy()
    keep = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self  # no GPU in the build container
    try:
        return cat.predict(x)
    finally:
        torch.Tensor.cuda = keep


if __name__ == "__main__":
    from oracle import kmeans_ref
    FactorCatalog = load_reference_factor_catalog()
    out = {}
    for i in range(len(CASES)):
        x, centres = case_inputs(i)
        labels = reference_labels(FactorCatalog, x, centres)
        mine, _ = kmeans_ref.predict(x, centres)
        ordered, _ = kmeans_ref.predict_ordered(x, centres)
        plain64, _ = kmeans_ref.predict(x.double(), centres.double())
        assert torch.equal(labels, mine) and torch.equal(labels, ordered), f"case {i}: oracle differs from the reference"
        out[f"labels{i}"] = labels.numpy().astype(np.uint8)
        print(f"case {i} {CASES[i]}: {labels.numel()} pixels, {(labels != plain64).float().mean().item():.3f} "
              f"of them decided by fp32 rounding (differ from an fp64 evaluation)")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "kmeans_reference.npz"), **out)
