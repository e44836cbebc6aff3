"""CPU: the page-inference oracle (oracle/analysis_ref.py) against the golden vectors produced by the reference's own
AnalysisSegmenter methods, the PIL crop it restates, and the product's host-side patch grid."""
import os

import numpy as np
import pytest
import torch

from oracle import analysis_ref as A


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "analysis_segmenter.npz"))
    rng = np.random.RandomState(20240)
    for i, (w, h, p, o) in enumerate(g["cases"].tolist()):
        n = len(g[f"boxes_{i}"])
        preds = torch.from_numpy(rng.rand(n, 3, p, p).astype(np.float32))
        yield g, i, w, h, p, (None if o < 0 else o), preds


def test_oracle_matches_reference_golden(golden_dir):
    for g, i, w, h, p, o, preds in _cases(golden_dir):
        boxes = A.calculate_bboxes_for_patches(w, h, p, o)
        np.testing.assert_array_equal(np.asarray(boxes), g[f"boxes_{i}"])
        assembled = A.assemble_predictions(preds, boxes, w, h)
        assert assembled.shape == (3, h, w)
        np.testing.assert_array_equal(assembled[:, ::37, ::41].numpy(), g[f"assembled_slice_{i}"])
        np.testing.assert_allclose(assembled.double().sum().item(), g[f"assembled_sum_{i}"], rtol=1e-12)
        np.testing.assert_array_equal(A.label_map(assembled)[::17, ::19].numpy().astype(np.uint8), g[f"labels_slice_{i}"])


def test_oracle_crop_is_pil_crop_plus_totensor_normalize():
    from PIL import Image
    rng = np.random.RandomState(3)
    page = rng.randint(0, 256, size=(150, 210, 3), dtype=np.uint8)
    boxes = A.calculate_bboxes_for_patches(210, 150, 128, None)
    got = A.crop_patches(page, boxes)
    img = Image.fromarray(page)
    for k, box in enumerate(boxes):
        patch = np.asarray(img.crop(box))  # PIL pads with zeros outside the image
        want = (torch.from_numpy(patch.copy()).permute(2, 0, 1).float().div(255) - 0.5) / 0.5
        assert torch.equal(got[k], want)
    assert got.min() == -1.0  # the padding


def test_product_patch_grid_is_the_reference_enumeration(golden_dir):
    from segmentation.analysis_segmenter import AnalysisSegmenter
    for g, i, w, h, p, o, _ in _cases(golden_dir):
        seg = AnalysisSegmenter(torch.nn.Identity(), p, "cpu", patch_overlap=o or 0)
        np.testing.assert_array_equal(np.asarray(seg.calculate_bboxes_for_patches(w, h)), g[f"boxes_{i}"])
    seg = AnalysisSegmenter(torch.nn.Identity(), 256, "cpu", patch_overlap_factor=0.3)
    assert seg.patch_overlap == 77
    with pytest.raises(AssertionError):
        AnalysisSegmenter(torch.nn.Identity(), 256, "cpu", patch_overlap=10, patch_overlap_factor=0.5)
