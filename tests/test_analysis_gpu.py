"""GPU parity: patch-wise page inference (csrc/page_ops.hip, segmentation/analysis_segmenter.py) against the oracle
and the reference-made golden vectors.  Crop, assemble and label maps are bit-exact."""
import os

import numpy as np
import pytest
import torch

from oracle import analysis_ref as A

pytestmark = pytest.mark.gpu


def test_crop_patches_bit_exact(device):
    import sis_hip
    from segmentation.analysis_segmenter import AnalysisSegmenter
    rng = np.random.RandomState(5)
    for (w, h, p, o, c) in [(700, 500, 256, None, 3), (300, 520, 128, 100, 1), (256, 256, 256, None, 3), (97, 33, 64, 7, 4)]:
        page = rng.randint(0, 256, size=(h, w, c), dtype=np.uint8)
        seg = AnalysisSegmenter(torch.nn.Identity(), p, device, patch_overlap=o or 0)
        xs, ys = seg.patch_grid(w, h)
        got = sis_hip.crop_patches_u8(torch.from_numpy(page).to(device), xs, ys, p).cpu()
        want = A.crop_patches(page, A.calculate_bboxes_for_patches(w, h, p, o))
        assert torch.equal(got, want), (w, h, p, o)


def test_assemble_matches_reference_golden(device, golden_dir):
    import sis_hip
    from segmentation.analysis_segmenter import AnalysisSegmenter
    g = np.load(os.path.join(golden_dir, "analysis_segmenter.npz"))
    rng = np.random.RandomState(20240)
    for i, (w, h, p, o) in enumerate(g["cases"].tolist()):
        o = None if o < 0 else o
        boxes = A.calculate_bboxes_for_patches(w, h, p, o)
        preds = torch.from_numpy(rng.rand(len(boxes), 3, p, p).astype(np.float32))
        seg = AnalysisSegmenter(torch.nn.Identity(), p, device, patch_overlap=o or 0)
        out, labels = seg.assemble_predictions(preds.to(device), (w, h), with_labels=True)
        np.testing.assert_array_equal(out[:, ::37, ::41].cpu().numpy(), g[f"assembled_slice_{i}"])
        np.testing.assert_array_equal(labels[::17, ::19].cpu().numpy(), g[f"labels_slice_{i}"])
        want = A.assemble_predictions(preds, boxes, w, h)
        assert torch.equal(out.cpu(), want)
        assert torch.equal(labels.cpu().long(), A.label_map(want))


def test_segment_image_end_to_end(device):
    """A BaseSegmenter whose forward is element-wise, so the only GPU/CPU difference is softmax rounding."""
    from PIL import Image
    from networks.base_segmenter import BaseSegmenter
    from segmentation.analysis_segmenter import AnalysisSegmenter

    class Pointwise(BaseSegmenter):
        num_classes = 3

        def forward(self, x):
            return torch.stack([2.0 * x[:, 0], x[:, 1] - 0.25 * x[:, 2], -1.5 * x[:, 2] + 0.1], dim=1)

    rng = np.random.RandomState(11)
    page = rng.randint(0, 256, size=(333, 450, 3), dtype=np.uint8)
    net = Pointwise()
    seg = AnalysisSegmenter(net.to(device), 128, device, batch_size=5, patch_overlap=32)
    got = seg.segment_image(Image.fromarray(page))
    labels = seg.segment_labels(page)
    boxes = A.calculate_bboxes_for_patches(450, 333, 128, 32)
    with torch.no_grad():
        preds = net.predict(A.crop_patches(page, boxes))
    want = A.assemble_predictions(preds, boxes, 450, 333)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    top2 = torch.topk(want, 2, dim=0)[0]
    decided = (top2[0] - top2[1]) > 1e-5
    assert decided.float().mean() > 0.99
    assert torch.equal(labels.cpu().long()[decided], A.label_map(want)[decided])
    assert labels.dtype == torch.uint8 and tuple(labels.shape) == (333, 450)
