"""Golden vectors for the patch-wise page inference, produced by the REFERENCE's own methods
(segmentation/analysis_segmenter.py: calculate_bboxes_for_patches, assemble_predictions), run in this container.

    python tests/golden/make_golden_analysis.py      -> tests/golden/analysis_segmenter.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.load_reference import load_reference_analysis_segmenter  # noqa: E402

ref = load_reference_analysis_segmenter()
cases = [  # (width, height, patch, overlap or None)
    (700, 500, 256, None), (256, 256, 256, None), (513, 300, 256, None), (1000, 777, 256, 64), (300, 520, 128, 100),
]
out = {"cases": np.asarray([[w, h, p, -1 if o is None else o] for w, h, p, o in cases], dtype=np.int64)}
rng = np.random.RandomState(20240)
for i, (w, h, p, o) in enumerate(cases):
    me = types.SimpleNamespace(patch_size=p, patch_overlap=o, device="cpu", network=types.SimpleNamespace(num_classes=3),
                               progress_bar=lambda it, **kw: it)
    boxes = ref.AnalysisSegmenter.calculate_bboxes_for_patches(me, w, h)
    out[f"boxes_{i}"] = np.asarray([tuple(b) for b in boxes], dtype=np.int64)
    preds = torch.from_numpy(rng.rand(len(boxes), 3, p, p).astype(np.float32))
    patches = [{"prediction": preds[k], "bbox": boxes[k]} for k in range(len(boxes))]
    assembled = ref.AnalysisSegmenter.assemble_predictions(me, patches, (w, h))
    out[f"pred_seed_{i}"] = np.asarray([20240, i])
    # the predictions are re-derived from the RandomState stream in the test; keep checksums and a strided slice
    out[f"assembled_sum_{i}"] = np.asarray(assembled.double().sum().item())
    out[f"assembled_slice_{i}"] = assembled[:, ::37, ::41].numpy()
    out[f"labels_slice_{i}"] = torch.max(assembled, dim=0)[1][::17, ::19].numpy().astype(np.uint8)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "analysis_segmenter.npz"), **out)
print({k: v.shape for k, v in out.items()})
