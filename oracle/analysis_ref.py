"""ORACLE (test infrastructure, not product): patch-wise inference of a page image, restated on the CPU.

Follows /root/reference/stylegan_code_finder/segmentation/analysis_segmenter.py:
  calculate_bboxes_for_patches :83-113   fixed overlap, or the automatic overlap that spreads the surplus of
                                         ceil(size / patch) windows evenly (integer division)
  crop_and_batch_patches       :115-130  PIL crop (zero padding outside the image) -> ToTensor (u8 / 255) ->
                                         Normalize(0.5, 0.5) = (t - 0.5) / 0.5
  predict_patches              :132-145  network.predict per batch (networks/base_segmenter.py:54-57)
  assemble_predictions         :147-167  element-wise maximum of the overlapping patch predictions, -inf start
and networks/base_segmenter.py:59-62 (predict_classes = index of the first maximum over classes).
Pinned by tests/golden/analysis_segmenter.npz, produced by the reference's own methods (make_golden_analysis.py).
"""
import math

import numpy as np
import torch


def calculate_bboxes_for_patches(image_width, image_height, patch_size, patch_overlap=None):
    """List of (left, top, right, bottom), row-major, exactly as the reference enumerates them."""
    boxes = []
    if patch_overlap is not None:
        y = 0
        while y < image_height:
            x = 0
            while x < image_width:
                boxes.append((x, y, x + patch_size, y + patch_size))
                x += patch_size - patch_overlap
            y += patch_size - patch_overlap
        return boxes
    nx = math.ceil(image_width / patch_size)
    ny = math.ceil(image_height / patch_size)
    ox = (nx * patch_size - image_width) // nx
    oy = (ny * patch_size - image_height) // ny
    for yi in range(ny):
        top = int(yi * (patch_size - oy))
        for xi in range(nx):
            left = int(xi * (patch_size - ox))
            boxes.append((left, top, left + patch_size, top + patch_size))
    return boxes


def crop_patches(image_u8_hwc, boxes):
    """uint8 [H,W,C] -> float32 [N,C,P,P] in [-1,1]; pixels outside the image are PIL's zero padding (-> -1)."""
    img = torch.as_tensor(np.ascontiguousarray(image_u8_hwc))
    h, w, c = img.shape
    out = []
    for left, top, right, bottom in boxes:
        patch = torch.zeros((bottom - top, right - left, c), dtype=torch.uint8)
        y1, x1 = min(bottom, h), min(right, w)
        patch[:y1 - top, :x1 - left] = img[top:y1, left:x1]
        t = patch.permute(2, 0, 1).to(torch.float32).div(255)  # ToTensor
        out.append((t - 0.5) / 0.5)                             # Normalize((.5,.5,.5), (.5,.5,.5))
    return torch.stack(out, 0)


def assemble_predictions(predictions, boxes, width, height):
    """predictions [N,C,P,P] -> [C,H,W] element-wise maximum over the patches covering each pixel."""
    n_classes = predictions.shape[1]
    out = torch.full((height, width, n_classes), float("-inf"))
    for pred, (left, top, right, bottom) in zip(predictions, boxes):
        right, bottom = min(right, width), min(bottom, height)
        window = out[top:bottom, left:right, :]
        out[top:bottom, left:right, :] = torch.maximum(window, pred.permute(1, 2, 0)[:bottom - top, :right - left, :])
    return out.permute(2, 0, 1)


def label_map(assembled):
    """Index of the first maximal class per pixel (torch.max(dim)[1], base_segmenter.py:61)."""
    return torch.max(assembled, dim=0)[1]
