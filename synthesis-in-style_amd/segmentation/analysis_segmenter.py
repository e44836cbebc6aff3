"""Patch-wise segmentation of a whole page, device-resident (reference: segmentation/analysis_segmenter.py:19-188).

Same class name, patch-grid rules, hyper-parameters and method names as the reference's ``AnalysisSegmenter``; what
changes is where the work happens.  The reference crops PIL patches on the host, runs them batch by batch and merges
the predictions with one ``torch.maximum`` slice assignment per patch.  Here the page is uploaded once as uint8,
``sis_crop_patches_u8`` produces every normalised patch in one pass, the network predicts ``batch_size`` patches at a
time, and ``sis_assemble_max`` gathers the per-pixel maximum (and, for ``segment_labels``, the label map) in one pass.

The constructor takes the network itself (the reference builds it from a checkpoint's config through its train
builder -- `load_network`, :73-81 -- which is checkpoint / config-file IO outside the hot path).
"""
import math
from typing import Iterator, List, Sequence, Tuple

import numpy as np
import torch

import sis_hip

BBox = Tuple[int, int, int, int]  # left, top, right, bottom (utils/segmentation_utils.py:23-27)


class AnalysisSegmenter:
    def __init__(self, network, patch_size: int, device, batch_size: int = 1, max_image_size: int = 0,
                 patch_overlap: int = 0, patch_overlap_factor: float = 0.0):
        self.network = network.eval()
        self.device = torch.device(device)
        self.batch_size = batch_size
        self.patch_size = int(patch_size)
        self.max_image_size = max_image_size
        self.set_patch_overlap(patch_overlap, patch_overlap_factor)

    def set_patch_overlap(self, patch_overlap: int, patch_overlap_factor: float) -> None:
        assert patch_overlap == 0 or patch_overlap_factor == 0.0, \
            "Only one of 'patch_overlap' and 'patch_overlap_factor' should be specified "
        if patch_overlap != 0:
            assert 0 < patch_overlap < self.patch_size, \
                f"The value of 'patch_overlap' should be in the following range: 0 < patch_overlap < patch_size " \
                f"({self.patch_size} px) "
            self.patch_overlap = patch_overlap
        elif patch_overlap_factor != 0.0:
            assert 0.0 < patch_overlap_factor < 1.0, \
                "The value of 'patch_overlap_factor' should be in the following range: 0.0 < patch_overlap_factor < 1.0 "
            self.patch_overlap = math.ceil(patch_overlap_factor * self.patch_size)
        else:
            self.patch_overlap = None

    def set_hyperparams(self, hyperparam_config: dict) -> None:
        if "patch_overlap" in hyperparam_config:
            self.set_patch_overlap(*hyperparam_config["patch_overlap"])
        if "min_confidence" in hyperparam_config:
            self.network.min_confidence = hyperparam_config["min_confidence"]
        if "min_contour_area" in hyperparam_config:
            self.network.min_contour_area = hyperparam_config["min_contour_area"]

    # ---- patch grid ------------------------------------------------------------------------------------------
    def patch_grid(self, image_width: int, image_height: int) -> Tuple[List[int], List[int]]:
        """Left edges and top edges of the patch columns / rows; the reference's list of boxes is their row-major
        product (both of its enumeration rules produce a grid)."""
        size = self.patch_size
        if self.patch_overlap is not None:
            step = size - self.patch_overlap
            return list(range(0, image_width, step)), list(range(0, image_height, step))
        nx, ny = math.ceil(image_width / size), math.ceil(image_height / size)
        step_x = size - (nx * size - image_width) // nx
        step_y = size - (ny * size - image_height) // ny
        return [xi * step_x for xi in range(nx)], [yi * step_y for yi in range(ny)]

    def calculate_bboxes_for_patches(self, image_width: int, image_height: int) -> Tuple[BBox, ...]:
        xs, ys = self.patch_grid(image_width, image_height)
        size = self.patch_size
        return tuple((x, y, x + size, y + size) for y in ys for x in xs)

    # ---- inference ---------------------------------------------------------------------------------------------
    def _page_tensor(self, image) -> torch.Tensor:
        """uint8 [H,W,C] on the device from a PIL image (converted to the network's colour space and thumbnailed
        like the reference, :169-180), a numpy array or a tensor."""
        if hasattr(image, "convert"):  # PIL
            channels = getattr(self.network, "num_input_channels", 3)
            if channels not in (1, 3):
                raise ValueError("Can not convert input image to desired format, Network desires inputs with "
                                 f"{channels} channels.")
            image = image.convert("RGB" if channels == 3 else "L")
            if self.max_image_size and self.max_image_size > 0 and any(s > self.max_image_size for s in image.size):
                image.thumbnail((self.max_image_size, self.max_image_size))
            image = np.array(image)  # a writable copy: torch does not wrap read-only arrays silently
        page = torch.as_tensor(image)
        if page.dim() == 2:
            page = page.unsqueeze(-1)
        if page.dtype != torch.uint8 or page.dim() != 3:
            raise ValueError("expected a PIL image or a uint8 [H, W, C] array")
        return page.to(self.device).contiguous()

    def crop_and_batch_patches(self, page: torch.Tensor) -> Iterator[dict]:
        height, width, _ = page.shape
        xs, ys = self.patch_grid(width, height)
        patches = sis_hip.crop_patches_u8(page, xs, ys, self.patch_size)
        boxes = self.calculate_bboxes_for_patches(width, height)
        for i in range(0, len(boxes), self.batch_size):
            yield {'images': patches[i:i + self.batch_size], 'bboxes': boxes[i:i + self.batch_size]}

    def predict_patches(self, patches: Iterator[dict]) -> torch.Tensor:
        with torch.no_grad():
            return torch.cat([self.network.predict(batch['images']) for batch in patches], dim=0)

    def assemble_predictions(self, predictions: torch.Tensor, output_size: Sequence[int], with_labels: bool = False):
        """``output_size`` = (width, height) as in the reference (:147-167)."""
        width, height = output_size
        xs, ys = self.patch_grid(width, height)
        return sis_hip.assemble_max(predictions, xs, ys, height, width, with_labels=with_labels)

    def segment_image(self, image) -> torch.Tensor:
        """[classes, H, W] per-pixel maximum of the overlapping patch predictions."""
        page = self._page_tensor(image)
        predictions = self.predict_patches(self.crop_and_batch_patches(page))
        return self.assemble_predictions(predictions, (page.shape[1], page.shape[0]))

    def segment_labels(self, image) -> torch.Tensor:
        """uint8 [H, W] label map (first maximal class, networks/base_segmenter.py:59-62) of ``segment_image``."""
        page = self._page_tensor(image)
        predictions = self.predict_patches(self.crop_and_batch_patches(page))
        return self.assemble_predictions(predictions, (page.shape[1], page.shape[0]), with_labels=True)[1]
