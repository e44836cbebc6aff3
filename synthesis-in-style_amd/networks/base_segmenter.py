"""Common base of the segmentation networks (reference: networks/base_segmenter.py:11-65).

On the hot path only two things matter: it is the nn.Module base class of EMANet / TransUNet, and
``predict_classes`` yields the argmax label map.  The reference's inference-time clean-up (confidence
threshold, then OpenCV contour removal below ``min_contour_area``) is CPU post-processing outside the
training step; the threshold is kept (pure tensor op), contour removal is not reimplemented and raises if
requested.
"""
from typing import Any

import torch
import torch.nn.functional as F
from torch import nn


class BaseSegmenter(nn.Module):
    def __init__(self, background_class_id: int = 0, min_confidence: float = 0.0, min_contour_area: int = 0,
                 num_input_channels: int = 3):
        super().__init__()
        self.background_class_id = background_class_id
        self.min_confidence = min_confidence
        self.min_contour_area = min_contour_area
        self.num_input_channels = num_input_channels

    def postprocess(self, predictions: torch.Tensor) -> torch.Tensor:
        if self.min_contour_area > 0:
            raise NotImplementedError("contour-area filtering is OpenCV post-processing (out of the hot path)")
        return torch.where(predictions < self.min_confidence, torch.zeros_like(predictions), predictions)

    def predict(self, x: torch.Tensor) -> torch.Tensor:
        return self.postprocess(F.softmax(self.forward(x), dim=1))

    def predict_classes(self, x: torch.Tensor) -> torch.Tensor:
        return torch.argmax(self.predict(x), dim=1, keepdim=True)

    def forward(self, x: Any):
        raise NotImplementedError
