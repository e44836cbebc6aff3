"""Dice loss of the TransUNet objective (reference: networks/trans_u_net/utils.py:7-42).

loss = mean_c [ 1 - (2 * sum(p_c * t_c) + 1e-5) / (sum(p_c^2) + sum(t_c^2) + 1e-5) ], sums over the whole batch,
p = softmax(logits) when ``softmax=True``, t = one-hot labels.  Written over whole tensors (one reduction per
term for all classes) instead of a Python loop over classes.
"""
import torch
import torch.nn as nn


class DiceLoss(nn.Module):
    def __init__(self, n_classes):
        super().__init__()
        self.n_classes = n_classes

    def _one_hot_encoder(self, input_tensor):
        classes = torch.arange(self.n_classes, device=input_tensor.device).view(1, -1, *([1] * (input_tensor.dim() - 1)))
        return (input_tensor.unsqueeze(1) == classes).float()

    def forward(self, inputs, target, weight=None, softmax=False):
        if softmax:
            inputs = torch.softmax(inputs, dim=1)
        target = self._one_hot_encoder(target)
        assert inputs.size() == target.size(), f'predict {inputs.size()} & target {target.size()} shape do not match'
        dims = [d for d in range(inputs.dim()) if d != 1]
        smooth = 1e-5
        intersect = (inputs * target).sum(dims)
        denom = (inputs * inputs).sum(dims) + (target * target).sum(dims)
        dice = 1 - (2 * intersect + smooth) / (denom + smooth)
        if weight is not None:
            dice = dice * torch.as_tensor(weight, dtype=dice.dtype, device=dice.device)
        return dice.sum() / self.n_classes
