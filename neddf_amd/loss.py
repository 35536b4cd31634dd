"""Objective functions -- host-side mirror of neddf/loss/*.py.

These operate on the per-ray outputs of render_rays ([B] / [B,3] tensors), a few
kilobytes per step: they stay plain torch expressions, exactly the reference's,
so that autograd hands the composite / field backward kernels the very same
upstream gradients.
"""
from abc import ABC, abstractmethod
from typing import Dict

import torch
from torch import Tensor, nn


class BaseLoss(ABC, nn.Module):
    """base_loss.py:7-88: weight * loss(outputs[key_output], targets[key_target]) for the fine pass and, when
    weight_coarse > 0, the same on '<key_output>_coarse' under '<key_loss>_coarse'."""

    def __init__(self, key_output: str, key_target: str, key_loss: str, weight: float = 1.0,
                 weight_coarse: float = 0.1) -> None:
        super().__init__()
        self.weight, self.weight_coarse = weight, weight_coarse
        self.key_output, self.key_target, self.key_loss = key_output, key_target, key_loss

    def forward(self, outputs: Dict[str, Tensor], targets: Dict[str, Tensor]) -> Dict[str, Tensor]:
        assert self.key_output in outputs
        assert self.key_target in targets
        loss_dict: Dict[str, Tensor] = {
            self.key_loss: self.weight * self.loss(outputs[self.key_output], targets[self.key_target])}
        if self.weight_coarse > 0.0:
            key_output_coarse = "{}_coarse".format(self.key_output)
            assert key_output_coarse in outputs
            loss_dict["{}_coarse".format(self.key_loss)] = self.weight_coarse * self.loss(
                outputs[key_output_coarse], targets[self.key_target])
        return loss_dict

    @abstractmethod
    def loss(self, output: Tensor, target: Tensor) -> Tensor:
        raise NotImplementedError()


class ColorLoss(BaseLoss):
    """color_loss.py: mean squared colour error."""

    def __init__(self, weight: float = 1.0, weight_coarse: float = 0.1) -> None:
        super().__init__("color", "color", "color", weight, weight_coarse)

    def loss(self, output: Tensor, target: Tensor) -> Tensor:
        return torch.mean(torch.square(output - target))


class MaskBCELoss(BaseLoss):
    """mask_bce_loss.py: binary cross entropy between 1 - transmittance (clamped to [1e-6, 1-1e-6]) and the mask."""

    def __init__(self, weight: float = 1.0, weight_coarse: float = 0.1) -> None:
        super().__init__("transmittance", "mask", "mask", weight, weight_coarse)

    def loss(self, output: Tensor, target: Tensor) -> Tensor:
        mask_output = torch.clamp(1.0 - output, 1e-6, 1.0 - 1e-6)
        return -torch.mean(target * torch.log(mask_output) + (1.0 - target) * torch.log(1.0 - mask_output))


class MaskMSELoss(BaseLoss):
    """mask_mse_loss.py: squared error between the clamped 1 - transmittance and the mask."""

    def __init__(self, weight: float = 1.0, weight_coarse: float = 0.1) -> None:
        super().__init__("transmittance", "mask", "mask", weight, weight_coarse)

    def loss(self, output: Tensor, target: Tensor) -> Tensor:
        mask_output = torch.clamp(1.0 - output, 1e-6, 1.0 - 1e-6)
        return torch.mean(torch.square(mask_output - target))


class FieldsConstraintLoss(BaseLoss):
    """fields_constraint_loss.py: mean of the integrated field penalties (the target is ignored)."""

    def __init__(self, weight: float = 1.0, weight_coarse: float = 0.1) -> None:
        super().__init__("fields_penalty", "fields_penalty", "fields_penalty", weight, weight_coarse)

    def loss(self, output: Tensor, target: Tensor) -> Tensor:
        return torch.mean(output)
