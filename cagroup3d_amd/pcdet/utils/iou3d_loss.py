"""IoU3DLoss (pcdet/utils/iou3d_loss.py:14-95): 1 - IoU, rotated (sort_vertices path) or axis-aligned."""
import torch
import torch.nn as nn

from ...ops.rotated_iou import cal_iou_3d, rotated_iou3d  # noqa: F401
from .loss_utils import AxisAlignedBboxOverlaps3D, weight_reduce_loss


def iou_3d_loss(pred, target, weight=None, reduction="mean", avg_factor=None):
    loss = 1 - rotated_iou3d(pred, target).unsqueeze(0)        # (1, n), the shape the reference's batched form returns
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


def _corners(b):
    half = b[..., 3:6] / 2
    return torch.cat((b[..., 0:3] - half, b[..., 0:3] + half), dim=-1)


def axis_aligned_iou_loss(pred, target, weight=None, reduction="mean", avg_factor=None):
    loss = 1 - AxisAlignedBboxOverlaps3D()(_corners(pred), _corners(target), is_aligned=True)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class IoU3DLoss(nn.Module):
    def __init__(self, with_yaw=True, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.loss_function = iou_3d_loss if with_yaw else axis_aligned_iou_loss
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if weight is not None and not torch.any(weight > 0):
            return pred.sum() * weight.sum()
        assert reduction_override in (None, "none", "mean", "sum")
        if weight is not None and weight.dim() > 1:
            weight = weight.mean(-1)
        return self.loss_weight * self.loss_function(pred, target, weight,
                                                     reduction=reduction_override or self.reduction,
                                                     avg_factor=avg_factor, **kwargs)
