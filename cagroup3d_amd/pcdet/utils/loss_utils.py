"""The losses of pcdet/utils/loss_utils.py that CAGroup3D uses, restated:
WeightedSmoothL1Loss (:76-137), axis_aligned_bbox_overlaps_3d / AxisAlignedBboxOverlaps3D (:370-538),
binary_cross_entropy / CrossEntropy (:780-887), py_sigmoid_focal_loss / FocalLoss (:917-1040),
smooth_l1_loss / SmoothL1Loss (:1042-1123)."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def reduce_loss(loss, reduction):
    if reduction == "none":
        return loss
    return loss.mean() if reduction == "mean" else loss.sum()


def weight_reduce_loss(loss, weight=None, reduction="mean", avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == "mean":
        return loss.sum() / avg_factor
    if reduction != "none":
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


class WeightedSmoothL1Loss(nn.Module):
    """Code-wise weighted smooth-L1, no reduction (loss_utils.py:76-137).  Unlike the reference the
    code weights are a buffer that follows the module's device (the reference calls .cuda() in
    __init__, loss_utils.py:98)."""

    def __init__(self, beta=1.0 / 9.0, code_weights=None):
        super().__init__()
        self.beta = beta
        cw = None if code_weights is None else torch.tensor(np.asarray(code_weights, dtype=np.float32))
        self.register_buffer("code_weights", cw, persistent=False)

    @staticmethod
    def smooth_l1_loss(diff, beta):
        n = torch.abs(diff)
        if beta < 1e-5:
            return n
        return torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)

    def forward(self, input, target, weights=None):
        target = torch.where(torch.isnan(target), input, target)
        diff = input - target
        if self.code_weights is not None:
            diff = diff * self.code_weights.to(diff.device).view(1, 1, -1)
        loss = self.smooth_l1_loss(diff, self.beta)
        if weights is not None:
            assert weights.shape[0] == loss.shape[0] and weights.shape[1] == loss.shape[1]
            loss = loss * weights.unsqueeze(-1)
        return loss


def axis_aligned_bbox_overlaps_3d(bboxes1, bboxes2, mode="iou", is_aligned=False, eps=1e-6):
    """IoU / GIoU of axis-aligned boxes in (x1,y1,z1,x2,y2,z2) form (loss_utils.py:419-538)."""
    assert mode in ("iou", "giou")
    assert bboxes1.size(-1) == 6 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 6 or bboxes2.size(0) == 0
    batch_shape = bboxes1.shape[:-2]
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new(batch_shape + ((rows,) if is_aligned else (rows, cols)))

    def vol(b):
        return (b[..., 3] - b[..., 0]) * (b[..., 4] - b[..., 1]) * (b[..., 5] - b[..., 2])
    a1, a2 = vol(bboxes1), vol(bboxes2)
    if is_aligned:
        p1, p2 = bboxes1, bboxes2
        union_base = a1 + a2
    else:
        p1, p2 = bboxes1[..., :, None, :], bboxes2[..., None, :, :]
        union_base = a1[..., None] + a2[..., None, :]
    lt, rb = torch.max(p1[..., :3], p2[..., :3]), torch.min(p1[..., 3:], p2[..., 3:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1] * wh[..., 2]
    union = torch.clamp(union_base - overlap, min=eps)
    ious = overlap / union
    if mode == "iou":
        return ious
    ewh = (torch.max(p1[..., 3:], p2[..., 3:]) - torch.min(p1[..., :3], p2[..., :3])).clamp(min=0)
    earea = torch.clamp(ewh[..., 0] * ewh[..., 1] * ewh[..., 2], min=eps)
    return ious - (earea - union) / earea


class AxisAlignedBboxOverlaps3D(object):
    def __call__(self, bboxes1, bboxes2, mode="iou", is_aligned=False):
        assert bboxes1.size(-1) == bboxes2.size(-1) == 6
        return axis_aligned_bbox_overlaps_3d(bboxes1, bboxes2, mode, is_aligned)


def _expand_onehot_labels(labels, label_weights, label_channels, ignore_index):
    bin_labels = labels.new_full((labels.size(0), label_channels), 0)
    valid = (labels >= 0) & (labels != ignore_index)
    inds = torch.nonzero(valid & (labels < label_channels), as_tuple=False)
    if inds.numel() > 0:
        bin_labels[inds, labels[inds]] = 1
    valid = valid.view(-1, 1).expand(labels.size(0), label_channels).float()
    if label_weights is None:
        w = valid
    else:
        w = label_weights.view(-1, 1).repeat(1, label_channels) * valid
    return bin_labels, w, valid


def binary_cross_entropy(pred, label, weight=None, reduction="mean", avg_factor=None, class_weight=None,
                         ignore_index=-100):
    """Sigmoid BCE with avg_factor + eps in the mean (loss_utils.py:810-846)."""
    if pred.dim() != label.dim():
        label, weight, valid = _expand_onehot_labels(label, weight, pred.size(-1), ignore_index)
    else:
        valid = ((label >= 0) & (label != ignore_index)).float()
        weight = valid if weight is None else weight * valid
    if avg_factor is None and reduction == "mean":
        avg_factor = valid.sum().item()
    loss = F.binary_cross_entropy_with_logits(pred, label.float(), pos_weight=class_weight, reduction="none")
    loss = loss * weight.float()
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == "mean":
        return loss.sum() / (avg_factor + torch.finfo(torch.float32).eps)
    if reduction != "none":
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


class CrossEntropy(nn.Module):
    def __init__(self, use_sigmoid=True, reduction="mean", class_weight=None, loss_weight=1.0, ignore_index=-100):
        super().__init__()
        assert use_sigmoid, "Now we only support sigmoid implementation."
        self.reduction, self.class_weight = reduction, class_weight
        self.loss_weight, self.ignore_index = loss_weight, ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, ignore_index=None,
                **kwargs):
        assert reduction_override in (None, "none", "mean", "sum")
        cw = None if self.class_weight is None else cls_score.new_tensor(self.class_weight)
        return self.loss_weight * binary_cross_entropy(
            cls_score, label, weight, class_weight=cw, reduction=reduction_override or self.reduction,
            avg_factor=avg_factor, ignore_index=self.ignore_index if ignore_index is None else ignore_index, **kwargs)


def py_sigmoid_focal_loss(pred, target, weight=None, gamma=2.0, alpha=0.25, reduction="mean", avg_factor=None):
    p = pred.sigmoid()
    target = target.type_as(pred)
    pt = (1 - p) * target + p * (1 - target)
    fw = (alpha * target + (1 - alpha) * (1 - target)) * pt.pow(gamma)
    loss = F.binary_cross_entropy_with_logits(pred, target, reduction="none") * fw
    if weight is not None and weight.shape != loss.shape:
        if weight.size(0) == loss.size(0):
            weight = weight.view(-1, 1)
        else:
            assert weight.numel() == loss.numel()
            weight = weight.view(loss.size(0), -1)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class FocalLoss(nn.Module):
    """Sigmoid focal loss; label -1 = background (loss_utils.py:964-1040).  NOTE: like the reference
    (:1024) this REWRITES `target` in place (-1 -> num_classes)."""

    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, reduction="mean", loss_weight=1.0):
        super().__init__()
        assert use_sigmoid is True, "Only sigmoid focal loss supported now."
        self.gamma, self.alpha, self.reduction, self.loss_weight = gamma, alpha, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        assert reduction_override in (None, "none", "mean", "sum")
        nc = pred.size(1)
        target[target < 0] = nc
        onehot = F.one_hot(target.long(), num_classes=nc + 1)[:, :nc]
        return self.loss_weight * py_sigmoid_focal_loss(pred, onehot, weight, gamma=self.gamma, alpha=self.alpha,
                                                        reduction=reduction_override or self.reduction,
                                                        avg_factor=avg_factor)


def smooth_l1_loss(pred, target, weight=None, beta=1.0, reduction="mean", avg_factor=None):
    assert beta > 0
    assert pred.size() == target.size() and target.numel() > 0
    d = torch.abs(pred - target)
    loss = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    return weight_reduce_loss(loss, weight, reduction, avg_factor)


class SmoothL1Loss(nn.Module):
    def __init__(self, beta=1.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, "none", "mean", "sum")
        return self.loss_weight * smooth_l1_loss(pred, target, weight, beta=self.beta,
                                                 reduction=reduction_override or self.reduction,
                                                 avg_factor=avg_factor, **kwargs)
