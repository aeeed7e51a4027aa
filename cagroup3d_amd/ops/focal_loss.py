"""Fused sigmoid focal loss with per-row weights (the element-wise chain of `py_sigmoid_focal_loss`, reference
pcdet/utils/loss_utils.py:903-961, as one pass forward and one pass backward through the C-ABI)."""
from ctypes import c_float, c_int32, c_int64

import torch

from .. import _lib
from .._lib import ptr


class _FocalRowsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, label, row_w, gamma, alpha):
        lib = _lib.get()
        pred = pred.contiguous()
        label = label.to(torch.int32).contiguous()
        row_w = row_w.to(torch.float32).contiguous()
        lib.check(pred, label, row_w)
        n, c = pred.shape
        nb = int(lib.raw("cg3d_focal_loss_nblocks")(n, c))
        partial = torch.empty(nb, dtype=torch.float32, device=pred.device)
        lib.call("cg3d_focal_loss_fwd", ptr(pred), ptr(label), ptr(row_w), c_int64(n), c_int32(c), c_float(gamma),
                 c_float(alpha), ptr(partial), lib.stream())
        ctx.save_for_backward(pred, label, row_w)
        ctx.ga = (float(gamma), float(alpha))
        return partial.sum()

    @staticmethod
    def backward(ctx, g):
        pred, label, row_w = ctx.saved_tensors
        lib = _lib.get()
        n, c = pred.shape
        g = g.to(torch.float32).contiguous()
        dpred = torch.empty_like(pred)
        lib.call("cg3d_focal_loss_bwd", ptr(pred), ptr(label), ptr(row_w), ptr(g), c_int64(n), c_int32(c),
                 c_float(ctx.ga[0]), c_float(ctx.ga[1]), ptr(dpred), lib.stream())
        return dpred, None, None, None, None


def sigmoid_focal_loss_rows(pred, label, row_w, gamma=2.0, alpha=0.25):
    """sum_i row_w[i] * sum_c focal(pred[i, c], label[i] == c).  `label` outside [0, C) = background row
    (FocalLoss.forward rewrites -1 to C, loss_utils.py:1024)."""
    return _FocalRowsFunction.apply(pred, label, row_w, gamma, alpha)
