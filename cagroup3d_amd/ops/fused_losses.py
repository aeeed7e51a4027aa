"""Fused loss terms of the dense head through the C-ABI (one pass forward, one pass backward each):

* `positives_loss`   centerness BCE + axis-aligned IoU loss over the positive points of the class maps
                     (reference dense_heads/cagroup_head.py:532-546 with `_bbox_pred_to_bbox` :654-668 and
                     `axis_aligned_bbox_overlaps_3d`, pcdet/utils/loss_utils.py:419-538);
* `smooth_l1_rows`   smooth-L1 with per-row weights, reduction 'sum' (the vote loss, cagroup_head.py:512-519).
"""
from ctypes import c_float, c_int32, c_int64

import torch

from .. import _lib
from .._lib import ptr


class _PositivesLoss(torch.autograd.Function):
    """bbox_pred [N,6]: the ScanNet form (axis-aligned IoU, cg3d_pos_loss); [N,8]: the yaw form ('fcaf3d' decode + rotated IoU,
    cg3d_pos_loss_yaw of include/cagroup3d_stages.h)."""

    @staticmethod
    def forward(ctx, centerness, bbox_pred, points, ctr_t, bbox_t, scene, n_pos, ctr_denorm, pos, wc, wb, eps):
        lib = _lib.get()
        cent = centerness.contiguous().view(-1)
        bbox = bbox_pred.contiguous()
        assert bbox.shape[1] in (6, 8) and cent.shape[0] == bbox.shape[0]
        ctx.sfx = sfx = "_yaw" if bbox.shape[1] == 8 else ""
        pts, ct, bt = points.contiguous(), ctr_t.to(torch.float32).contiguous(), bbox_t.to(torch.float32).contiguous()
        sc, ps = scene.to(torch.int64).contiguous(), pos.to(torch.int64).contiguous()
        npn, cdn = n_pos.to(torch.float32).contiguous(), ctr_denorm.to(torch.float32).contiguous()
        lib.check(cent, bbox, pts, ct, bt, sc, ps, npn, cdn)
        npos = int(ps.shape[0])
        nb = int(lib.raw("cg3d_pos_loss%s_nblocks" % sfx)(npos))
        partial = torch.empty((nb, 2), dtype=torch.float32, device=cent.device)
        lib.call("cg3d_pos_loss%s_fwd" % sfx, ptr(cent), ptr(bbox), ptr(pts), ptr(ct), ptr(bt), c_int32(bt.shape[1]), ptr(sc), ptr(npn),
                 ptr(cdn), ptr(ps), c_int64(npos), c_float(wc), c_float(wb), c_float(eps), ptr(partial), lib.stream())
        ctx.save_for_backward(cent, bbox, pts, ct, bt, sc, npn, cdn, ps)
        ctx.meta = (float(wc), float(wb), float(eps), tuple(centerness.shape))
        return partial.sum(0)                       # [loss_centerness, loss_bbox]

    @staticmethod
    def backward(ctx, g):
        cent, bbox, pts, ct, bt, sc, npn, cdn, ps = ctx.saved_tensors
        wc, wb, eps, cshape = ctx.meta
        lib = _lib.get()
        g = g.to(torch.float32).contiguous()
        dcent = torch.zeros_like(cent)              # rows that are not positives get no gradient
        dbbox = torch.zeros_like(bbox)
        lib.call("cg3d_pos_loss%s_bwd" % ctx.sfx, ptr(cent), ptr(bbox), ptr(pts), ptr(ct), ptr(bt), c_int32(bt.shape[1]), ptr(sc), ptr(npn),
                 ptr(cdn), ptr(ps), c_int64(ps.shape[0]), c_float(wc), c_float(wb), c_float(eps), ptr(g), ptr(dcent), ptr(dbbox),
                 lib.stream())
        return (dcent.view(cshape), dbbox) + (None,) * 10


def positives_loss(centerness, bbox_pred, points, ctr_t, bbox_t, scene, n_pos, ctr_denorm, pos, wc, wb, eps):
    """-> tensor [2] = (sum_i wc / (n_pos[scene_i] + eps) * BCE_i,  sum_i wb * ctr_t_i / ctr_denorm[scene_i] * (1 - IoU_i)), i over
    the rows `pos`; gradients flow to `centerness` and `bbox_pred` ([N,6] face distances)."""
    return _PositivesLoss.apply(centerness, bbox_pred, points, ctr_t, bbox_t, scene, n_pos, ctr_denorm, pos, wc, wb, eps)


class _SmoothL1Rows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, w, beta):
        lib = _lib.get()
        pred, target, w = pred.contiguous(), target.to(torch.float32).contiguous(), w.to(torch.float32).contiguous().view(-1)
        lib.check(pred, target, w)
        n, d = pred.shape
        nb = int(lib.raw("cg3d_focal_loss_nblocks")(n, d))
        partial = torch.empty(nb, dtype=torch.float32, device=pred.device)
        lib.call("cg3d_smooth_l1_rows_fwd", ptr(pred), ptr(target), ptr(w), c_int64(n), c_int32(d), c_float(beta), ptr(partial),
                 lib.stream())
        ctx.save_for_backward(pred, target, w)
        ctx.beta = float(beta)
        return partial.sum()

    @staticmethod
    def backward(ctx, g):
        pred, target, w = ctx.saved_tensors
        lib = _lib.get()
        g = g.to(torch.float32).contiguous().view(-1)
        dpred = torch.empty_like(pred)
        lib.call("cg3d_smooth_l1_rows_bwd", ptr(pred), ptr(target), ptr(w), ptr(g), c_int64(pred.shape[0]), c_int32(pred.shape[1]),
                 c_float(ctx.beta), ptr(dpred), lib.stream())
        return dpred, None, None, None


def smooth_l1_rows(pred, target, w, beta):
    """sum_i w[i] * sum_j smooth_l1(pred[i,j] - target[i,j]; beta)"""
    return _SmoothL1Rows.apply(pred, target, w, beta)
