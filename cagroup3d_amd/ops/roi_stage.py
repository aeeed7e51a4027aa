"""Second-stage (RoI head) training ops through the C-ABI of include/cagroup3d_stages.h:

* `roi_match`        best same-class 3D IoU of every (padded) RoI against its scene's ground truth
                     (reference roi_heads/target_assigner/cagroup_proposal_target_layer.py:204-238);
* `roi_targets`      sampled RoIs -> RoIs, matched boxes, canonical-frame boxes, masks, regression targets
                     (:22-63, roi_heads/cagroup_roi_head.py:291-326,551-577);
* `roi_grid_coords`  7^3 grid points of every RoI quantised to voxel coordinates (cagroup_roi_head.py:46-68,199-261);
* `roi_reg_loss`     masked code-weighted smooth-L1 (cagroup_roi_head.py:551-590).

Each replaces a chain of 15-60 tensor launches over a few hundred rows by one launch."""
from ctypes import c_float, c_int32, c_int64

import numpy as np
import torch

from .. import _lib
from .._lib import ptr


def roi_match(boxes, labels, roi_off, nb, rin, enlarge, gt_boxes, n_gt):
    """-> (max_ov float32 [nb*rin], assign int32 [nb*rin]); roi_off / n_gt: int32 device tensors [nb+1] / [nb]."""
    lib = _lib.get()
    boxes, labels, gt_boxes = boxes.contiguous(), labels.contiguous(), gt_boxes.contiguous()
    assert boxes.dtype == torch.float32 and labels.dtype == torch.int64 and gt_boxes.dtype == torch.float32
    lib.check(boxes, labels, roi_off, gt_boxes, n_gt)
    dev = gt_boxes.device
    max_ov = torch.empty(nb * rin, dtype=torch.float32, device=dev)
    assign = torch.empty(nb * rin, dtype=torch.int32, device=dev)
    lib.call("cg3d_roi_match", ptr(boxes), ptr(labels), ptr(roi_off), c_int32(nb), c_int32(rin), c_float(enlarge),
             ptr(gt_boxes), c_int32(gt_boxes.shape[1]), c_int32(gt_boxes.shape[2]), ptr(n_gt), ptr(max_ov), ptr(assign), lib.stream())
    return max_ov, assign


def roi_targets(boxes, scores, labels, roi_off, nb, rin, enlarge, gt_boxes, max_ov, assign, keep, rsel, code_size, reg_fg,
                cls_fg, cls_bg):
    """keep int32 [nb*rsel] (device) -> dict with the keys ProposalTargetLayer.forward + assign_targets produce, plus
    'reg_targets' [nb*rsel, code_size]."""
    lib = _lib.get()
    boxes, scores, labels, gt_boxes = boxes.contiguous(), scores.contiguous(), labels.contiguous(), gt_boxes.contiguous()
    lib.check(boxes, scores, labels, roi_off, gt_boxes, max_ov, assign, keep)
    dev, m = gt_boxes.device, nb * rsel
    blk = torch.empty((3 * m * 7 + 4 * m + m * code_size,), dtype=torch.float32, device=dev)       # one allocation for the float outputs
    off = 0

    def take(n, shape):
        nonlocal off
        t = blk[off:off + n].view(shape)
        off += n
        return t
    o_rois, o_src, o_gt = take(m * 7, (nb, rsel, 7)), take(m * 7, (nb, rsel, 7)), take(m * 7, (nb, rsel, 7))
    o_gl, o_iou, o_sc, o_cl = take(m, (nb, rsel)), take(m, (nb, rsel)), take(m, (nb, rsel)), take(m, (nb, rsel))
    o_rt = take(m * code_size, (m, code_size))
    li = torch.empty((2, nb, rsel), dtype=torch.int64, device=dev)
    lib.call("cg3d_roi_targets", ptr(boxes), ptr(scores), ptr(labels), ptr(roi_off), c_int32(nb), c_int32(rin), c_float(enlarge),
             ptr(gt_boxes), c_int32(gt_boxes.shape[1]), c_int32(gt_boxes.shape[2]), ptr(max_ov), ptr(assign), ptr(keep),
             c_int32(rsel), c_int32(code_size), c_float(reg_fg), c_float(cls_fg), c_float(cls_bg), c_float(cls_fg - cls_bg),
             ptr(o_rois), ptr(o_src), ptr(o_gt), ptr(o_gl), ptr(o_iou), ptr(o_sc), ptr(li[0]), ptr(li[1]), ptr(o_cl), ptr(o_rt),
             lib.stream())
    return {"rois": o_rois, "gt_of_rois": o_gt, "gt_of_rois_src": o_src, "gt_label_of_rois": o_gl, "gt_iou_of_rois": o_iou,
            "roi_scores": o_sc, "roi_labels": li[0], "reg_valid_mask": li[1], "rcnn_cls_labels": o_cl, "reg_targets": o_rt}


def roi_grid_coords(rois, rois_per_scene, grid, with_yaw, voxel_size, clamp_lo, clamp_hi, coord_key):
    """rois float32 [n,7] -> int32 [n * grid^3, 4] (scene, x, y, z) voxel coordinates, RoI-major, duplicates kept."""
    lib = _lib.get()
    rois = rois.contiguous()
    lib.check(rois)
    n = rois.shape[0]
    out = torch.empty((n * grid ** 3, 4), dtype=torch.int32, device=rois.device)
    lib.call("cg3d_roi_grid_coords", ptr(rois), c_int64(n), c_int32(rois_per_scene), c_int32(grid), c_int32(1 if with_yaw else 0),
             c_float(voxel_size), c_float(clamp_lo), c_float(clamp_hi), c_int32(coord_key), ptr(out), lib.stream())
    return out


class _RoiRegLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reg, target, valid, code_w, beta, weight):
        lib = _lib.get()
        reg, target, valid = reg.contiguous(), target.contiguous(), valid.contiguous().view(-1)
        assert valid.dtype == torch.int64 and reg.shape == target.shape
        lib.check(reg, target, valid, code_w)
        m, cs = reg.shape
        out = torch.empty(2, dtype=torch.float32, device=reg.device)
        lib.call("cg3d_roi_reg_loss_fwd", ptr(reg), ptr(target), ptr(valid), ptr(code_w), c_int64(m), c_int32(cs), c_float(beta),
                 c_float(weight), ptr(out), lib.stream())
        ctx.save_for_backward(reg, target, valid, code_w, out)
        ctx.meta = (float(beta), float(weight))
        return out[0]

    @staticmethod
    def backward(ctx, g):
        reg, target, valid, code_w, out = ctx.saved_tensors
        beta, weight = ctx.meta
        lib = _lib.get()
        g = g.to(torch.float32).contiguous().view(-1)
        dreg = torch.empty_like(reg)
        lib.call("cg3d_roi_reg_loss_bwd", ptr(reg), ptr(target), ptr(valid), ptr(code_w), c_int64(reg.shape[0]),
                 c_int32(reg.shape[1]), c_float(beta), c_float(weight), ptr(out), ptr(g), ptr(dreg), lib.stream())
        return dreg, None, None, None, None, None


def roi_reg_loss(reg, target, valid, code_w, beta, weight):
    """weight / max(#valid, 1) * sum_{valid rows} smooth_l1((reg - target) * code_w; beta) -> scalar; gradient to `reg`."""
    return _RoiRegLoss.apply(reg, target, valid, code_w, beta, weight)


def subsample_rois_host(ov, roi_per_image, fg_ratio, reg_fg_thresh, cls_fg_thresh, cls_bg_thresh_l0, hard_bg_ratio):
    """ProposalTargetLayer.subsample_rois / sample_bg_inds (cagroup_proposal_target_layer.py:127-202) on a numpy row of best
    overlaps: the same pools and the same host RNG calls in the same order (np.random.permutation / np.random.rand for the
    foreground, torch.randint for the background) -- index arithmetic in numpy instead of ~40 tiny CPU tensor ops per scene."""
    fg_per_image = int(np.round(fg_ratio * roi_per_image))
    fg_thresh = min(reg_fg_thresh, cls_fg_thresh)
    fg_inds = np.nonzero(ov >= np.float32(fg_thresh))[0]
    easy_bg = np.nonzero(ov < np.float32(cls_bg_thresh_l0))[0]
    hard_bg = np.nonzero((ov < np.float32(reg_fg_thresh)) & (ov >= np.float32(cls_bg_thresh_l0)))[0]
    n_fg, n_bg = fg_inds.size, hard_bg.size + easy_bg.size

    def draw(pool, k):
        return pool[torch.randint(low=0, high=int(pool.size), size=(k,)).numpy()]

    def sample_bg(n):
        if hard_bg.size > 0 and easy_bg.size > 0:
            n_hard = min(int(n * hard_bg_ratio), int(hard_bg.size))
            hard = draw(hard_bg, n_hard)
            return np.concatenate([hard, draw(easy_bg, n - n_hard)])
        if hard_bg.size > 0:
            return draw(hard_bg, n)
        if easy_bg.size > 0:
            return draw(easy_bg, n)
        raise NotImplementedError
    if n_fg > 0 and n_bg > 0:
        take = min(fg_per_image, n_fg)
        perm = np.random.permutation(n_fg)
        fg = fg_inds[perm[:take]]
        bg = sample_bg(roi_per_image - take)
    elif n_fg > 0:
        rnd = np.floor(np.random.rand(roi_per_image) * n_fg).astype(np.int64)
        fg = fg_inds[rnd]
        bg = fg[fg < 0]
    elif n_bg > 0:
        fg = fg_inds
        bg = sample_bg(roi_per_image)
    else:
        raise NotImplementedError("no RoIs to sample: FG=%d BG=%d" % (n_fg, n_bg))
    return np.concatenate([fg, bg]).astype(np.int32)
