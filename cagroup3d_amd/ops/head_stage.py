"""Dense-head stage ops through the C-ABI of include/cagroup3d_stages.h:

* `class_rows`    selection per class + pad voxels + the [votes ; originals] point set + its two quantisations
                  (reference dense_heads/cagroup_head.py:209-258) -- three launches and ONE host read in place of ~60 tensor
                  launches (nonzero, argsort, the index arithmetic of the class-major layout, two floors ...);
* `gather_rows2`  rows of a table held in two pieces (the reference concatenates [vote features ; backbone features] first).
"""
from ctypes import c_float, c_int32, c_int64

import torch

from .. import _lib
from .. import me as _me
from .._lib import ptr


def class_rows(hit, coords, pad_row, offsets, n_vote, voxel_size, ts, vs_tab, expand, n_batch, before_read=None):
    """hit bool / uint8 [N, C]; coords int32 [N,4]; pad_row int32 [B]; offsets float32 [N, n_vote*3]; vs_tab float32 [C,3].
    -> (src int32 [T], fine int32 [T,4], coarse int32 [T,4], sel list[C]) with T = (sum(sel) + C*B) * (n_vote + 1)."""
    lib = _lib.get()
    N, C = hit.shape
    hit8 = hit.contiguous().view(torch.uint8) if hit.dtype == torch.bool else hit.contiguous()
    coords, pad_row, offsets, vs_tab = coords.contiguous(), pad_row.contiguous(), offsets.contiguous(), vs_tab.contiguous()
    assert coords.dtype == torch.int32 and pad_row.dtype == torch.int32 and offsets.dtype == torch.float32
    lib.check(hit8, coords, pad_row, offsets, vs_tab)
    dev = coords.device
    nblk = int(lib.raw("cg3d_class_nblk")(N))
    work = torch.empty((C + 6) * nblk + C + 6, dtype=torch.int32, device=dev)
    block_off, totals = work[:(C + 6) * nblk], work[(C + 6) * nblk:]
    lib.call("cg3d_class_count", ptr(hit8), c_int64(N), c_int32(C), ptr(coords), ptr(block_off), ptr(totals), lib.stream())
    if before_read is not None:
        # the caller's data-only host reads (ground-truth padding mask, scene sizes): behind the counting launch, so that THIS
        # stage's read -- the first of the step that waits for the device -- finds its result ready when they return
        before_read()
    sel = totals[:C].tolist()                                     # the stage's host read (the reference: torch.nonzero)
    # the device-bound half of the step ends with this read: what was kept out of it (the class branches' optimizer rows and
    # weight copies, me.LATE_MODE "defer") is launched now, under the host's work on the class maps
    _me.run_late(join=False)
    T = (sum(sel) + C * n_batch) * (n_vote + 1)
    out = torch.empty((T, 9), dtype=torch.int32, device=dev)      # one allocation: fine | coarse | src (16-byte aligned rows first)
    flat = out.view(-1)
    fine, coarse, src = flat[:4 * T].view(T, 4), flat[4 * T:8 * T].view(T, 4), flat[8 * T:]
    lib.call("cg3d_class_rows", ptr(hit8), c_int64(N), c_int32(C), c_int32(n_batch), ptr(block_off), ptr(totals), ptr(coords),
             ptr(pad_row), ptr(offsets), c_int32(n_vote), c_float(voxel_size), c_int32(ts), ptr(vs_tab), c_int32(expand),
             ptr(src), ptr(fine), ptr(coarse), lib.stream())
    return src, fine, coarse, sel


class _GatherRows2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fa, fb, idx):
        lib = _lib.get()
        fa, fb = fa.contiguous(), fb.contiguous()
        lib.check(fa, fb, idx)
        n, c = idx.shape[0], fa.shape[1]
        assert fb.shape[1] == c and idx.dtype == torch.int32
        out = torch.empty((n, c), dtype=torch.float32, device=fa.device)
        lib.call("cg3d_gather_rows2", ptr(fa), ptr(fb), c_int64(fa.shape[0]), ptr(idx), ptr(out), c_int64(n), c_int32(c), lib.stream())
        ctx.save_for_backward(idx)
        ctx.shapes = (fa.shape[0], fb.shape[0], c)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        na, nb, c = ctx.shapes
        lib = _lib.get()
        dout = dout.contiguous()
        d = torch.zeros((na + nb, c), dtype=torch.float32, device=dout.device)       # one fill for both pieces
        lib.call("cg3d_scatter_add_rows2", ptr(dout), ptr(idx), ptr(d), ptr(d[na:]), c_int64(na), c_int64(idx.shape[0]), c_int32(c),
                 lib.stream())
        return d[:na], d[na:], None


def gather_rows2(fa, fb, idx):
    """out[i] = cat([fa, fb])[idx[i]] without the concatenation; gradients are scattered back into the two pieces."""
    return _GatherRows2.apply(fa, fb, idx)


def vote_targets(xyz, ins, sem, gt_ctr, n_gt, n_classes, n_ins, vox_xyz, vox_scene, nearest):
    """ScanNet-form vote targets of all scenes (reference cagroup_head.py:454-498) in two calls: per-instance bounding boxes and
    the ground-truth centre each instance votes for (cg3d_instance_centers), then the masked per-voxel offsets
    (cg3d_vote_targets).  xyz float32 [B,P,3]; ins / sem int64 [B,P]; gt_ctr float32 [B,G,3], n_gt int32 [B]; vox_xyz [N,3],
    vox_scene / nearest int64 [N] -> (off_t [N,3], off_m float32 [N])."""
    lib = _lib.get()
    xyz, ins, sem, gt_ctr = xyz.contiguous(), ins.contiguous(), sem.contiguous(), gt_ctr.contiguous()
    vox_xyz, vox_scene, nearest = vox_xyz.contiguous(), vox_scene.contiguous(), nearest.contiguous()
    assert ins.dtype == torch.int64 and sem.dtype == torch.int64 and vox_scene.dtype == torch.int64 and nearest.dtype == torch.int64
    lib.check(xyz, ins, sem, gt_ctr, n_gt, vox_xyz, vox_scene, nearest)
    B, P = ins.shape
    dev = xyz.device
    centers = torch.empty((B, n_ins, 3), dtype=torch.float32, device=dev)
    ws = torch.empty(8 * B * n_ins, dtype=torch.int32, device=dev)
    lib.call("cg3d_instance_centers", ptr(xyz), ptr(ins), ptr(sem), c_int32(B), c_int32(P), c_int32(n_ins), ptr(gt_ctr),
             c_int32(gt_ctr.shape[1]), ptr(n_gt), c_int32(n_classes), ptr(centers), ptr(ws), lib.stream())
    N = vox_xyz.shape[0]
    out = torch.empty((N, 4), dtype=torch.float32, device=dev)
    flat = out.view(-1)
    off_t, off_m = flat[:3 * N].view(N, 3), flat[3 * N:]
    lib.call("cg3d_vote_targets", ptr(vox_xyz), ptr(vox_scene), ptr(nearest), c_int64(N), ptr(ins), c_int32(P), ptr(centers),
             c_int32(n_ins), ptr(off_t), ptr(off_m), lib.stream())
    return off_t, off_m


class _HeadOutputs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, reg, scale, coords, vs_tab, n_batch, cls, boost):
        lib = _lib.get()
        reg, scale = reg.contiguous(), scale.contiguous()
        lib.check(reg, scale, coords, vs_tab, cls)
        n, nd = reg.shape
        assert coords.dtype == torch.int32 and coords.shape == (n, 4) and (cls is None or cls.is_contiguous())
        bbox = torch.empty_like(reg)
        points = torch.empty((n, 3), dtype=torch.float32, device=reg.device)
        lib.call("cg3d_head_outputs_fwd", ptr(reg), c_int32(nd), ptr(coords), c_int64(n), c_int32(n_batch), ptr(scale), ptr(vs_tab),
                 c_int32(scale.shape[0]), c_float(boost), ptr(cls), ptr(bbox), ptr(points), lib.stream())
        ctx.save_for_backward(reg, scale, coords, bbox)
        ctx.n_batch = n_batch
        ctx.mark_non_differentiable(points)
        ctx.set_materialize_grads(False)
        return bbox, points

    @staticmethod
    def backward(ctx, dbbox, _dpoints):
        if dbbox is None:
            return (None,) * 7
        reg, scale, coords, bbox = ctx.saved_tensors
        lib = _lib.get()
        dbbox = dbbox.contiguous()
        dreg = torch.empty_like(reg)
        dscale = torch.empty_like(scale)
        lib.call("cg3d_head_outputs_bwd", ptr(dbbox), ptr(bbox), ptr(reg), c_int32(reg.shape[1]), ptr(coords), c_int64(reg.shape[0]),
                 c_int32(ctx.n_batch), ptr(scale), c_int32(scale.shape[0]), ptr(dreg), ptr(dscale), lib.stream())
        return dreg, dscale, None, None, None, None, None


def head_outputs(reg, scale, coords, vs_tab, n_batch, cls=None, boost=0.0):
    """Class-branch prediction outputs of all class maps (reference cagroup_head.py:627-652): bbox_pred = (exp(reg[:, :6] *
    scale[class]), reg[:, 6:]) with gradients to `reg` and `scale`, points = coords * class voxel size; `boost` is added IN
    PLACE to every row's own-class logit of `cls` (a constant: the gradient of `cls` is unchanged)."""
    with torch.no_grad():
        cls_raw = cls.detach() if cls is not None and boost else None
    return _HeadOutputs.apply(reg, scale, coords, vs_tab, n_batch, cls_raw, float(boost))
