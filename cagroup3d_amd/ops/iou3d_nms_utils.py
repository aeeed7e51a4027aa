"""Mirror of pcdet/ops/iou3d_nms/iou3d_nms_utils.py:48-116 on the gfx950 C-ABI.

Same function names, argument meaning and return values as the reference.  Differences by design:
the greedy NMS scan runs on the device (no mask D2H copy, no host loop); only the kept COUNT is read
back, because the reference API returns a variable-length index tensor."""
from ctypes import c_float, c_int32, c_int64

import torch

from .. import _lib
from .._lib import ptr


def _pair(name, boxes_a, boxes_b):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    lib = _lib.get()
    a, b = boxes_a.contiguous().float(), boxes_b.contiguous().float()
    lib.check(a, b)
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    lib.call(name, ptr(a), c_int64(a.shape[0]), ptr(b), c_int64(b.shape[0]), ptr(out), lib.stream())
    return out


def points_in_boxes(points, boxes, point_seg=None, box_seg=None):
    """(n,3) x (g,7) -> bool (n,g): strictly inside the rotated box (find_points_in_boxes, cagroup3d_assigner.py:9-36);
    with the two int32 segment vectors a point only counts for boxes of its own segment (scene)."""
    lib = _lib.get()
    p, b = points[:, :3].contiguous().float(), boxes[:, :7].contiguous().float()
    ps = point_seg.to(torch.int32).contiguous() if point_seg is not None else None
    bs = box_seg.to(torch.int32).contiguous() if box_seg is not None else None
    lib.check(p, b, ps, bs)
    out = torch.empty((p.shape[0], b.shape[0]), dtype=torch.uint8, device=p.device)
    lib.call("cg3d_points_in_boxes", ptr(p), c_int64(p.shape[0]), ptr(b), c_int32(b.shape[0]), ptr(ps), ptr(bs), ptr(out),
             lib.stream())
    return out.bool()


def boxes_bev_iou_cpu(boxes_a, boxes_b):
    """(N,7),(M,7) CPU tensors or numpy arrays -> (N,M) rotated BEV IoU on the host (iou3d_nms_utils.py:12-29 over
    `boxes_iou_bev_cpu`, iou3d_cpu.cpp:232-252): the library's host instantiation of the functions its kernels run."""
    is_numpy = not torch.is_tensor(boxes_a)
    a = torch.as_tensor(boxes_a).float().contiguous()
    b = torch.as_tensor(boxes_b).float().contiguous()
    assert not (a.is_cuda or b.is_cuda), "Only support CPU tensors"
    assert a.shape[1] == 7 and b.shape[1] == 7
    out = a.new_zeros((a.shape[0], b.shape[0]))
    _lib.get().call("cg3d_boxes_iou_bev_cpu", ptr(a), c_int64(a.shape[0]), ptr(b), c_int64(b.shape[0]), ptr(out))
    return out.numpy() if is_numpy else out


def boxes_overlap_bev(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) rotated BEV intersection area (iou3d_nms.cpp:49-66)."""
    return _pair("cg3d_boxes_overlap_bev", boxes_a, boxes_b)


def boxes_iou_bev(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) rotated BEV IoU (iou3d_nms_utils.py:32-45)."""
    return _pair("cg3d_boxes_iou_bev", boxes_a, boxes_b)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N,7),(M,7) -> (N,M) 3D IoU = BEV overlap x height overlap / union (iou3d_nms_utils.py:48-81)."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_hmax = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_hmin = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_hmax = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_hmin = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = boxes_overlap_bev(boxes_a, boxes_b)
    overlaps_h = torch.clamp(torch.min(a_hmax, b_hmax) - torch.max(a_hmin, b_hmin), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _nms_sorted(boxes_sorted, thresh, rotated):
    """boxes already sorted by descending score -> (keep int64 [n] device, num_keep int32 [1] device)."""
    lib = _lib.get()
    lib.check(boxes_sorted)
    n = boxes_sorted.shape[0]
    dev = boxes_sorted.device
    cb = (n + 63) // 64
    mask = torch.empty(max(n * cb, 1), dtype=torch.int64, device=dev)
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    num = torch.zeros(1, dtype=torch.int32, device=dev)
    lib.call("cg3d_nms", ptr(boxes_sorted), c_int64(n), c_float(thresh), c_int32(1 if rotated else 0), ptr(mask),
             ptr(keep), ptr(num), lib.stream())
    return keep, num, mask


def _nms(boxes, scores, thresh, rotated, pre_maxsize=None):
    assert boxes.shape[1] == 7
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].contiguous().float()
    keep, num, _ = _nms_sorted(b, float(thresh), rotated)
    return order[keep[:int(num.item())]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated-BEV NMS (iou3d_nms_utils.py:84-100): indices into `boxes`, best first."""
    return _nms(boxes, scores, thresh, True, pre_maxsize)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """Axis-aligned-BEV NMS (iou3d_nms_utils.py:103-116)."""
    return _nms(boxes, scores, thresh, False)


def nms_batched_sorted(boxes_sorted, seg_off_cpu, thresh, rotated):
    """Many independent NMS problems in two launches.

    boxes_sorted: (sum n_s, 7), each segment sorted by descending score; seg_off_cpu: python list /
    CPU int64 tensor of nseg+1 offsets.  Returns (keep int64 [sum n_s] with per-segment LOCAL indices
    at the segment's offset, num_keep int32 [nseg]) -- both on the device, no host sync."""
    lib = _lib.get()
    lib.check(boxes_sorted)
    import numpy as np
    from ..me import h2d
    seg = np.asarray(seg_off_cpu, dtype=np.int64)
    nseg = seg.shape[0] - 1
    sizes = seg[1:] - seg[:-1]
    max_seg = int(sizes.max()) if nseg > 0 else 0
    moff = np.zeros(nseg + 1, dtype=np.int64)
    moff[1:] = np.cumsum(sizes * ((sizes + 63) // 64))
    dev = boxes_sorted.device
    seg_d, moff_d = h2d(torch.from_numpy(seg), torch.int64, dev), h2d(torch.from_numpy(moff[:-1].copy()), torch.int64, dev)
    mask = torch.empty(max(int(moff[-1]), 1), dtype=torch.int64, device=dev)
    keep = torch.empty(max(boxes_sorted.shape[0], 1), dtype=torch.int64, device=dev)
    num = torch.zeros(max(nseg, 1), dtype=torch.int32, device=dev)
    lib.call("cg3d_nms_batched", ptr(boxes_sorted), ptr(seg_d), ptr(moff_d), c_int32(nseg), c_int64(max_seg),
             c_float(thresh), c_int32(1 if rotated else 0), ptr(mask), ptr(keep), ptr(num), lib.stream())
    return keep, num[:nseg]


class iou3d_nms_cuda:  # noqa: N801  (the reference's extension module name, iou3d_nms_api.cpp:11-17)
    """The reference extension's two NMS entry points in their literal shape -- `nms_gpu(boxes, keep, thresh) -> num_kept`
    with `boxes` sorted by descending score on the device and `keep` an int64 HOST tensor (iou3d_nms.h:9-12,
    iou3d_nms.cpp:90-186) -- over cg3d_nms_gpu / cg3d_nms_normal_gpu.  Blocks the host like the reference; the detector
    itself uses the device-resident forms above."""

    @staticmethod
    def _run(name, boxes, keep, thresh):
        lib = _lib.get()
        boxes = boxes.contiguous().float()
        lib.check(boxes)
        assert keep.dtype == torch.int64 and not keep.is_cuda and keep.is_contiguous() and keep.numel() >= boxes.shape[0]
        n = boxes.shape[0]
        ws = torch.empty(max(int(lib.raw("cg3d_nms_gpu_ws_bytes")(n)), 16), dtype=torch.uint8, device=boxes.device)
        rc = lib.raw(name)(ptr(boxes), c_int64(n), keep.data_ptr(), c_float(thresh), ptr(ws), lib.stream())
        if rc < 0:
            raise _lib.CG3DError("%s failed: %d" % (name, rc))
        return int(rc)

    @staticmethod
    def nms_gpu(boxes, keep, thresh):
        return iou3d_nms_cuda._run("cg3d_nms_gpu", boxes, keep, thresh)

    @staticmethod
    def nms_normal_gpu(boxes, keep, thresh):
        return iou3d_nms_cuda._run("cg3d_nms_normal_gpu", boxes, keep, thresh)
