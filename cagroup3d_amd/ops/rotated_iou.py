"""Differentiable rotated 3D IoU: mirror of pcdet/ops/rotated_iou/{box_intersection_2d.py,
oriented_iou_loss.py:86-109, cuda_op/cuda_ext.py} with `sort_vertices` on the gfx950 C-ABI.

The torch half is restated (same formulas, so gradients match the reference's autograd graph);
the vertex ordering is the non-differentiable native op."""
from ctypes import c_int32

import torch
from torch.autograd import Function

from .. import _lib
from .._lib import ptr

EPSILON = 1e-8


class SortVertices(Function):
    @staticmethod
    def forward(ctx, vertices, mask, num_valid):
        """vertices (B,N,24,2) fp32, mask (B,N,24) bool, num_valid (B,N) int32 -> idx (B,N,9) int32."""
        lib = _lib.get()
        v = vertices.contiguous().float()
        mk = mask.contiguous().to(torch.uint8)
        nv = num_valid.contiguous().to(torch.int32)
        lib.check(v, mk, nv)
        b, n, m = v.shape[0], v.shape[1], v.shape[2]
        idx = torch.zeros((b, n, 9), dtype=torch.int32, device=v.device)
        lib.call("cg3d_sort_vertices", c_int32(b), c_int32(n), c_int32(m), ptr(v), ptr(mk), ptr(nv), ptr(idx),
                 lib.stream())
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, gradout):
        return None, None, None


sort_v = SortVertices.apply


def box_intersection_th(corners1, corners2):
    """Edge-edge intersections of rectangle pairs: (B,N,4,2) x2 -> points (B,N,4,4,2), mask (B,N,4,4)
    (box_intersection_2d.py:13-54)."""
    nxt = [1, 2, 3, 0]
    l1 = torch.cat([corners1, corners1[:, :, nxt, :]], dim=3)
    l2 = torch.cat([corners2, corners2[:, :, nxt, :]], dim=3)
    e1 = l1.unsqueeze(3).repeat([1, 1, 1, 4, 1])
    e2 = l2.unsqueeze(2).repeat([1, 1, 4, 1, 1])
    x1, y1, x2, y2 = e1[..., 0], e1[..., 1], e1[..., 2], e1[..., 3]
    x3, y3, x4, y4 = e2[..., 0], e2[..., 1], e2[..., 2], e2[..., 3]
    num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4)
    den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4)
    t = den_t / num
    t[num == .0] = -1.
    mask_t = (t > 0) * (t < 1)
    den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3)
    u = -den_u / num
    u[num == .0] = -1.
    mask_u = (u > 0) * (u < 1)
    mask = mask_t * mask_u
    t = den_t / (num + EPSILON)
    inter = torch.stack([x1 + t * (x2 - x1), y1 + t * (y2 - y1)], dim=-1)
    return inter * mask.float().unsqueeze(-1), mask


def box1_in_box2(corners1, corners2):
    """Corners of box1 inside box2, edges inclusive (box_intersection_2d.py:57-82)."""
    a, b, d = corners2[:, :, 0:1, :], corners2[:, :, 1:2, :], corners2[:, :, 3:4, :]
    ab, am, ad = b - a, corners1 - a, d - a
    p_ab, n_ab = torch.sum(ab * am, dim=-1), torch.sum(ab * ab, dim=-1)
    p_ad, n_ad = torch.sum(ad * am, dim=-1), torch.sum(ad * ad, dim=-1)
    c1 = (p_ab / n_ab > -1e-6) * (p_ab / n_ab < 1 + 1e-6)
    c2 = (p_ad / n_ad > -1e-6) * (p_ad / n_ad < 1 + 1e-6)
    return c1 * c2


def build_vertices(corners1, corners2, c1_in_2, c2_in_1, inters, mask_inter):
    """24 candidate vertices per pair: 4 + 4 corners, 16 intersections (box_intersection_2d.py:101-124)."""
    B, N = corners1.size()[0], corners1.size()[1]
    vertices = torch.cat([corners1, corners2, inters.view([B, N, -1, 2])], dim=2)
    mask = torch.cat([c1_in_2, c2_in_1, mask_inter.view([B, N, -1])], dim=2)
    return vertices, mask


def sort_indices(vertices, mask):
    """(box_intersection_2d.py:127-147)"""
    num_valid = torch.sum(mask.int(), dim=2).int()
    mean = torch.sum(vertices * mask.float().unsqueeze(-1), dim=2, keepdim=True) / num_valid.unsqueeze(-1).unsqueeze(-1)
    return sort_v(vertices - mean, mask, num_valid).long()


def calculate_area(idx_sorted, vertices):
    """Shoelace area of the ordered polygon (box_intersection_2d.py:150-166)."""
    idx_ext = idx_sorted.unsqueeze(-1).repeat([1, 1, 1, 2])
    sel = torch.gather(vertices, 2, idx_ext)
    total = sel[:, :, 0:-1, 0] * sel[:, :, 1:, 1] - sel[:, :, 0:-1, 1] * sel[:, :, 1:, 0]
    return torch.abs(torch.sum(total, dim=2)) / 2, sel


def oriented_box_intersection_2d(corners1, corners2):
    """(box_intersection_2d.py:169-184)"""
    inters, mask_inter = box_intersection_th(corners1, corners2)
    c12, c21 = box1_in_box2(corners1, corners2), box1_in_box2(corners2, corners1)
    vertices, mask = build_vertices(corners1, corners2, c12, c21, inters, mask_inter)
    return calculate_area(sort_indices(vertices, mask), vertices)


def box2corners_th(box):
    """(B,N,5) x,y,w,h,alpha -> (B,N,4,2) corners (oriented_iou_loss.py:6-35)."""
    B = box.size()[0]
    x, y, w, h, alpha = box[..., 0:1], box[..., 1:2], box[..., 2:3], box[..., 3:4], box[..., 4:5]
    x4 = box.new_tensor([0.5, -0.5, -0.5, 0.5]).view(1, 1, 4) * w
    y4 = box.new_tensor([0.5, 0.5, -0.5, -0.5]).view(1, 1, 4) * h
    corners = torch.stack([x4, y4], dim=-1)
    sin, cos = torch.sin(alpha), torch.cos(alpha)
    rot_T = torch.stack([torch.cat([cos, sin], dim=-1), torch.cat([-sin, cos], dim=-1)], dim=-2)
    rotated = torch.bmm(corners.view([-1, 4, 2]), rot_T.view([-1, 2, 2])).view([B, -1, 4, 2])
    rotated = rotated + torch.cat([x, y], dim=-1).unsqueeze(2)
    return rotated


def cal_iou(box1, box2):
    """(oriented_iou_loss.py:38-58)"""
    c1, c2 = box2corners_th(box1), box2corners_th(box2)
    inter_area, _ = oriented_box_intersection_2d(c1, c2)
    u = box1[:, :, 2] * box1[:, :, 3] + box2[:, :, 2] * box2[:, :, 3] - inter_area
    return inter_area / u, c1, c2, u


def cal_iou_3d(box3d1, box3d2, verbose=False):
    """(B,N,7) x,y,z,w,h,l,alpha pairs -> (B,N) rotated 3D IoU (oriented_iou_loss.py:86-109)."""
    box1, box2 = box3d1[..., [0, 1, 3, 4, 6]], box3d2[..., [0, 1, 3, 4, 6]]
    zmax1, zmin1 = box3d1[..., 2] + box3d1[..., 5] * 0.5, box3d1[..., 2] - box3d1[..., 5] * 0.5
    zmax2, zmin2 = box3d2[..., 2] + box3d2[..., 5] * 0.5, box3d2[..., 2] - box3d2[..., 5] * 0.5
    z_overlap = (torch.min(zmax1, zmax2) - torch.max(zmin1, zmin2)).clamp_min(0.)
    iou_2d, c1, c2, u = cal_iou(box1, box2)
    inter3d = iou_2d * u * z_overlap
    v1 = box3d1[..., 3] * box3d1[..., 4] * box3d1[..., 5]
    v2 = box3d2[..., 3] * box3d2[..., 4] * box3d2[..., 5]
    u3d = v1 + v2 - inter3d
    if verbose:
        z_range = (torch.max(zmax1, zmax2) - torch.min(zmin1, zmin2)).clamp_min(0.)
        return inter3d / u3d, c1, c2, z_range, u3d
    return inter3d / u3d


# ------------------------------------------------------------------------------------------------ fused form
FUSED = __import__("os").environ.get("CG3D_FUSED_ROT_IOU", "1") != "0"


class RotatedIoU3D(Function):
    """cal_iou_3d of aligned pairs as one launch each way (cg3d_rotated_iou3d_{fwd,bwd}, include/cagroup3d_stages.h): the same
    expressions in the same order, the gradient with respect to `pred` written out analytically along the reference's autograd
    graph (the reference: ~80 tensor launches forward, as many again backward, around its one native kernel)."""

    @staticmethod
    def forward(ctx, pred, target):
        from ctypes import c_int64
        lib = _lib.get()
        p, t = pred.contiguous().float(), target.contiguous().float()
        assert p.dim() == 2 and p.shape[1] == 7 and t.shape == p.shape
        lib.check(p, t)
        iou = torch.empty(p.shape[0], dtype=torch.float32, device=p.device)
        lib.call("cg3d_rotated_iou3d_fwd", ptr(p), ptr(t), c_int64(p.shape[0]), ptr(iou), lib.stream())
        ctx.save_for_backward(p, t)
        return iou

    @staticmethod
    def backward(ctx, g):
        from ctypes import c_int64
        p, t = ctx.saved_tensors
        lib = _lib.get()
        g = g.contiguous().float()
        d = torch.empty_like(p)
        lib.call("cg3d_rotated_iou3d_bwd", ptr(p), ptr(t), c_int64(p.shape[0]), ptr(g), ptr(d), lib.stream())
        return d, None


def rotated_iou3d(pred, target):
    """(n,7) x (n,7) -> (n,) rotated 3D IoU, differentiable in `pred`; a target that needs a gradient takes the tensor form."""
    if FUSED and not target.requires_grad and pred.dim() == 2 and pred.shape[1] == 7:
        return RotatedIoU3D.apply(pred, target)
    return cal_iou_3d(pred[None, ...], target[None, ...])[0]
