"""Mirror of pcdet/ops/knn/knn.py:16-66 (KNN autograd Function) on the gfx950 C-ABI."""
from ctypes import c_int32

import torch
from torch.autograd import Function

from .. import _lib
from .._lib import ptr


def _scratch(lib, b, n, m, k, device):
    nbytes = int(lib.raw("cg3d_knn_ws_bytes")(b, n, m, k))
    return torch.empty(nbytes // 8, dtype=torch.int64, device=device) if nbytes > 0 else None


class KNN(Function):
    @staticmethod
    def forward(ctx, k, xyz, center_xyz=None, transposed=False):
        """xyz (B,N,3), center_xyz (B,npoint,3) -> idx (B,k,npoint) int32 of the k nearest xyz rows."""
        assert k > 0
        if center_xyz is None:
            center_xyz = xyz
        if transposed:
            xyz = xyz.transpose(2, 1).contiguous()
            center_xyz = center_xyz.transpose(2, 1).contiguous()
        assert xyz.is_contiguous() and center_xyz.is_contiguous()
        xyz, center_xyz = xyz.float(), center_xyz.float()
        assert center_xyz.device == xyz.device, "center_xyz and xyz should be put on the same device"
        lib = _lib.get()
        lib.check(xyz, center_xyz)
        B, npoint, _ = center_xyz.shape
        N = xyz.shape[1]
        idx = torch.zeros((B, npoint, k), dtype=torch.int32, device=xyz.device)
        dist2 = torch.zeros((B, npoint, k), dtype=torch.float32, device=xyz.device)
        lib.call("cg3d_knn", c_int32(B), c_int32(N), c_int32(npoint), c_int32(k), ptr(xyz),
                 ptr(center_xyz), ptr(idx), ptr(dist2), ptr(_scratch(lib, B, xyz.shape[1], npoint, k, xyz.device)), lib.stream())
        idx = idx.transpose(2, 1).contiguous()
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


knn = KNN.apply


def knn_with_dist(k, xyz, center_xyz):
    """(idx [B,npoint,k] int32, dist2 [B,npoint,k] fp32) -- the raw KNN_OP.knn_wrapper outputs (knn.cpp:28-41)."""
    lib = _lib.get()
    xyz, center_xyz = xyz.contiguous().float(), center_xyz.contiguous().float()
    lib.check(xyz, center_xyz)
    B, npoint, _ = center_xyz.shape
    idx = torch.zeros((B, npoint, k), dtype=torch.int32, device=xyz.device)
    dist2 = torch.zeros((B, npoint, k), dtype=torch.float32, device=xyz.device)
    lib.call("cg3d_knn", c_int32(B), c_int32(xyz.shape[1]), c_int32(npoint), c_int32(k), ptr(xyz), ptr(center_xyz),
             ptr(idx), ptr(dist2), ptr(_scratch(lib, B, xyz.shape[1], npoint, k, xyz.device)), lib.stream())
    return idx, dist2
