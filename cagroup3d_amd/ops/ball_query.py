"""Mirror of `ball_query` in pcdet/ops/pointnet2/pointnet2_batch/pointnet2_utils.py:205-228 (BallQuery autograd Function
over `pointnet2_batch_cuda.ball_query_wrapper`) on the gfx950 C-ABI (`cg3d_ball_query`, csrc/knn.hip)."""
from ctypes import c_float, c_int32

import torch
from torch.autograd import Function

from .. import _lib
from .._lib import ptr


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3) reference points, new_xyz (B,npoint,3) ball centres -> idx (B,npoint,nsample) int32: the first
        `nsample` reference rows inside each ball (ascending index), unfilled slots repeat the first hit."""
        assert new_xyz.is_contiguous() and xyz.is_contiguous()
        lib = _lib.get()
        xyz, new_xyz = xyz.float(), new_xyz.float()
        lib.check(xyz, new_xyz)
        B, N, _ = xyz.shape
        npoint = new_xyz.shape[1]
        idx = torch.zeros((B, npoint, nsample), dtype=torch.int32, device=xyz.device)
        lib.call("cg3d_ball_query", c_int32(B), c_int32(N), c_int32(npoint), c_float(radius), c_int32(nsample), ptr(new_xyz),
                 ptr(xyz), ptr(idx), lib.stream())
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, a=None):
        return None, None, None, None


ball_query = BallQuery.apply
