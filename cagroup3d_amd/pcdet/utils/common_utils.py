"""The two helpers of pcdet/utils/common_utils.py the hot path uses."""
import numpy as np
import torch


def check_numpy_to_torch(x):
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).float(), True
    return x, False


def rotate_points_along_z(points, angle):
    """points (B,N,3+C), angle (B,) -> rotated about +z, x towards y (common_utils.py:35-57)."""
    points, is_np = check_numpy_to_torch(points)
    angle, _ = check_numpy_to_torch(angle)
    c, s = torch.cos(angle), torch.sin(angle)
    z, o = torch.zeros_like(angle), torch.ones_like(angle)
    rot = torch.stack((c, s, z, -s, c, z, z, z, o), dim=1).view(-1, 3, 3).float()
    out = torch.cat((torch.matmul(points[:, :, 0:3], rot), points[:, :, 3:]), dim=-1)
    return out.numpy() if is_np else out
