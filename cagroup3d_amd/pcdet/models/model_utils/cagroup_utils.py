"""Mirror of pcdet/models/model_utils/cagroup_utils.py (196 lines in the reference)."""
import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn


def reduce_mean(tensor):
    """Mean over ranks (cagroup_utils.py:6-12): all-reduce(SUM) of tensor / world_size over RCCL."""
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    t = tensor.clone().div_(dist.get_world_size())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def parse_params(param_cfg):
    """{'NAME': .., 'LOSS_WEIGHT': 1} -> {'loss_weight': 1} (cagroup_utils.py:14-25)."""
    return {k.lower(): v for k, v in param_cfg.items() if k.lower() != "name"}


def rotation_3d_in_axis(points, angles, axis=0):
    """points (N,M,3), angles (N,) -> rotated points (cagroup_utils.py:27-67)."""
    s, c = torch.sin(angles), torch.cos(angles)
    one, zero = torch.ones_like(c), torch.zeros_like(c)
    if axis == 1:
        rows = [[c, zero, -s], [zero, one, zero], [s, zero, c]]
    elif axis in (2, -1):
        rows = [[c, -s, zero], [s, c, zero], [zero, zero, one]]
    elif axis == 0:
        rows = [[zero, c, -s], [zero, s, c], [one, zero, zero]]
    else:
        raise ValueError("axis should in range [0, 1, 2], got %s" % axis)
    rot_t = torch.stack([torch.stack(r) for r in rows])
    return torch.einsum("aij,jka->aik", (points, rot_t))


class Scale(nn.Module):
    """Learnable scalar multiplier (cagroup_utils.py:69-84)."""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))


class CAGroupResidualCoder(object):
    """Box residual coder (cagroup_utils.py:91-196): centre offsets / BEV diagonal, log size ratios,
    heading difference or (cos, sin) of the heading."""

    def __init__(self, code_size=6, encode_angle_by_sincos=False, **kwargs):
        self.code_size = code_size + (1 if encode_angle_by_sincos else 0)
        self.encode_angle_by_sincos = encode_angle_by_sincos

    def encode_torch(self, boxes, anchors):
        anchors[:, 3:6] = torch.clamp_min(anchors[:, 3:6], min=1e-5)   # in place, as the reference
        boxes[:, 3:6] = torch.clamp_min(boxes[:, 3:6], min=1e-5)
        a = torch.split(anchors, 1, dim=-1)
        g = torch.split(boxes, 1, dim=-1)
        diag = torch.sqrt(a[3] ** 2 + a[4] ** 2)
        out = [(g[0] - a[0]) / diag, (g[1] - a[1]) / diag, (g[2] - a[2]) / a[5],
               torch.log(g[3] / a[3]), torch.log(g[4] / a[4]), torch.log(g[5] / a[5])]
        if self.code_size > 6:
            out += [torch.cos(g[6]), torch.sin(g[6])] if self.encode_angle_by_sincos else [g[6] - a[6]]
            out += [gg - aa for gg, aa in zip(g[7:], a[7:])]
        return torch.cat(out, dim=-1)

    def decode_torch(self, box_encodings, anchors):
        a = torch.split(anchors, 1, dim=-1)
        t = torch.split(box_encodings, 1, dim=-1)
        diag = torch.sqrt(a[3] ** 2 + a[4] ** 2)
        out = [t[0] * diag + a[0], t[1] * diag + a[1], t[2] * a[5] + a[2],
               torch.exp(t[3]) * a[3], torch.exp(t[4]) * a[4], torch.exp(t[5]) * a[5]]
        if self.code_size > 6:
            if self.encode_angle_by_sincos:
                out.append(torch.atan2(t[7], t[6]) + a[6])
                rest_t = t[8:]
            else:
                out.append(t[6] + a[6])
                rest_t = t[7:]
            out += [tt + aa for tt, aa in zip(rest_t, a[7:])]
        return torch.cat(out, dim=-1)
