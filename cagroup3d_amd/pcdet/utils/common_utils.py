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


class DeferredLog(dict):
    """A tb_dict whose numbers are still on the device: the reference reads every loss term back inside the model
    (`.item()` per term, pcdet/models/dense_heads/cagroup_head.py:555, roi_heads/cagroup_roi_head.py) -- a stream
    synchronisation in the middle of the forward pass.  Here the terms stay device tensors until somebody LOOKS at the
    dict (any read access): one device -> host copy for all of them, at the reader's time -- in a training loop that is
    after the optimizer step has been queued, in a loop that logs every n-th iteration not at all in between."""

    def __init__(self, names=(), values=None):
        super().__init__()
        self._pend = []
        if values is not None:
            self.defer(names, values)

    def defer(self, names, values):
        """names: the keys, values: a device tensor with one element per key."""
        self._pend.append((tuple(names), values.detach().reshape(-1)))
        return self

    def absorb(self, other):
        """dict.update that keeps another DeferredLog's pending numbers pending."""
        if isinstance(other, DeferredLog):
            self._pend += other._pend
            dict.update(self, dict.items(other))
        else:
            dict.update(self, other)
        return self

    def _sync(self):
        if self._pend:
            import torch
            pend, self._pend = self._pend, []
            flat = torch.cat([v.float() for _, v in pend]).cpu().tolist()      # the one host read
            i = 0
            for names, _ in pend:
                for n in names:
                    dict.__setitem__(self, n, flat[i])
                    i += 1

    def __getitem__(self, k):
        self._sync()
        return dict.__getitem__(self, k)

    def __iter__(self):
        self._sync()
        return dict.__iter__(self)

    def __len__(self):
        self._sync()
        return dict.__len__(self)

    def __contains__(self, k):
        self._sync()
        return dict.__contains__(self, k)

    def __repr__(self):
        self._sync()
        return dict.__repr__(self)

    def keys(self):
        self._sync()
        return dict.keys(self)

    def values(self):
        self._sync()
        return dict.values(self)

    def items(self):
        self._sync()
        return dict.items(self)

    def get(self, k, default=None):
        self._sync()
        return dict.get(self, k, default)

    def copy(self):
        self._sync()
        return dict(dict.items(self))
