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


class _PendingBlock:
    """Names + a device tensor with one element per name; `host` is filled by the first DeferredLog that reads it and shared by
    every log that absorbed the same block (tb_dict and disp_dict of one step: one device -> host copy for both)."""
    __slots__ = ("names", "values", "host")

    def __init__(self, names, values):
        self.names, self.values, self.host = tuple(names), values.detach().reshape(-1), None


class DeferredLog(dict):
    """A tb_dict whose numbers are still on the device: the reference reads every loss term back inside the model
    (`.item()` per term, pcdet/models/dense_heads/cagroup_head.py:555, roi_heads/cagroup_roi_head.py) -- a stream
    synchronisation in the middle of the forward pass.  Here the terms stay device tensors until somebody LOOKS at the
    dict (any read access): one device -> host copy for all of them, at the reader's time -- in a training loop that is
    after the optimizer step has been queued, in a loop that logs every n-th iteration not at all in between.

    Every dict method that reads or removes materialises first; a key the user wrote (`log[k] = v`, `update`, `setdefault`)
    before the first read wins over the pending device value of the same name."""

    def __init__(self, names=(), values=None):
        super().__init__()
        self._pend = []
        self._user = set()
        if values is not None:
            self.defer(names, values)

    def defer(self, names, values):
        """names: the keys, values: a device tensor with one element per key."""
        self._pend.append(_PendingBlock(names, values))
        return self

    def absorb(self, other):
        """dict.update that keeps another DeferredLog's pending numbers pending (and shares their one host read)."""
        if isinstance(other, DeferredLog):
            self._pend += other._pend
            for k, v in dict.items(other):
                dict.__setitem__(self, k, v)
            self._user |= other._user
        else:
            self.update(other)
        return self

    def _sync(self):
        if self._pend:
            import torch
            pend, self._pend = self._pend, []
            cold = [b for b in pend if b.host is None]
            if cold:
                flat = torch.cat([b.values.float() for b in cold]).cpu().tolist()      # the one host read
                i = 0
                for b in cold:
                    b.host = flat[i:i + len(b.names)]
                    i += len(b.names)
            for b in pend:
                for n, v in zip(b.names, b.host):
                    if n not in self._user:
                        dict.__setitem__(self, n, v)

    # -- writes: remember what the user set (it must survive a later materialisation)
    def __setitem__(self, k, v):
        self._user.add(k)
        dict.__setitem__(self, k, v)

    def update(self, *a, **kw):
        for k, v in dict(*a, **kw).items():
            self[k] = v

    def setdefault(self, k, default=None):
        self._sync()
        if not dict.__contains__(self, k):
            self[k] = default
        return dict.__getitem__(self, k)

    def __delitem__(self, k):
        self._sync()
        self._user.discard(k)
        dict.__delitem__(self, k)

    def pop(self, k, *default):
        self._sync()
        self._user.discard(k)
        return dict.pop(self, k, *default)

    def popitem(self):
        self._sync()
        k, v = dict.popitem(self)
        self._user.discard(k)
        return k, v

    def clear(self):
        self._pend, self._user = [], set()
        dict.clear(self)

    # -- reads
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

    def __eq__(self, o):
        self._sync()
        if isinstance(o, DeferredLog):
            o._sync()
        return dict.__eq__(self, o)

    def __ne__(self, o):
        return not self.__eq__(o)

    __hash__ = None

    def __or__(self, o):
        out = self.copy()
        out.update(o)
        return out

    def __ror__(self, o):
        out = dict(o)
        out.update(self.copy())
        return out

    def __ior__(self, o):
        self.update(o)
        return self

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
