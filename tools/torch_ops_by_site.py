"""Which source lines of the framework issue the torch (ATen) ops of a train step: a TorchDispatchMode counts every op and
charges it to the innermost cagroup3d_amd / bench frame on the Python stack (backward ops run on the autograd thread and are
charged to the autograd node instead).  dev tool; runs on the GPU box (or CPU with the oracle for a rough picture)."""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


VIEWS = ("aten.view.", "aten.detach.", "aten.select.", "aten.slice.", "aten.empty", "aten.alias.", "aten.expand.", "aten.unsqueeze.",
         "aten.t.", "aten.lift_fresh.", "aten.record_stream.", "aten.unbind.", "aten.split", "aten.squeeze.", "aten.permute.",
         "aten.transpose.", "aten._unsafe_view", "aten.reshape", "aten.as_strided", "aten.is_pinned", "aten._local_scalar_dense")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites, self.ops, self.fills, self.gemms = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)[:-1]):
            if fr.filename.startswith(ROOT) and "tools/" not in fr.filename:
                site = "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
                break
        name = str(func)
        if "_local_scalar_dense" in name:
            self.fills[("item", site)] += 1                    # a device -> host read (one copyBuffer + a sync)
        if any(k in name for k in VIEWS):
            return func(*args, **(kwargs or {}))               # no kernel behind these
        if site == "?":
            node = torch._C._current_autograd_node()
            if node is not None:
                site = "autograd: " + node.name()
        self.sites[site] += 1
        self.ops[str(func)] += 1
        if any(k in name for k in ("aten.max.", "aten.min.", "aten.amax.", "aten.amin.", "aten.aminmax.", "aten.sum.", "aten.any.", "aten.all.", "aten.argmax", "aten.argmin", "aten.topk", "aten.sort", "aten.unique", "aten._unique", "aten.nonzero", "aten.cumsum")):
            shp = " ".join(str(tuple(a.shape)) + ("s" if not a.is_contiguous() else "") for a in args if torch.is_tensor(a))
            self.gemms[(name.split(".")[1] + "." + name.split(".")[2], shp + " " + " ".join(str(a) for a in args[1:] if isinstance(a, (int, bool))), site)] += 1
        if any(k in name for k in ("aten.mm.", "aten.addmm.", "aten.bmm.", "aten.baddbmm.", "aten.matmul.")):
            shp = " x ".join(str(tuple(a.shape)) for a in args if torch.is_tensor(a))
            self.gemms[(name.split(".")[1], shp, site)] += 1
        if any(k in name for k in ("_to_copy", "zeros", "fill_", "copy_", "clone", "zero_", "ones", "full")):
            self.fills[(name.split(".")[1], site)] += 1
        return func(*args, **(kwargs or {}))


me.PRECISION = 1


me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device=dev)
for _ in range(3):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
c = Count()
from cagroup3d_amd import _lib  # noqa: E402
entries = collections.Counter()
_call = _lib.Library.call


def counted(self, name, *a):
    entries[name] += 1
    return _call(self, name, *a)


_lib.Library.call = counted
with c:
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
_lib.Library.call = _call
print("C-ABI calls in one step:", sum(entries.values()))
for s, n in entries.most_common(60):
    print("%5d  %s" % (n, s))
print("ops in one step:", sum(c.sites.values()))
for s, n in c.sites.most_common(90):
    print("%5d  %s" % (n, s))
print("--- fills / copies / casts by site")
for (op, site), n in c.fills.most_common(90):
    print("%5d  %-10s %s" % (n, op, site))
print("--- library GEMMs, reductions, sorts by site")
for (op, shp, site), n in c.gemms.most_common(150):
    print("%5d  %-16s %-44s %s" % (n, op, shp, site))
print("--- by op")
for s, n in c.ops.most_common(40):
    print("%5d  %s" % (n, s))
