"""Who launches the ~500 fill kernels of a step?  Python-level torch.zeros / zeros_like / new_zeros / full / ones calls
grouped by call site (dev tool; autograd-internal fills are not seen here)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench
me.PRECISION = 1
bench.PREFETCH = False
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(3):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
cnt = collections.Counter()


def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        f = sys._getframe(1)
        while f is not None and ("/torch/" in f.f_code.co_filename or "tools/" in f.f_code.co_filename):
            f = f.f_back
        site = "%s:%d" % (f.f_code.co_filename.split("repo/")[-1], f.f_lineno) if f else "?"
        cnt[(name, site)] += 1
        return fn(*a, **k)
    setattr(obj, name, w)


for n in ("zeros", "zeros_like", "full", "ones", "ones_like", "full_like", "empty_like"):
    wrap(torch, n)
for n in ("new_zeros", "new_full", "new_ones", "clone", "contiguous"):
    wrap(torch.Tensor, n)
bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
for (n, s), c in cnt.most_common(50):
    print("%5d  %-12s %s" % (c, n, s[:110]))
