"""Which Python frames issue the small fill / copy ops of a train step, over ALL threads (main, autograd, prefetch worker):
torch.profiler with stacks, grouped by (op, innermost frames).  dev tool; GPU box."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402
from cagroup3d_amd.optim import ClippedAdamW  # noqa: E402

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = ClippedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
batch = build_model.synthetic_batch("S50k", 4, device=dev)
for _ in range(4):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(2):
        bench.train_step(model, opt, batch, 10.0)
    torch.cuda.synchronize()
bench.finish_prefetch(model)
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::_local_scalar_dense", "aten::add_", "aten::add", "hipMemsetAsync", "hipMemcpyAsync",
        "hipMemcpyWithStream", "aten::index", "aten::index_put_", "aten::cat", "aten::arange")
cnt = collections.Counter()
for e in prof.events():
    if e.name in want:
        st = [f for f in (e.stack or []) if "cagroup3d_amd" in f or "bench.py" in f]
        cnt[(e.name, " <- ".join(s.split("/")[-1] for s in st[:2]) or "(no python frame: autograd engine / C library)")] += 1
for (name, st), n in cnt.most_common(120):
    print("%6.1f  %-26s %s" % (n / 2, name, st))
