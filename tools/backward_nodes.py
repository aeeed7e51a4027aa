"""Host time of the backward pass by autograd node (torch.profiler, CPU activity; the evaluate_function events).  dev tool; GPU box."""
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
opt = ClippedAdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
batch = build_model.synthetic_batch("S50k", int(os.environ.get("BATCH", "4")), device=dev)
for _ in range(6):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N):
        bench.train_step(model, opt, batch, 10.0)
    torch.cuda.synchronize()
bench.finish_prefetch(model)
acc = collections.Counter()
cnt = collections.Counter()
for e in prof.events():
    if e.name.startswith("autograd::engine::evaluate_function: "):
        k = e.name.split(": ", 1)[1]
        acc[k] += e.cpu_time_total
        cnt[k] += 1
tot = sum(acc.values())
print("backward nodes: %.2f ms per step in %d nodes (under the profiler)" % (tot / N / 1e3, sum(cnt.values()) / N))
for k, v in acc.most_common(40):
    print("  %8.1f us  x%-3d %s" % (v / N, cnt[k] // N, k))
