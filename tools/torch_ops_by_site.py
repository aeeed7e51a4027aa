"""Where the framework-op (torch) GPU time of a train step comes from: torch.profiler with stacks, device time of every
non-cg3d kernel attributed to the innermost cagroup3d_amd / bench source line that launched it (dev tool, GPU box)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402

me.PRECISION = 1
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device=dev)
for _ in range(4):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
STEPS = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(STEPS):
        bench.train_step(model, opt, batch, 10.0)
    torch.cuda.synchronize()
site_t, site_n, op_t = collections.Counter(), collections.Counter(), collections.Counter()
for ev in prof.events():
    if ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    dt = ev.self_device_time_total
    if dt <= 0:
        continue
    site = "?"
    for fr in ev.stack or []:
        if ("cagroup3d_amd" in fr or "bench.py" in fr) and "torch/" not in fr:
            site = fr.split("/repo/")[-1]
            break
    site_t[site] += dt
    site_n[site] += 1
    op_t[ev.name] += dt
tot = sum(site_t.values())
print(f"torch-op device time {tot / STEPS / 1e3:.2f} ms/step over {sum(site_n.values()) / STEPS:.0f} ops/step")
for s, t in site_t.most_common(60):
    print(f"{t / STEPS:9.1f} us {site_n[s] / STEPS:6.1f}  {s}")
print("--- by op")
for s, t in op_t.most_common(30):
    print(f"{t / STEPS:9.1f} us  {s}")
