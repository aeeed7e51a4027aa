"""Which dense GEMMs / big torch kernels run in a step, with shapes (dev tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cagroup3d_amd import build_model, me
import bench

me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(3):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = sorted(ka, key=lambda e: -e.device_time_total)
print("%-44s %6s %10s  %s" % ("op", "count", "gpu_ms", "shapes"))
for e in rows[:200]:
    print("%-44s %6d %10.3f  %s" % (e.key[:44], e.count, e.device_time_total / 1e3, str(e.input_shapes)[:150]))
