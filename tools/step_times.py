"""Per-step wall times (synchronised) to spot outliers (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench
me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
ts = []
for i in range(24):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join("%.0f" % t for t in ts))
import gc
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold())
gc.disable()
ts = []
for i in range(16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("gc off:", " ".join("%.0f" % t for t in ts))
