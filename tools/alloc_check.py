"""Does a steady-state training step still call hipMalloc / hit allocator retries?  (dev tool, GPU only)"""
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
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "reserved_bytes.all.current",
        "allocated_bytes.all.peak", "segment.all.current")
prev = None
for i in range(6):
    t0 = time.perf_counter()
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats()
    cur = {k: st.get(k, 0) for k in keys}
    print("step %d %.1f ms " % (i, dt) + " ".join("%s=%s" % (k.split(".")[0][:18], cur[k] - (prev[k] if prev and "bytes" not in k and "segment" not in k else 0)) for k in keys))
    prev = cur
