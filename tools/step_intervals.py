"""Per-step durations measured WITHOUT synchronising: one event per step on the main stream, intervals read at the end
(dev tool, GPU only).  Shows whether run-to-run noise of bench.py comes from a few outlier steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench
me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(5):
    bench.train_step(model, opt, batch, 10)
import gc
if os.environ.get("GC_FREEZE") == "1":
    gc.collect(); gc.freeze()
if os.environ.get("GC_FREEZE") == "2":
    gc.collect(); gc.freeze(); gc.disable()
for rep in range(3):
    evs = []
    for i in range(41):
        bench.train_step(model, opt, batch, 10)
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    torch.cuda.synchronize()
    d = [evs[i].elapsed_time(evs[i + 1]) for i in range(40)]
    s = sorted(d)
    print("gc", gc.get_count(), gc.get_stats()[2]["collections"], "reserved %.1f GB segments %d" % (torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_stats()["segment.all.allocated"]))
    print("rep %d mean %.1f median %.1f min %.1f p90 %.1f max %.1f | %s" % (rep, sum(d) / 40, s[20], s[0], s[36], s[-1], " ".join("%.0f" % x for x in d)))
