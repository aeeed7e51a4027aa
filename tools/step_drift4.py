"""Is the long-run drift a change of the WORKLOAD (the net trains on the fixed synthetic batch: votes move, class maps
grow)?  Total conv pairs / rows per step from the kernel profile (dev tool, GPU only)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
lr = float(os.environ.get("LR", cfg.OPTIMIZATION.LR))
opt = torch.optim.AdamW(model.parameters(), lr=lr, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY, fused=True)
print("lr", lr)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(5):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
me.KernelProfile.reset(); me.KernelProfile.enabled = True
marks, evs, losses = [], [], []
for i in range(80):
    tb = bench.train_step(model, opt, batch, 10)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    marks.append(len(me.KernelProfile.records)); losses.append(tb.get("loss_all"))
torch.cuda.synchronize()
recs = me.KernelProfile.records
prev, per = 0, []
for m in marks:
    rr = recs[prev:m]; prev = m
    per.append((len(rr), sum(r[4][4] for r in rr), sum(r[4][5] for r in rr), sum(r[0].elapsed_time(r[1]) for r in rr)))
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(79)]
for k in range(0, 80, 10):
    p = per[k:k + 10]
    print("steps %2d-%2d: %.1f ms/step | conv launches %.0f  pairs %.2f M  output rows %.2f M  conv time %.2f ms | loss %s" % (
        k, k + 9, sum(d[k:k + 10]) / len(d[k:k + 10]), sum(x[0] for x in p) / 10, sum(x[1] for x in p) / 1e7, sum(x[2] for x in p) / 1e7,
        sum(x[3] for x in p) / 10, "%.3f" % losses[k + 9] if losses[k + 9] is not None else None))
import collections
def agg(lo, hi):
    a = collections.Counter()
    prev = marks[lo - 1] if lo else 0
    for r in recs[prev:marks[hi - 1]]:
        m = r[4]
        a[(m[0], m[1], m[2], m[3])] += m[4] / (hi - lo)
    return a
a0, a1 = agg(0, 10), agg(70, 80)
print("conv pairs per step by (kind, K, cin, cout): early -> late")
for k in sorted(set(a0) | set(a1), key=lambda k: -(a1[k] - a0[k]))[:12]:
    print("  %-40s %10.2f M -> %10.2f M" % (k, a0[k] / 1e6, a1[k] / 1e6))
