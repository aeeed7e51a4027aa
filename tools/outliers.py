"""Which phase of a step do the outlier steps (2-4x the median) spend their time in?  Host clock per phase, no device
synchronisation added (dev tool, GPU only)."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
params = [p for p in model.parameters() if p.requires_grad]
prepared = None
rows = []
gcs = []
import threading, traceback
samples, stop = [], [False]
main_id = threading.get_ident()
def sampler():
    while not stop[0]:
        me_id = threading.get_ident()
        for tid, fr in sys._current_frames().items():
            if tid == me_id:
                continue
            st = traceback.extract_stack(fr)[-5:]
            samples.append((time.perf_counter(), ("main " if tid == main_id else "thr  ") + " <- ".join("%s:%d %s" % (os.path.basename(f.filename), f.lineno, f.name) for f in reversed(st))))
        time.sleep(0.004)
threading.Thread(target=sampler, daemon=True).start()
gc.callbacks.append(lambda phase, info: gcs.append((time.perf_counter(), phase, info.get("generation"))))
for i in range(105):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    b = bench.fresh(batch)
    if prepared is not None:
        b["prepared"] = prepared
    ret, tb, disp = model(b)
    t1 = time.perf_counter()
    ret["loss"].backward()
    t2 = time.perf_counter()
    torch.nn.utils.clip_grad_norm_(params, 10.0)
    opt.step()
    t3 = time.perf_counter()
    prepared = model.prefetch_coordinates(batch)
    t4 = time.perf_counter()
    if i >= 5:
        rows.append((t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3, i, t0, t4))
torch.cuda.synchronize()
stop[0] = True
tot = sorted(r[0] for r in rows)
print("host step time: median %.1f mean %.1f p90 %.1f max %.1f ms" % (tot[50] * 1e3, sum(tot) / len(tot) * 1e3, tot[90] * 1e3, tot[-1] * 1e3))
med = [sorted(r[k] for r in rows)[50] * 1e3 for k in range(1, 5)]
print("median phases: forward %.1f backward %.1f clip+opt %.1f prefetch %.1f" % tuple(med))
for r in sorted(rows, reverse=True)[:10]:
    g = [(ph, gen) for (t, ph, gen) in gcs if r[6] <= t <= r[7] and ph == "start"]
    print("step %3d total %.1f: forward %.1f backward %.1f clip+opt %.1f prefetch %.1f   gc starts %s" % (r[5], r[0] * 1e3, r[1] * 1e3, r[2] * 1e3, r[3] * 1e3, r[4] * 1e3, g))

import collections
for r in sorted(rows, reverse=True)[:6]:
    ss = [x[1] for x in samples if r[6] <= x[0] <= r[7]]
    # longest run of identical consecutive samples = where the thread sat
    best, cur, n = None, None, 0
    runs = collections.Counter()
    for x in ss:
        runs[x] += 1
    print("step %d (%.0f ms): top stacks" % (r[5], r[0] * 1e3))
    for k, v in runs.most_common(5):
        print("   %3d samples (~%d ms)  %s" % (v, v * 4, k[:300]))
