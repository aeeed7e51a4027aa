"""Per-workgroup trace of the tile kernel (dev build: CG3D_HIPCC_EXTRA=-DCG3D_TILE_TRACE python cagroup3d_amd/csrc/build.py
--force): shader clock vs the 100 MHz counter (effective clock), which workgroups shared a CU and when.  GPU only."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cagroup3d_amd import me, build_model, _lib

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
ts, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 128, 128)
nunits = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pts = build_model.synthetic_batch("S50k", 4, device="cuda")["points"]
coords = pts[:, :4].clone()
coords[:, 1:] /= 0.02
x = me.SparseTensor(coordinates=coords, features=pts[:, 4:] / 255.)
mgr = x.coordinate_manager
keys = {1: x.coordinate_map_key}
for t in (2, 4, 8, 16):
    keys[t] = mgr.stride(keys[t // 2], 2)
km = mgr.kernel_map(keys[ts], keys[ts], 3, 1, False)
P = int((km.nbr >= 0).sum())
xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
wf, _ = me._prep_frag(torch.randn(27, cin, cout, device="cuda") * 0.05, True, False)
ny = max(cout // 128, 1)
if nunits:
    nt = nunits // ny
    tiles = torch.tensor([(0, t * 128, 128) for t in range(nt)], dtype=torch.int32, device="cuda")
    plan = me.build_tile_plan(km.nbr, P, tiles=(tiles, nt))
else:
    plan = me.build_tile_plan(km.nbr, P)
nwg = (plan.ntile * ny + 7) // 8 * 8
buf = torch.zeros(nwg * 16, dtype=torch.int64, device="cuda")
lib = _lib.get()
assert lib.raw("cg3d_tile2_trace_set")(ctypes.c_void_p(buf.data_ptr())) == 0
for _ in range(3):
    me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1)
torch.cuda.synchronize()
buf.zero_()
me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(nwg, 16).astype(np.uint64)
bidx = np.nonzero(t[:, 1] > 0)[0]                     # row of the trace buffer = blockIdx.x
t = t[t[:, 1] > 0]
c0, c1, r0, r1 = (t[:, i].astype(np.float64) for i in range(4))
hw = t[:, 4]
cu = (hw >> np.uint64(8)) & np.uint64(15); sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
xcc = (hw >> np.uint64(32)) & np.uint64(15)
simd = (hw >> np.uint64(4)) & np.uint64(3)
wall_us = (r1.max() - r0.min()) / 100.0
print("ts%d %d->%d: %d workgroups, kernel span %.1f us (100 MHz counter)" % (ts, cin, cout, len(t), wall_us))
dur_us = (r1 - r0) / 100.0
clk = (c1 - c0) / np.maximum(dur_us, 1e-3) / 1e3
print("workgroup duration: mean %.1f us, min %.1f, max %.1f;  shader cycles / us over a workgroup's life: mean %.2f GHz (min %.2f, max %.2f)"
      % (dur_us.mean(), dur_us.min(), dur_us.max(), clk.mean(), clk.min(), clk.max()))
cuid = (xcc.astype(np.int64) * 64 + se.astype(np.int64) * 16 + sh.astype(np.int64) * 16 * 8 + cu.astype(np.int64))
ids, counts = np.unique(cuid, return_counts=True)
print("distinct CUs seen: %d; workgroups per CU: min %d max %d" % (len(ids), counts.min(), counts.max()))
# which blockIdx values share a CU in the first round (the hardware's placement, not ours)
first = (r0 - r0.min()) / 100.0 < 5.0
diffs = []
for c in ids:
    b = np.sort(bidx[(cuid == c) & first])
    if len(b) == 2:
        diffs.append(int(b[1] - b[0]))
if diffs:
    vals, cnt = np.unique(np.asarray(diffs), return_counts=True)
    top = np.argsort(-cnt)[:6]
    print("first-round pairs on one CU: blockIdx difference -> count:", ", ".join("%d: %d" % (vals[i], cnt[i]) for i in top))
# concurrency: for every CU, the time during which >= 2 of its workgroups overlap
ov = tot = 0.0
for c in ids:
    m = cuid == c
    ev = sorted([(a, 1) for a in r0[m]] + [(b, -1) for b in r1[m]])
    n = 0; last = ev[0][0]
    for tt, d in ev:
        if n >= 1: tot += tt - last
        if n >= 2: ov += tt - last
        n += d; last = tt
print("CU-busy time %.0f us summed over CUs, of which %.0f us (%.0f %%) with two workgroups resident" % (tot / 100, ov / 100, 100 * ov / max(tot, 1)))
start_us = (r0 - r0.min()) / 100.0
order = np.argsort(start_us)
print("start times (us) of workgroups by quantile:", " ".join("%.1f" % np.quantile(start_us, q) for q in (0, .25, .5, .75, .9, 1)))
print("end times   (us):", " ".join("%.1f" % np.quantile((r1 - r0.min()) / 100.0, q) for q in (0, .25, .5, .75, .9, 1)))

ph = t[:, 6:14].astype(np.float64)
names = ["start -> staged 0", "multiply 0", "-> staged 1", "multiply 1", "-> all waves done"]
prev = r0
print("phases of a workgroup (wave 0, 100 MHz counter), mean us over workgroups [first-quartile starters | last-quartile starters]:")
q1 = start_us <= np.quantile(start_us, .25); q4 = start_us >= np.quantile(start_us, .75)
for i, nm in enumerate(names):
    cur = ph[:, i]
    ok = cur > 0
    d = (cur - prev) / 100.0
    print("  %-20s %6.2f   [%6.2f | %6.2f]" % (nm, d[ok].mean(), d[ok & q1].mean() if (ok & q1).any() else float("nan"), d[ok & q4].mean() if (ok & q4).any() else float("nan")))
    prev = cur
d = (r1 - prev) / 100.0
print("  %-20s %6.2f" % ("exchange + store", d.mean()))
