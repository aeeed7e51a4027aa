"""Tile kernel time against the number of units in flight (dev tool, GPU only): the first N tiles of one S50k layer.
With two workgroups per CU, N = 256 units is one per CU, 512 two per CU: T(512) / T(256) says how much of a second
co-resident workgroup's work is hidden behind the first.   usage: python tools/mb_tile_scale.py [ts cin cout]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, build_model
from microbench_conv import timeit

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
ts, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 128, 128)
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
ntile_all = -(-km.n_out // 128)
print("ts%d %d->%d: %d tiles x %d channel blocks" % (ts, cin, cout, ntile_all, ny))
for units in (64, 128, 256, 384, 512, 640, 768, 1024, 1536, 2048):
    nt = units // ny
    if nt > ntile_all - 1:
        break
    tiles = torch.tensor([(0, t * 128, 128) for t in range(nt)], dtype=torch.int32, device="cuda")
    plan = me.build_tile_plan(km.nbr, P, tiles=(tiles, nt))
    t = timeit(lambda: me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1), 20, 3)
    print("  %5d units: %7.1f us   (%.3f us per unit per CU-slot pair)" % (nt * ny, t * 1e3, t * 1e3 / max(nt * ny / 256.0, 1.0)))
plan = me.build_tile_plan(km.nbr, P)
t = timeit(lambda: me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1), 20, 3)
print("  all %d units: %.1f us" % (ntile_all * ny, t * 1e3))
