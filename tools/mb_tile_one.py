"""One layer shape through the tile kernel, many launches (for rocprofv3 --pmc passes). dev tool, GPU only.
usage: python tools/mb_tile_one.py [ts cin cout]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic
from microbench_conv import timeit
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
ts, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 128, 128)
batch = synthetic.make_batch("S50k", 4)
pts = torch.from_numpy(batch["points"]).cuda()
coords = pts[:, :4].clone()
coords[:, 1:] /= 0.02
c = coords.floor().long()
def spread(v):
    v = v & 0x1FFFFF
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v
key = (c[:, 0] << 58) | (spread(c[:, 1] + 2048) << 2) | (spread(c[:, 2] + 2048) << 1) | spread(c[:, 3] + 2048)
order = key.argsort()
coords, pts = coords[order].contiguous(), pts[order].contiguous()
x = me.SparseTensor(coordinates=coords, features=pts[:, 4:] / 255.)
mgr = x.coordinate_manager
keys = {1: x.coordinate_map_key}
for t in (2, 4, 8, 16):
    keys[t] = mgr.stride(keys[t // 2], 2)
km = mgr.kernel_map(keys[ts], keys[ts], 3, 1, False)
P = int((km.nbr >= 0).sum())
xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
wf, _ = me._prep_frag(torch.randn(27, cin, cout, device="cuda") * 0.05, True, False)
plan = me.build_tile_plan(km.nbr, P)
t = timeit(lambda: me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1), 20, 3)
print("ts%d %d->%d rows %d: %.1f us" % (ts, cin, cout, km.n_out, t * 1e3))
