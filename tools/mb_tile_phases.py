"""Cycles per consumer phase of the tile kernel (CG3D_TILE_DBG=128 build variant with cycle-counter stamps). dev tool, GPU only."""
import sys, os, ctypes
os.environ["CG3D_TILE_DBG"] = os.environ.get("PH_DBG", "128")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic, _lib
from microbench_conv import timeit
me.PRECISION = 1
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
ntile = int(os.environ.get("PH_TILES", "0"))
if ntile:
    tiles = torch.tensor([(0, t * 128, 128) for t in range(ntile)], dtype=torch.int32, device="cuda")
    plan = me.build_tile_plan(km.nbr, P, tiles=(tiles, ntile))
else:
    plan = me.build_tile_plan(km.nbr, P)
lib = _lib.get()
buf = (ctypes.c_ulonglong * 8)()
me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1)
torch.cuda.synchronize()
lib.raw("cg3d_tile_debug_read")(buf, 1)
n = 10
for _ in range(n):
    me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1)
torch.cuda.synchronize()
lib.raw("cg3d_tile_debug_read")(buf, 1)
wgs = buf[7] / n
names = ["barrier wait", "desc read", "compute loop", "all-done wait", "exchange", "store", "compute prologue"]
tot = sum(buf[i] for i in range(7)) / n / wgs
print("ts%d %d->%d: %d tiles, %.0f workgroups; per workgroup (wave 0) %.0f cycles total (s_memtime ticks = 100 MHz? see ratio)" % (ts, cin, cout, plan.ntile, wgs, tot))
for i in range(7):
    print("  %-14s %10.0f ticks per workgroup  %5.1f %%" % (names[i], buf[i] / n / wgs, 100.0 * buf[i] / n / wgs / tot))
t = timeit(lambda: me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1), 20, 3)
print("kernel %.1f us" % (t * 1e3))
