"""Time per step of the tile kernel = slope over the number of live offsets processed (CG3D_TILE_MAXK), per knock-out variant,
on exactly NT tiles (one round of workgroups).  dev tool, GPU only.   usage: python tools/mb_tile_slope.py"""
import sys, os, subprocess
if len(sys.argv) == 1:
    for ntile, pad in ((256, 0), (512, 0)):
        for dbg in (16, 24):
            ts = []
            for mk in (3, 27):
                out = subprocess.run([sys.executable, __file__, str(ntile)], capture_output=True, text=True,
                                     env=dict(os.environ, CG3D_TILE_DBG=str(dbg), CG3D_TILE_MAXK=str(mk), CG3D_TILE_LDSPAD=str(pad)))
                ts.append(float(out.stdout.strip().split()[-1]) if out.stdout.strip() else float("nan"))
            print("tiles %d (%d) DBG %2d: maxk 3 -> %6.1f us, maxk 27 -> %6.1f us: %5.2f us per offset (2 steps), intercept %5.1f us" % (
                ntile, 1 if pad else 2, dbg - 16, ts[0], ts[1], (ts[1] - ts[0]) / 24, ts[0] - 3 * (ts[1] - ts[0]) / 24), flush=True)
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic
from microbench_conv import timeit
me.PRECISION = 1
ntile = int(sys.argv[1])
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
for t in (2, 4):
    keys[t] = mgr.stride(keys[t // 2], 2)
km = mgr.kernel_map(keys[4], keys[4], 3, 1, False)
P = int((km.nbr >= 0).sum())
cin = cout = 128
xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
wf, _ = me._prep_frag(torch.randn(27, cin, cout, device="cuda") * 0.05, True, False)
tiles = torch.tensor([(0, t * 128, 128) for t in range(ntile)], dtype=torch.int32, device="cuda")
plan = me.build_tile_plan(km.nbr, P, tiles=(tiles, ntile))
t = timeit(lambda: me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1), 20, 3)
print("%.2f" % (t * 1e3))
