"""Knock-out timing of the tile kernel (CG3D_TILE_DBG variants; results wrong on purpose). dev tool, GPU only."""
import sys, os, subprocess
if len(sys.argv) == 1:
    for dbg in (0, 1, 2, 3, 4, 8, 7, 15):
        out = subprocess.run([sys.executable, __file__, str(dbg)], env=dict(os.environ, CG3D_TILE_DBG=str(dbg)), capture_output=True, text=True)
        print("DBG=%-2d (1 no weight loads, 2 conflict-free A reads, 4 no MFMA, 8 no staging)\n%s" % (dbg, out.stdout.strip() or out.stderr[-400:]))
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic
from microbench_conv import timeit
me.PRECISION = 1
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
for ts in (2, 4, 8, 16):
    keys[ts] = mgr.stride(keys[ts // 2], 2)
line = []
for ts, cin, cout in ((2, 64, 64), (4, 128, 128), (8, 256, 256), (16, 512, 512)):
    km = mgr.kernel_map(keys[ts], keys[ts], 3, 1, False)
    P = int((km.nbr >= 0).sum())
    xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
    wf, _ = me._prep_frag(torch.randn(27, cin, cout, device="cuda") * 0.05, True, False)
    plan = me.build_tile_plan(km.nbr, P)
    t = timeit(lambda: me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1), 20, 3)
    line.append("ts%d %d->%d %.1f us" % (ts, cin, cout, t * 1e3))
print("   " + " | ".join(line))
