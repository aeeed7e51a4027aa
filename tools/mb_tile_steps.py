"""Per-step time stamps of workgroup 0 / wave 0 over its first 8 stages (CG3D_TILE_DBG=256). dev tool, GPU only."""
import sys, os, ctypes
os.environ["CG3D_TILE_DBG"] = os.environ.get("ST_DBG", "256")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic, _lib
me.PRECISION = 1
ts, cin, cout = 4, 128, 128
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
km = mgr.kernel_map(keys[ts], keys[ts], 3, 1, False)
P = int((km.nbr >= 0).sum())
xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
wf, _ = me._prep_frag(torch.randn(27, cin, cout, device="cuda") * 0.05, True, False)
plan = me.build_tile_plan(km.nbr, P)
lib = _lib.get()
for _ in range(3):
    me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
lib.raw("cg3d_tile_debug_steps")(buf)
t0 = buf[0]
for s in range(8):
    row = [buf[s * 16 + i] for i in range(16)]
    if row[0] == 0:
        continue
    d = [row[i + 1] - row[i] for i in range(14) if row[i + 1] > 0] 
    if s not in (2, 5): continue
    print("stage %d: start %7d, step ticks %s, end-start %d" % (s, row[0] - t0, " ".join("%5d" % v for v in d), row[15] - row[0]))
