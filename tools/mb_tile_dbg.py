"""Tile kernel on the S50k x 4 layer shapes, one process, library chosen by CG3D_DEV_LIB (dev tool, GPU only).
usage: [CG3D_DEV_LIB=cagroup3d_amd/csrc/dev/libcg3d_dbg3.so] [OUT16=1] [STATS=1] [CHECK=1] python tools/mb_tile_dbg.py [label]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_int32, c_int64
from cagroup3d_amd import _lib
if os.environ.get("CG3D_DEV_LIB"):
    _lib.HIP_LIB_PATH = os.path.abspath(os.environ["CG3D_DEV_LIB"])
from cagroup3d_amd import me, synthetic
from cagroup3d_amd._lib import ptr
from microbench_conv import timeit
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
batch = synthetic.make_batch("S50k", 4)
pts = torch.from_numpy(batch["points"]).cuda()
coords = pts[:, :4].clone()
coords[:, 1:] /= 0.02
x = me.SparseTensor(coordinates=coords, features=pts[:, 4:] / 255.)
mgr = x.coordinate_manager
keys = {1: x.coordinate_map_key}
for t in (2, 4, 8, 16):
    keys[t] = mgr.stride(keys[t // 2], 2)
out = []
check = os.environ.get("CHECK") == "1"
out16 = os.environ.get("OUT16") == "1"
want_stats = os.environ.get("STATS") == "1"
ucap = int(os.environ.get("UCAP", "511"))
lib = _lib.get()
shapes = ((2, 64, 64), (4, 128, 128), (8, 256, 256), (16, 512, 512))
if os.environ.get("SHAPES"):
    shapes = tuple(tuple(int(v) for v in s.split(",")) for s in os.environ["SHAPES"].split(";"))
for ts, cin, cout in shapes:
    km = mgr.kernel_map(keys[ts], keys[ts], 3, 1, False)
    P = int((km.nbr >= 0).sum())
    xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
    w = torch.randn(27, cin, cout, device="cuda") * 0.05
    wf, _ = me._prep_frag(w, True, False)
    plan = me.build_tile_plan(km.nbr, P, ucap=ucap)
    y = torch.empty((plan.n_out, cout), dtype=torch.int16 if out16 else torch.float32, device="cuda")
    stats = torch.zeros(1024 * 2 * cout, device="cuda") if want_stats else None

    def run():
        lib.call("cg3d_spconv_tile_fwd", ptr(xin), ptr(wf), ptr(plan.slots), ptr(plan.live), ptr(plan.pass_tab), ptr(plan.npass),
                 ptr(plan.ulist), c_int32(plan.maxpass), c_int32(plan.ucap), ptr(plan.tiles), c_int64(plan.ntile), ptr(plan.order), ptr(None), ptr(y),
                 c_int64(km.n_in), c_int64(plan.n_out), c_int32(plan.K), c_int32(cin), c_int32(cout), c_int32(1),
                 c_int32(2 if out16 else 0), ptr(stats), lib.stream())
    t = timeit(run, 30, 5)
    s = "ts%d %d->%d %.1f us" % (ts, cin, cout, t * 1e3)
    if check:
        y_old = me._conv_implicit_bf16(xin, me._prep_bf16_t(w), km.nbr, None, km.n_out, cin, cout, P)
        run()
        y_new = y.view(torch.bfloat16).float() if out16 else y
        s += " (relerr %.1e)" % float((y_old - y_new).abs().max() / y_old.abs().max())
    out.append(s)
print("%-14s %s" % (sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("CG3D_DEV_LIB", "default")), " | ".join(out)))
