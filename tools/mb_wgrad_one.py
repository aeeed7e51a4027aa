"""One layer shape through the bf16 weight-gradient kernel, many launches (for rocprofv3 --pmc passes). dev tool, GPU only.
usage: python tools/mb_wgrad_one.py [ts cin cout]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ctypes import c_int32, c_int64
import torch
from cagroup3d_amd import _lib
if os.environ.get("CG3D_DEV_LIB"):
    _lib.HIP_LIB_PATH = os.path.abspath(os.environ["CG3D_DEV_LIB"])
from cagroup3d_amd import me, synthetic
from cagroup3d_amd._lib import ptr
from microbench_conv import timeit
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
ts, cin, cout = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 128, 128)
batch = synthetic.make_batch("S50k", 4)
pts = torch.from_numpy(batch["points"]).cuda()
coords = pts[:, :4].clone()
coords[:, 1:] /= 0.02
x = me.SparseTensor(coordinates=coords, features=pts[:, 4:] / 255.)
mgr = x.coordinate_manager
keys = {1: x.coordinate_map_key}
for t in (2, 4, 8, 16):
    keys[t] = mgr.stride(keys[t // 2], 2)
km = mgr.kernel_map(keys[ts], keys[ts], 3, 1, False)
pin, pout, _, P = km.pairs(None)
xb = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
dyb = me._to_bf16(torch.randn(km.n_out, cout, device="cuda"))
dw = torch.empty(27, cin, cout, device="cuda")
seg, nseg = km.wgrad_segments(me._wgrad_seg_len(P, cin, cout, 1, 27), None)
lib = _lib.get()
f = lambda: lib.call("cg3d_spconv_pairs_wgrad", ptr(xb), ptr(dyb), ptr(pin), ptr(pout), ptr(seg), c_int64(nseg), ptr(dw),
                     c_int32(27), c_int32(cin), c_int32(cout), c_int32(2), lib.stream())
t = timeit(f, 20, 3)
print("wgrad ts%d %d->%d rows %d pairs %d nseg %d: %.1f us  (%.0f GB/s of gathered rows)" % (ts, cin, cout, km.n_out, P, nseg, t * 1e3, P * 2.0 * (cin + cout) / t / 1e6))
