"""Timing of the output-stationary bf16 conv kernel on the layer shapes of the S50k backbone (dev tool, GPU only).
usage: python tools/mb_implicit.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic
from microbench_conv import timeit

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
batch = synthetic.make_batch("S50k", 4)
pts = torch.from_numpy(batch["points"]).cuda()
coords = pts[:, :4].clone()
coords[:, 1:] /= 0.02
if os.environ.get("SORT") == "morton":
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
tot = 0.0
for ts, cin, cout, n in ((2, 64, 64, 8), (4, 128, 128, 24), (4, 256, 128, 3), (8, 256, 256, 6), (8, 512, 256, 2), (16, 512, 512, 7)):
    km = mgr.kernel_map(keys[ts], keys[ts], 3, 1, False)
    P = int((km.nbr >= 0).sum())
    xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
    wb = me._prep_bf16_t(torch.randn(27, cin, cout, device="cuda") * 0.05)
    t = timeit(lambda: me._conv_implicit_bf16(xin, wb, km.nbr, None, km.n_out, cin, cout, P), 20, 3)
    tot += t * n
    dense = 2.0 * km.n_out * 27 * cin * cout
    print("ts%-2d %4d->%4d rows %7d pairs %8d  %8.1f us  x%-2d  dense %6.1f TF/s  pairs %6.1f TF/s  gather %6.0f GB/s" % (
        ts, cin, cout, km.n_out, P, t * 1e3, n, dense / t / 1e9, 2.0 * P * cin * cout / t / 1e9, 2.0 * P * cin / t / 1e6))
print("weighted sum %.3f ms/step" % tot)
