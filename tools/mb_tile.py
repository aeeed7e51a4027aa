"""LDS-staged tile kernel (cg3d_spconv_tile_fwd) vs the direct-operand kernel it replaces, on the S50k layer shapes
(dev tool, GPU only).   usage: [SORT=morton] [UCAP=511] python tools/mb_tile.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic
from microbench_conv import timeit

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
cfg = os.environ.get("CFG", "S50k")
batch = synthetic.make_batch(cfg, 4)
pts = torch.from_numpy(batch["points"]).cuda()
coords = pts[:, :4].clone()
coords[:, 1:] /= (0.01 if cfg == "S200k" else 0.02)
if os.environ.get("SORT", "morton") == "morton":
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
ucap = int(os.environ.get("UCAP", "511"))
tot_old = tot_new = 0.0
shapes = ((1, 1, 64, 64, 3), (1, 2, 64, 64, 3), (2, 2, 64, 64, 8), (2, 4, 64, 128, 1), (4, 4, 128, 128, 24), (4, 4, 256, 128, 3), (4, 8, 128, 256, 1),
          (8, 8, 256, 256, 6), (8, 8, 512, 256, 2), (8, 16, 256, 512, 1), (16, 16, 512, 512, 7))
for tin, tout, cin, cout, n in shapes:
    km = mgr.kernel_map(keys[tin], keys[tout], 3, 1, False)
    P = int((km.nbr >= 0).sum())
    xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
    w = torch.randn(27, cin, cout, device="cuda") * 0.05
    wb = me._prep_bf16_t(w)
    wf, _ = me._prep_frag(w, True, False)
    t_plan = timeit(lambda: me.build_tile_plan(km.nbr, P, ucap=ucap), 5, 1)
    plan = me.build_tile_plan(km.nbr, P, ucap=ucap)
    npass = plan.npass.float()
    ucnt = plan.pass_tab[:, 0, 3].float()
    t_old = timeit(lambda: me._conv_implicit_bf16(xin, wb, km.nbr, None, km.n_out, cin, cout, P), 20, 3)
    res = []
    for ks in (1, 2, 4):
        res.append(timeit(lambda: me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, ks), 20, 3))
    y_old = me._conv_implicit_bf16(xin, wb, km.nbr, None, km.n_out, cin, cout, P)
    y_new = me._conv_tile(xin, wf, plan, None, cin, cout, km.n_in, P, 1)
    err = float((y_old - y_new).abs().max() / y_old.abs().max())
    best = min(res)
    tot_old += t_old * n
    tot_new += best * n
    flops = 2.0 * P * cin * cout
    byts = 2.0 * km.n_in * cin + 4.0 * km.n_out * cout + 2.0 * 27 * cin * cout + 2.0 * 27 * km.n_out
    bound = max(flops / 2.5e15, byts / 8e12) * 1e3
    print("ts%-2d->%-2d %4d->%4d rows %7d pairs %8d | old %7.1f us | tile ks1 %7.1f ks2 %7.1f ks4 %7.1f us | x%-2d | 8(d) bound %5.1f us -> %4.1f%% | "
          "passes avg %.2f max %d, rows/pass0 avg %.0f max %d | plan %.0f us | relerr %.1e" % (
              tin, tout, cin, cout, km.n_out, P, t_old * 1e3, res[0] * 1e3, res[1] * 1e3, res[2] * 1e3, n, bound * 1e3, 100 * bound / best,
              float(npass.mean()), int(npass.max()), float(ucnt.mean()), int(ucnt.max()), t_plan * 1e3, err))
print("weighted sum old %.3f ms/step, tile %.3f ms/step" % (tot_old, tot_new))
