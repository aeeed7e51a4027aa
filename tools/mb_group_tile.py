"""Class-branch convolutions (grouped, K = 729 / 125, 64 -> 64) on the LDS-staged tile kernel vs the dense-map kernel:
forward, data gradient and the plan build, on a synthetic 18-group surface map of the S50k class-map size."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cagroup3d_amd import _lib, me  # noqa: E402
from util import surface_coords  # noqa: E402


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def blob_coords(G, per_group, fill, seed):
    """Voted class maps are volumetric blobs, not surfaces: per group a few boxes filled to `fill`."""
    import numpy as np
    rng = np.random.RandomState(seed)
    out = []
    for g in range(G):
        pts = []
        for _ in range(3):
            side = int(round((per_group / 3 / fill) ** (1 / 3)))
            o = rng.randint(0, 40, 3)
            grid = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3) + o
            pts.append(grid[rng.rand(len(grid)) < fill])
        p = np.unique(np.concatenate(pts), axis=0)
        out.append(np.c_[np.full(len(p), g), p])
    return torch.from_numpy(np.concatenate(out)).float()


def main():
    me.PRECISION = 1
    G = 18
    for ks, per_group, fill in ((9, 1550, 0.45), (5, 150, 0.6)):
        coords = blob_coords(G, per_group, fill, ks).cuda()
        x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1, device="cuda"))
        mgr = x.coordinate_manager
        km = mgr.kernel_map(x.coordinate_map_key, x.coordinate_map_key, ks, 1, False)
        b = torch.bincount(x.C[:, 0].long(), minlength=G).cpu().numpy()
        bounds = (0,) + tuple(int(v) for v in b.cumsum())
        K, cin, cout = ks ** 3, 64, 64
        ws = [torch.randn(K, cin, cout, device="cuda") / (cin * 27) ** 0.5 for _ in range(G)]
        f = torch.randn(km.n_in, cin, device="cuda")
        dy = torch.randn(km.n_out, cout, device="cuda")
        _, _, _, P = km.pairs(bounds)
        x16, dy16 = me._to_bf16(f), me._to_bf16(dy)
        tiles = km.tiles(bounds)
        print(f"K={K} rows={km.n_out} pairs={P} occupancy={P / K / km.n_out:.3f} tiles={tiles[1]}", flush=True)
        # dense-map kernel
        wt, wp = me._prep_bf16_group(ws, True), me._prep_bf16_group(ws, False)
        t_old_f = timeit(lambda: me._conv_implicit_bf16(x16, wt, km.nbr, None, km.n_out, cin, cout, P, tiles))
        t_old_b = timeit(lambda: me._conv_implicit_bf16(dy16, wp, km.nbrT, None, km.n_in, cout, cin, P, tiles))
        y_old = me._conv_implicit_bf16(x16, wt, km.nbr, None, km.n_out, cin, cout, P, tiles)
        dx_old = me._conv_implicit_bf16(dy16, wp, km.nbrT, None, km.n_in, cout, cin, P, tiles)
        # tile kernel
        t_plan = timeit(lambda: me.build_tile_plan(km.nbr, P, tiles), 5, 1)
        plan = me.build_tile_plan(km.nbr, P, tiles)
        npass = plan.npass.cpu().numpy()
        print(f"  plan build {t_plan:.0f} us; passes/tile mean {npass.mean():.2f} max {npass.max()}; staged rows {int(plan.cursor[0])}"
              f" ({int(plan.cursor[0]) / max(km.n_out, 1):.2f} per output row)")
        lv = plan.live.cpu().numpy()
        bits = sum(((lv >> b) & 1).sum() for b in range(4))
        print(f"  live offsets per tile {float((lv > 0).sum(1).mean()):.0f} of {K}; live 32-row blocks among them {bits / max(int((lv > 0).sum()), 1) / 4:.2f}")
        wft, wfp = me._prep_bf16_group(ws, True, True), me._prep_bf16_group(ws, False, True)
        for ksplit in (1, 2, 4, 8):
            if ksplit > 1 and plan.ntile * ksplit > 1024:
                continue
            t_f = timeit(lambda: me._conv_tile(x16, wft, plan, None, cin, cout, km.n_in, P, ksplit, G))
            t_b = timeit(lambda: me._conv_tile(dy16, wfp, plan, None, cout, cin, km.n_out, P, ksplit, G, wrev=True))
            print(f"  ksplit {ksplit}: tile fwd {t_f:.0f} us (dense-map {t_old_f:.0f}), dgrad {t_b:.0f} us (dense-map {t_old_b:.0f})")
        y = me._conv_tile(x16, wft, plan, None, cin, cout, km.n_in, P, 1, G)
        dx = me._conv_tile(dy16, wfp, plan, None, cout, cin, km.n_out, P, 1, G, wrev=True)
        print(f"  max |diff| fwd {float((y - y_old).abs().max()):.2e} (|y| {float(y_old.abs().max()):.2f}), dgrad {float((dx - dx_old).abs().max()):.2e}")


if __name__ == "__main__":
    main()
