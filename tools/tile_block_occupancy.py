"""How much of a 128-row tile's multiply work lands on empty 32-row blocks?  (CPU, oracle library.)

For the 3^3 stride-1 maps of the S50k x 4 batch at every backbone stride: per live (tile, offset) of the tile plan the
kernel multiplies all 128 rows; this counts the rows that have a neighbour, and the 32-row (one MFMA block) and 16-row
sub-blocks of a live (tile, offset) that hold none at all.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cagroup3d_amd import _lib, build_model, me as ME  # noqa: E402

ROOT = os.path.join(os.path.dirname(__file__), "..")
with _lib.use_library(_lib.bind(os.path.join(ROOT, "oracle", "liboracle.so"))):
    batch = build_model.synthetic_batch(sys.argv[1] if len(sys.argv) > 1 else "S50k", 4, device="cpu")
    pts = batch["points"]
    coords = pts[:, :4].clone()
    coords[:, 1:] /= 0.02
    ME.set_coords_only(True)
    sp = ME.SparseTensor(coordinates=coords, features=pts[:, 4:])
    ME.set_coords_only(False)
    mgr, key = sp.coordinate_manager, sp.coordinate_map_key
    for level in range(5):
        key = mgr.stride(key, 2)
        km = mgr.kernel_map(key, key, 3, 1, False)
        nbr = km.nbr.numpy()                     # [27, n]
        K, n = nbr.shape
        nt = -(-n // 128)
        pad = np.full((K, nt * 128), -1, np.int32)
        pad[:, :n] = nbr
        v = (pad >= 0).reshape(K, nt, 128)
        live = v.any(2)                           # [K, nt]
        rows_live = v.sum(2)[live]
        b32 = v.reshape(K, nt, 4, 32).any(3)[live]          # [L, 4]
        b16 = v.reshape(K, nt, 8, 16).any(3)[live]
        b64 = v.reshape(K, nt, 2, 64).any(3)[live]
        print(f"stride {2 << level:3d}: rows {n:7d} tiles {nt:5d}  pairs {int(v.sum()):9d}  live offsets / tile {live.sum() / nt:5.2f}"
              f"  fill of a live (tile, offset) {rows_live.mean() / 128:.3f}  non-empty 64-blocks {b64.mean():.3f}"
              f"  32-blocks {b32.mean():.3f}  16-blocks {b16.mean():.3f}")

# ---- the same after permuting the rows inside a window by their 27-bit neighbour mask
def blocks_after_sort(v_full, n, window, blk):
    """v_full bool [K, n]; rows sorted by mask inside windows of `window` rows; -> (work in units of blk rows, live (tile, off))"""
    K = v_full.shape[0]
    # offsets ordered by how evenly they split the rows (closest to half first) make the leading bits the informative ones
    frac = v_full.mean(1)
    order_k = np.argsort(np.abs(frac - 0.5))
    mask = np.zeros(n, np.int64)
    for i, k in enumerate(order_k):
        mask |= v_full[k].astype(np.int64) << (K - 1 - i)
    perm = np.arange(n)
    for s in range(0, n, window):
        e = min(n, s + window)
        perm[s:e] = s + np.argsort(mask[s:e], kind="stable")
    vp = v_full[:, perm]
    nt = -(-n // 128)
    pad = np.zeros((K, nt * 128), bool)
    pad[:, :n] = vp
    t = pad.reshape(K, nt, 128)
    live = t.any(2)
    b = t.reshape(K, nt, 128 // blk, blk).any(3)
    return b.sum(), live.sum() * (128 // blk)


with _lib.use_library(_lib.bind(os.path.join(ROOT, "oracle", "liboracle.so"))):
    key = sp.coordinate_map_key
    for level in range(4):
        key = mgr.stride(key, 2)
        nbr = mgr.kernel_map(key, key, 3, 1, False).nbr.numpy()
        v = nbr >= 0
        n = v.shape[1]
        out = []
        for window in (128, 512, 1024, 4096):
            for blk in (32, 16):
                w, l = blocks_after_sort(v, n, window, blk)
                out.append(f"w{window}/b{blk} {w / l:.3f}")
        print(f"stride {2 << level:3d}: non-empty blocks / blocks of live (tile, offset) in the unsorted plan: " + "  ".join(out))
