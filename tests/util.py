"""Seeded input generators shared by the parity tests."""
import numpy as np
import torch


def rand_coords(n, batch=2, extent=24, seed=0, dup=0.15):
    g = torch.Generator().manual_seed(seed)
    c = torch.randint(-extent, extent, (n, 3), generator=g)
    b = torch.randint(0, batch, (n, 1), generator=g)
    coords = torch.cat([b, c], 1).int()
    ndup = int(n * dup)
    if ndup:
        src = torch.randint(0, n, (ndup,), generator=g)
        dst = torch.randint(0, n, (ndup,), generator=g)
        coords[dst] = coords[src]
    return coords.contiguous()


def surface_coords(n, batch=2, extent=40, seed=0):
    """Voxels on a few planes + boxes: neighbourhood occupancy like an indoor scan."""
    rng = np.random.RandomState(seed)
    out = []
    for b in range(batch):
        pts = []
        m = n // 3
        xy = rng.randint(-extent, extent, (m, 2))
        pts.append(np.c_[xy, np.zeros(m, int)])                       # floor
        xz = rng.randint(-extent, extent, (m, 2))
        pts.append(np.c_[xz[:, 0], np.full(m, extent), np.abs(xz[:, 1]) // 2])  # wall
        c = rng.randint(-extent // 2, extent // 2, (n - 2 * m, 3))
        c[:, rng.randint(0, 3)] = 5                                     # a slab
        pts.append(c)
        p = np.concatenate(pts)
        out.append(np.c_[np.full(len(p), b), p])
    return torch.from_numpy(np.concatenate(out)).int().contiguous()


def rand_boxes(n, seed=0, yaw=True, extent=4.0, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    ctr = (torch.rand(n, 3, generator=g) - 0.5) * 2 * extent
    size = torch.rand(n, 3, generator=g) * 1.5 + 0.2
    ang = (torch.rand(n, 1, generator=g) - 0.5) * 6.28 if yaw else torch.zeros(n, 1)
    return torch.cat([ctr, size, ang], 1).float().contiguous().to(device)


def morton_keys(coords):
    """numpy uint64 keys of int32 [n,4] (batch, x, y, z) rows: batch << 45 | 15-bit interleave of the coordinates biased
    by 2^14 (x most significant) -- the order cg3d_morton_order sorts by."""
    c = np.asarray(coords).astype(np.int64)

    def spread(v):
        v = v.astype(np.uint64) & np.uint64(0x7fff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    return (c[:, 0].astype(np.uint64) << np.uint64(45)) | (spread(c[:, 1] + 16384) << np.uint64(2)) | \
        (spread(c[:, 2] + 16384) << np.uint64(1)) | spread(c[:, 3] + 16384)
