"""Synthetic ScanNet / SUN RGB-D shaped scenes (pure numpy).

seed = 1000 * config_id + scene_idx.  A scene is a room (floor + 4 walls) with axis-aligned
(or yawed) boxes on the floor; points are sampled area-weighted on all surfaces with N(0, 4 mm)
noise and random colours; GT boxes are the objects.  Produces exactly the `batch_dict` keys the
reference's collate hands to the detector (pcdet/datasets/dataset.py:159-230,
pcdet/models/detectors/cagroup3d.py:27-50,99-157).
"""
import numpy as np

CONFIGS = {
    # name: (config_id, n_points, n_classes, yaw, single_view)
    "S50k": (1, 50000, 18, False, False),
    "S100k-yaw": (4, 100000, 10, True, False),
    "S200k": (5, 200000, 18, False, False),
    "S5k": (9, 5000, 18, False, False),   # small parity-test scene
    "S5k-yaw": (8, 5000, 10, True, False),
}


def _box_surface_points(rng, center, size, yaw, n):
    """n points on the 5 visible faces (no bottom) of a box."""
    dx, dy, dz = size
    areas = np.array([dx * dy, dx * dz, dx * dz, dy * dz, dy * dz])
    face = rng.choice(5, size=n, p=areas / areas.sum())
    u, v = rng.rand(n) - 0.5, rng.rand(n) - 0.5
    p = np.zeros((n, 3))
    m = face == 0
    p[m] = np.c_[u[m] * dx, v[m] * dy, np.full(m.sum(), dz / 2)]
    for f, sgn in ((1, -1), (2, 1)):
        m = face == f
        p[m] = np.c_[u[m] * dx, np.full(m.sum(), sgn * dy / 2), v[m] * dz]
    for f, sgn in ((3, -1), (4, 1)):
        m = face == f
        p[m] = np.c_[np.full(m.sum(), sgn * dx / 2), u[m] * dy, v[m] * dz]
    c, s = np.cos(yaw), np.sin(yaw)
    xy = p[:, :2] @ np.array([[c, s], [-s, c]])
    p = np.c_[xy, p[:, 2]]
    return p + np.asarray(center)[None]


def make_scene(config="S50k", scene_idx=0):
    cid, n_pts, n_cls, with_yaw, _ = CONFIGS[config]
    rng = np.random.RandomState(1000 * cid + scene_idx)
    L, W, H = rng.uniform(4, 8), rng.uniform(3, 6), rng.uniform(2.4, 3.0)
    n_obj = rng.randint(12, 21)
    sizes = np.c_[rng.uniform(0.4, 1.8, n_obj), rng.uniform(0.4, 1.0, n_obj), rng.uniform(0.4, 1.6, n_obj)]
    centers = np.c_[rng.uniform(-L / 2 + 0.9, L / 2 - 0.9, n_obj), rng.uniform(-W / 2 + 0.5, W / 2 - 0.5, n_obj),
                    sizes[:, 2] / 2]
    yaws = rng.uniform(-np.pi, np.pi, n_obj) if with_yaw else np.zeros(n_obj)
    labels = rng.randint(0, n_cls, n_obj)
    # surfaces: floor, 4 walls, objects
    areas = [L * W, L * H, L * H, W * H, W * H] + [2 * (s[0] * s[2] + s[1] * s[2]) + s[0] * s[1] for s in sizes]
    areas = np.asarray(areas)
    counts = rng.multinomial(n_pts, areas / areas.sum())
    pts, ins, sem = [], [], []
    u = rng.rand(counts[0], 2) - 0.5
    pts.append(np.c_[u[:, 0] * L, u[:, 1] * W, np.zeros(counts[0])])
    for w, (ax, sgn) in enumerate(((1, -1), (1, 1), (0, -1), (0, 1))):
        k = counts[1 + w]
        a, h = rng.rand(k) - 0.5, rng.rand(k) * H
        if ax == 1:
            pts.append(np.c_[a * L, np.full(k, sgn * W / 2), h])
        else:
            pts.append(np.c_[np.full(k, sgn * L / 2), a * W, h])
    nb = counts[:5].sum()
    ins.append(np.zeros(nb, np.int64))          # instance 0 = structure
    sem.append(np.full(nb, n_cls, np.int64))    # background class id == n_classes
    for o in range(n_obj):
        k = counts[5 + o]
        pts.append(_box_surface_points(rng, centers[o], sizes[o], yaws[o], k))
        ins.append(np.full(k, o + 1, np.int64))
        sem.append(np.full(k, labels[o], np.int64))
    pts = np.concatenate(pts) + rng.normal(0, 0.004, (n_pts, 3))
    perm = rng.permutation(n_pts)
    pts, ins, sem = pts[perm], np.concatenate(ins)[perm], np.concatenate(sem)[perm]
    rgb = rng.randint(0, 256, (n_pts, 3)).astype(np.float32)
    gt = np.c_[centers, sizes, yaws, labels].astype(np.float32)  # [G, 8]: x,y,z,dx,dy,dz,heading,class
    return {"points": np.c_[pts, rgb].astype(np.float32), "gt_boxes": gt,
            "instance_mask": ins, "semantic_mask": sem}


def make_batch(config="S50k", batch_size=4, first_scene=0):
    """Collated batch_dict (numpy): points [sum N, 7] with the batch index in column 0, gt_boxes
    [B, Gmax, 8] zero-padded (dataset.py:176-186), mask lists (dataset.py:221-222)."""
    scenes = [make_scene(config, first_scene + i) for i in range(batch_size)]
    pts = np.concatenate([np.c_[np.full(len(s["points"]), i, np.float32), s["points"]] for i, s in enumerate(scenes)])
    gmax = max(len(s["gt_boxes"]) for s in scenes)
    gt = np.zeros((batch_size, gmax, 8), np.float32)
    for i, s in enumerate(scenes):
        gt[i, : len(s["gt_boxes"])] = s["gt_boxes"]
    return {"points": pts.astype(np.float32), "gt_boxes": gt, "batch_size": batch_size,
            "instance_mask": [s["instance_mask"] for s in scenes],
            "semantic_mask": [s["semantic_mask"] for s in scenes], "cur_epoch": 0}
