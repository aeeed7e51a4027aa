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
    "S100k-yaw": (4, 100000, 10, True, True),     # SUN RGB-D shaped: one depth camera in a room corner
    "S200k": (5, 200000, 18, False, False),
    "S5k": (9, 5000, 18, False, False),   # small parity-test scene
    "S2k": (13, 2000, 18, False, False),  # the smallest scene that still has rows at every stride (CPU tests of whole launch programs)
    "S5k-yaw": (8, 5000, 10, True, False),
    "S5k-yaw-sv": (7, 5000, 10, True, True),      # small single-view scene (tests)
    # LEARNABLE variants (the convergence / mAP evidence runs, tools/synthetic_convergence.py): the class of an object is
    # a function of its shape (6 length bins x 3 height bins) instead of SURVEY 8(d)'s uniformly random label, which no
    # detector can predict on held-out scenes
    "S50k-shape": (11, 50000, 18, False, False),
    "S20k-shape": (12, 20000, 18, False, False),
}
SHAPE_LABELS = ("S50k-shape", "S20k-shape")


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


def _visible(cam, pts, centers, sizes, yaws):
    """Which points a camera at `cam` sees: the segment cam -> point must not cross any object box (slab test in the box
    frame; a point on a face turned away from the camera is hidden by its own box)."""
    vis = np.ones(len(pts), bool)
    d = pts - cam[None]
    for c, sz, yw in zip(centers, sizes, yaws):
        co, si = np.cos(-yw), np.sin(-yw)
        rot = np.array([[co, -si, 0.0], [si, co, 0.0], [0.0, 0.0, 1.0]])
        o = (cam - c) @ rot.T
        dd = d @ rot.T
        half = np.asarray(sz) / 2 - 1e-4                       # slightly shrunk: points ON a visible face stay visible
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (-half[None] - o[None]) / dd
            t2 = (half[None] - o[None]) / dd
        tmin = np.nanmax(np.minimum(t1, t2), axis=1)
        tmax = np.nanmin(np.maximum(t1, t2), axis=1)
        vis &= ~((tmax >= np.maximum(tmin, 0.0)) & (tmin < 1.0 - 1e-3))
    return vis


def make_scene(config="S50k", scene_idx=0):
    cid, n_pts, n_cls, with_yaw, single_view = CONFIGS[config]
    if single_view:
        return _make_scene_single_view(config, scene_idx)
    return _make_scene_full(config, scene_idx)


def _make_scene_single_view(config, scene_idx):
    """the SUN RGB-D-shaped scene keeps only what ONE corner camera sees -- the full scene is
    sampled 4x denser, occluded points (behind or on the far side of an object) are dropped, and n_points of the visible
    ones are kept: fewer surfaces at a higher density, as a single depth frame has."""
    cid, n_pts, n_cls, with_yaw, _ = CONFIGS[config]
    dense = _make_scene_full(config, scene_idx, n_points=4 * n_pts)
    rng = np.random.RandomState(1000 * cid + scene_idx + 500000)
    p = dense["points"][:, :3].astype(np.float64)
    lo, hi = p.min(0), p.max(0)
    cam = np.array([lo[0] + 0.15, lo[1] + 0.15, 1.5])
    gt = dense["gt_boxes"]
    vis = _visible(cam, p, gt[:, :3].astype(np.float64), gt[:, 3:6].astype(np.float64), gt[:, 6].astype(np.float64))
    keep = np.nonzero(vis)[0]
    keep = keep[rng.permutation(len(keep))[:n_pts]] if len(keep) >= n_pts else keep[rng.randint(0, len(keep), n_pts)]
    return {"points": dense["points"][keep], "gt_boxes": gt, "instance_mask": dense["instance_mask"][keep],
            "semantic_mask": dense["semantic_mask"][keep]}


def _make_scene_full(config="S50k", scene_idx=0, n_points=None):
    cid, n_pts, n_cls, with_yaw, _ = CONFIGS[config]
    n_pts = n_points or n_pts
    rng = np.random.RandomState(1000 * cid + scene_idx)
    L, W, H = rng.uniform(4, 8), rng.uniform(3, 6), rng.uniform(2.4, 3.0)
    n_obj = rng.randint(12, 21)
    sizes = np.c_[rng.uniform(0.4, 1.8, n_obj), rng.uniform(0.4, 1.0, n_obj), rng.uniform(0.4, 1.6, n_obj)]
    centers = np.c_[rng.uniform(-L / 2 + 0.9, L / 2 - 0.9, n_obj), rng.uniform(-W / 2 + 0.5, W / 2 - 0.5, n_obj),
                    sizes[:, 2] / 2]
    yaws = rng.uniform(-np.pi, np.pi, n_obj) if with_yaw else np.zeros(n_obj)
    labels = rng.randint(0, n_cls, n_obj)
    if config in SHAPE_LABELS:
        labels = (np.clip(((sizes[:, 0] - 0.4) / 1.4 * 6).astype(int), 0, 5) * 3 + np.clip(((sizes[:, 2] - 0.4) / 1.2 * 3).astype(int), 0, 2))
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
