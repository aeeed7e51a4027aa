"""The committed scene generator: determinism and the single-view (SUN RGB-D-shaped) configuration."""
import numpy as np

from cagroup3d_amd import synthetic


def test_scenes_are_deterministic_and_shaped_like_the_collate_output():
    a, b = synthetic.make_scene("S5k", 3), synthetic.make_scene("S5k", 3)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert a["points"].shape == (5000, 6) and a["gt_boxes"].shape[1] == 8
    batch = synthetic.make_batch("S5k", 2)
    assert batch["points"].shape == (10000, 7) and batch["gt_boxes"].shape[0] == 2 and len(batch["instance_mask"]) == 2


def test_single_view_scene_keeps_only_what_one_camera_sees():
    sv = synthetic.make_scene("S5k-yaw-sv", 0)
    full = synthetic._make_scene_full("S5k-yaw-sv", 0)
    assert sv["points"].shape == (5000, 6) and np.array_equal(sv["gt_boxes"], full["gt_boxes"])
    p = sv["points"][:, :3].astype(np.float64)
    dense = synthetic._make_scene_full("S5k-yaw-sv", 0, n_points=20000)["points"][:, :3].astype(np.float64)
    cam = np.array([dense[:, 0].min() + 0.15, dense[:, 1].min() + 0.15, 1.5])
    gt = sv["gt_boxes"].astype(np.float64)
    # every kept point is visible; a good part of the dense scene is not (object backs, shadows on floor and walls)
    assert synthetic._visible(cam, p, gt[:, :3], gt[:, 3:6], gt[:, 6]).all()
    hidden = ~synthetic._visible(cam, dense, gt[:, :3], gt[:, 3:6], gt[:, 6])
    assert 0.1 < hidden.mean() < 0.9
    # fewer surfaces at a higher density: the same number of points falls into fewer voxels
    vox = lambda q: len(np.unique(np.floor(q / 0.04).astype(np.int64), axis=0))
    assert vox(p) < vox(full["points"][:, :3])
    # some objects are not seen at all (their instance never appears), as in a single depth frame
    assert len(np.unique(sv["instance_mask"])) < len(np.unique(full["instance_mask"]))
