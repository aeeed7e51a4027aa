"""ScanNet / SUN RGB-D scene loaders, train-time augmentation and batch collation for the CAGroup3D path
(SURVEY 8(f) rank 2).  Mirrors, on the same on-disk formats and with the same host RNG call sequence:

  pcdet/datasets/scannet/scannet_dataset.py:11-273, pcdet/datasets/sunrgbd/sunrgbd_dataset.py:63-259
      `<root>/points/<id>.bin` float32 [N,6] (x,y,z,r,g,b); ScanNet also `<root>/instance_mask/<id>.bin`,
      `<root>/semantic_mask/<id>.bin` int64 [N]; `<root>/<name>_infos_<split>.pkl` = list of
      {'point_cloud': {'lidar_idx'}, 'annos': {'gt_num', 'name', 'location', 'dimensions', 'gt_boxes_upright_depth',
      'class', ['axis_align_matrix']}} (mmdet3d 0.15 info files)
  pcdet/datasets/augmentor/data_augmentor.py:43-133,245-326 and augmentor_utils.py:8-104,146-198,707-755
      the augmentations the two dataset yamls name, applied in yaml order, then heading wrap + box mask
  pcdet/datasets/dataset.py:159-230 (collate_batch)

Functions keep the reference's in-place semantics on the arrays they are handed and draw from `np.random` exactly
where the reference does, so a seeded run reproduces its augmented scenes (tests/test_indoor_dataset.py)."""
import copy
import os
import pickle

import numpy as np

# ------------------------------------------------------------------------------------------ augmentation primitives


def _rotate_z(xyz, angle):
    """rows (x,y,z,...) rotated about +z by `angle` (x towards y), float32 like common_utils.rotate_points_along_z."""
    c, s = np.float32(np.cos(np.float32(angle))), np.float32(np.sin(np.float32(angle)))
    rot = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1]], dtype=np.float32)
    out = np.array(xyz, dtype=np.float32, copy=True)
    out[:, :3] = xyz[:, :3].astype(np.float32) @ rot
    return out


def random_flip_along_x(gt_boxes, points):
    if np.random.choice([False, True], replace=False, p=[0.5, 0.5]):
        gt_boxes[:, 1] = -gt_boxes[:, 1]
        gt_boxes[:, 6] = -gt_boxes[:, 6]
        points[:, 1] = -points[:, 1]
    return gt_boxes, points


def random_flip_along_y(gt_boxes, points):
    if np.random.choice([False, True], replace=False, p=[0.5, 0.5]):
        gt_boxes[:, 0] = -gt_boxes[:, 0]
        gt_boxes[:, 6] = -(gt_boxes[:, 6] + np.pi)
        points[:, 0] = -points[:, 0]
    return gt_boxes, points


def global_rotation(gt_boxes, points, rot_range, heading_sign=1.0):
    """heading_sign +1: pcdet convention (ScanNet yaml), -1: the mmdet3d variant used by the SUN RGB-D yaml."""
    angle = np.random.uniform(rot_range[0], rot_range[1])
    points = _rotate_z(points, angle)
    gt_boxes[:, 0:3] = _rotate_z(gt_boxes[:, 0:3], angle)
    gt_boxes[:, 6] += heading_sign * angle
    return gt_boxes, points


def global_scaling(gt_boxes, points, scale_range):
    if scale_range[1] - scale_range[0] < 1e-3:
        return gt_boxes, points
    s = np.random.uniform(scale_range[0], scale_range[1])
    points[:, :3] *= s
    gt_boxes[:, :6] *= s
    return gt_boxes, points


def random_translation(gt_boxes, points, std, axis):
    off = np.random.normal(0, std, 1)
    points[:, axis] += off
    gt_boxes[:, axis] += off
    return gt_boxes, points


def global_alignment(points, axis_align_matrix, rotation_axis=2):
    rot, trans = axis_align_matrix[:3, :3], axis_align_matrix[:3, -1]
    unit = np.zeros(3)
    unit[rotation_axis] = 1.0
    ok = np.allclose(np.linalg.det(rot), 1.0) and (rot[rotation_axis, :] == unit).all() and (rot[:, rotation_axis] == unit).all()
    assert ok, "invalid rotation matrix %s" % rot
    points[:, :3] = points[:, :3] @ rot.T
    points[:, :3] += trans
    return points


def point_seg_class_mapping(semantic_mask, valid_cat_ids, max_cat_id):
    assert max_cat_id >= max(valid_cat_ids)
    table = np.full(int(max_cat_id) + 1, len(valid_cat_ids), dtype=np.int64)     # everything else -> background class
    table[np.asarray(valid_cat_ids)] = np.arange(len(valid_cat_ids))
    return table[semantic_mask]


def points_random_sampling(points, num_samples):
    choices = np.random.choice(points.shape[0], num_samples, replace=points.shape[0] < num_samples)
    return points[choices], choices


def limit_period(val, offset=0.5, period=np.pi):
    return val - np.floor(val / period + offset) * period


class IndoorAugmentor:
    """The `DATA_AUGMENTOR_*` section of a dataset yaml: a queue of named steps over {'points', 'gt_boxes', masks ...}."""

    def __init__(self, aug_cfg):
        disabled = aug_cfg.get("DISABLE_AUG_LIST", [])
        self.queue = [(c["NAME"], c) for c in aug_cfg["AUG_CONFIG_LIST"] if c["NAME"] not in disabled]
        for name, _ in self.queue:
            if not hasattr(self, "_" + name):
                raise NotImplementedError("augmentation %s is not on the CAGroup3D path" % name)

    def forward(self, d):
        for name, cfg in self.queue:
            d = getattr(self, "_" + name)(d, cfg)
        d["gt_boxes"][:, 6] = limit_period(d["gt_boxes"][:, 6], offset=0.5, period=2 * np.pi)
        if "gt_boxes_mask" in d:
            m = d.pop("gt_boxes_mask")
            d["gt_boxes"], d["gt_names"] = d["gt_boxes"][m], d["gt_names"][m]
        return d

    @staticmethod
    def _random_world_flip(d, cfg):
        for ax in cfg["ALONG_AXIS_LIST"]:
            assert ax in ("x", "y")
            d["gt_boxes"], d["points"] = (random_flip_along_x if ax == "x" else random_flip_along_y)(d["gt_boxes"], d["points"])
        return d

    @staticmethod
    def _rot_range(cfg):
        r = cfg["WORLD_ROT_ANGLE"]
        return list(r) if isinstance(r, (list, tuple)) else [-r, r]

    @staticmethod
    def _random_world_rotation(d, cfg):
        d["gt_boxes"], d["points"] = global_rotation(d["gt_boxes"], d["points"], IndoorAugmentor._rot_range(cfg), 1.0)
        return d

    @staticmethod
    def _random_world_rotation_mmdet3d(d, cfg):
        d["gt_boxes"], d["points"] = global_rotation(d["gt_boxes"], d["points"], IndoorAugmentor._rot_range(cfg), -1.0)
        return d

    @staticmethod
    def _random_world_scaling(d, cfg):
        d["gt_boxes"], d["points"] = global_scaling(d["gt_boxes"], d["points"], cfg["WORLD_SCALE_RANGE"])
        return d

    @staticmethod
    def _random_world_translation(d, cfg):
        if cfg["NOISE_TRANSLATE_STD"] == 0:
            return d
        for ax in cfg["ALONG_AXIS_LIST"]:
            d["gt_boxes"], d["points"] = random_translation(d["gt_boxes"], d["points"], cfg["NOISE_TRANSLATE_STD"], "xyz".index(ax))
        return d

    @staticmethod
    def _global_alignment(d, cfg):
        m = d["axis_align_matrix"]
        assert m.shape == (4, 4)
        d["points"] = global_alignment(d["points"], m, cfg["rotation_axis"])
        return d

    @staticmethod
    def _point_seg_class_mapping(d, cfg):
        d["semantic_mask"] = point_seg_class_mapping(d["semantic_mask"], cfg["valid_cat_ids"], cfg["max_cat_id"])
        return d

    @staticmethod
    def _indoor_point_sample(d, cfg):
        d["points"], ch = points_random_sampling(d["points"], cfg["num_points"])
        for k in ("instance_mask", "semantic_mask"):
            if d.get(k) is not None:
                d[k] = d[k][ch]
        return d


# ------------------------------------------------------------------------------------------ datasets

class IndoorDataset:
    """One split of ScanNet (`kind='scannet'`) or SUN RGB-D (`kind='sunrgbd'`) from the reference's processed folder."""

    def __init__(self, dataset_cfg, class_names, training=True, root_path=None, kind=None, filter_empty_gt=True):
        self.cfg, self.class_names, self.training = dataset_cfg, list(class_names), training
        self.kind = kind or ("sunrgbd" if "sunrgbd" in str(dataset_cfg.get("DATASET", "")).lower() else "scannet")
        self.root = str(root_path if root_path is not None else dataset_cfg["DATA_PATH"])
        mode = "train" if training else "test"
        self.point_cloud_range = np.array(dataset_cfg["POINT_CLOUD_RANGE"], dtype=np.float32)
        self.get_items = list(dataset_cfg.get("GET_ITEM_LIST", ["points"]))
        self.filter_empty_gt = filter_empty_gt
        infos = []
        for name in dataset_cfg["INFO_PATH"][mode]:
            p = os.path.join(self.root, name)
            if os.path.exists(p):
                with open(p, "rb") as f:
                    infos.extend(pickle.load(f))
        self.infos = infos * int(dataset_cfg.get("REPEAT", {}).get(mode, 1))
        self.augmentor = IndoorAugmentor(dataset_cfg["DATA_AUGMENTOR_TRAIN" if training else "DATA_AUGMENTOR_TEST"])

    def __len__(self):
        return len(self.infos)

    def _file(self, sub, idx):
        name = str(idx).zfill(6) if self.kind == "sunrgbd" else str(idx)
        path = os.path.join(self.root, sub, name + ".bin")
        assert os.path.exists(path), path
        return path

    def __getitem__(self, index):
        info = copy.deepcopy(self.infos[index])
        idx, annos = info["point_cloud"]["lidar_idx"], info["annos"]
        d = {"frame_id": idx}
        if annos["gt_num"] != 0:
            if self.kind == "sunrgbd":
                boxes = np.asarray(annos["gt_boxes_upright_depth"], dtype=np.float32)[:, :7]
            else:       # ScanNet boxes are axis-aligned: heading 0
                boxes = np.concatenate([annos["location"], annos["dimensions"], np.zeros((len(annos["location"]), 1))], 1)
            d.update(gt_names=np.asarray(annos["name"]), gt_boxes=boxes.astype(np.float32))
        else:
            d.update(gt_names=np.array([]), gt_boxes=np.zeros((0, 7), dtype=np.float32))
        if "points" in self.get_items:
            d["points"] = np.fromfile(self._file("points", idx), dtype=np.float32).reshape(-1, 6)
        if "instance_mask" in self.get_items:
            d["instance_mask"] = np.fromfile(self._file("instance_mask", idx), dtype=np.int64)
        if "semantic_mask" in self.get_items:
            d["semantic_mask"] = np.fromfile(self._file("semantic_mask", idx), dtype=np.int64)
        if self.kind == "scannet":
            d["axis_align_matrix"] = (np.array(annos["axis_align_matrix"]).astype(np.float32) if "axis_align_matrix" in annos
                                      else np.eye(4, dtype=np.float32))
        d = self.prepare_data(d)
        if self.training and len(d["gt_boxes"]) == 0 and self.filter_empty_gt:
            return self[np.random.randint(len(self))]
        return d

    def prepare_data(self, d):
        d["gt_boxes_mask"] = np.array([n in self.class_names for n in d["gt_names"]], dtype=np.bool_)
        d = self.augmentor.forward(d)
        keep = np.array([i for i, n in enumerate(d["gt_names"]) if n in self.class_names], dtype=np.int64)
        names = d["gt_names"][keep]
        cls = np.array([self.class_names.index(n) for n in names], dtype=np.int32)
        d["gt_boxes"] = np.concatenate((d["gt_boxes"][keep], cls.reshape(-1, 1).astype(np.float32)), axis=1)
        # (the indoor datasets override prepare_data and never run the yaml's DATA_PROCESSOR range mask:
        #  scannet_dataset.py:152-205 vs dataset.py:148)
        d.pop("gt_names", None)
        d.pop("axis_align_matrix", None)
        return d

    def gt_annos(self):
        return [copy.deepcopy(i["annos"]) for i in self.infos]

    @staticmethod
    def collate_batch(samples):
        """List of per-scene dicts -> batch_dict: points get the batch index in column 0, gt_boxes are zero-padded to
        the batch maximum, the two mask lists stay per-scene lists (dataset.py:159-230)."""
        out = {"batch_size": len(samples)}
        out["points"] = np.concatenate([np.pad(s["points"], ((0, 0), (1, 0)), mode="constant", constant_values=i)
                                        for i, s in enumerate(samples)], axis=0)
        gmax = max(len(s["gt_boxes"]) for s in samples)
        gt = np.zeros((len(samples), gmax, samples[0]["gt_boxes"].shape[-1]), dtype=np.float32)
        for i, s in enumerate(samples):
            gt[i, :len(s["gt_boxes"])] = s["gt_boxes"]
        out["gt_boxes"] = gt
        for k in ("semantic_mask", "instance_mask"):
            if k in samples[0]:
                out[k] = [s[k] for s in samples]
        out["frame_id"] = np.stack([s["frame_id"] for s in samples], axis=0)
        return out


def write_processed_scene(root, idx, points, gt_boxes, gt_names, kind="scannet", instance_mask=None, semantic_mask=None,
                          class_ids=None, axis_align_matrix=None):
    """Store one scene in the reference's processed layout and return its info-file entry (used by the tests and to
    export synthetic scenes; the real datasets are produced by the reference's mmdet3d-style converters)."""
    name = str(idx).zfill(6) if kind == "sunrgbd" else str(idx)
    for sub, arr, dt in (("points", points, np.float32), ("instance_mask", instance_mask, np.int64),
                         ("semantic_mask", semantic_mask, np.int64)):
        if arr is not None:
            os.makedirs(os.path.join(root, sub), exist_ok=True)
            np.asarray(arr, dtype=dt).tofile(os.path.join(root, sub, name + ".bin"))
    gt_boxes = np.asarray(gt_boxes, dtype=np.float32).reshape(-1, 7)
    annos = {"gt_num": len(gt_boxes), "name": np.asarray(gt_names), "location": gt_boxes[:, :3], "dimensions": gt_boxes[:, 3:6],
             "gt_boxes_upright_depth": gt_boxes if kind == "sunrgbd" else gt_boxes[:, :6],
             "class": np.asarray(class_ids if class_ids is not None else np.zeros(len(gt_boxes)), dtype=np.int64)}
    if axis_align_matrix is not None:
        annos["axis_align_matrix"] = np.asarray(axis_align_matrix)
    return {"point_cloud": {"num_features": 6, "lidar_idx": idx}, "annos": annos}
