"""Indoor data pipeline: augmentation against vectors produced by the reference's own
DataAugmentor + yamls (tests/golden/make_dataset_fixtures.py), on-disk format round trip, collation, and one
training step of the detector on a batch that went through the whole loader."""
import copy
import os
import pickle
import sys

import numpy as np
import pytest
import yaml

from cagroup3d_amd.pcdet.datasets import indoor_dataset as ds

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_dataset_fixtures as mk          # noqa: E402  (the committed scene generator; the reference is not imported)

G = np.load(os.path.join(HERE, "golden", "indoor_dataset_vectors.npz"))
CFG_DIR = os.path.join(os.path.dirname(HERE), "cagroup3d_amd", "cfgs", "dataset_configs")


def _cfg(kind):
    return yaml.safe_load(open(os.path.join(CFG_DIR, "%s_dataset.yaml" % kind)))


@pytest.mark.parametrize("kind,yaw", [("scannet", False), ("sunrgbd", True)])
@pytest.mark.parametrize("mode", ["train", "test"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_augmentation_pipeline_matches_reference(kind, yaw, mode, seed):
    aug = ds.IndoorAugmentor(_cfg(kind)["DATA_AUGMENTOR_" + mode.upper()])
    d = mk.scene(100 + seed, yaw=yaw)
    np.random.seed(seed)
    r = aug.forward(copy.deepcopy(d))
    tag = "%s_%s_%d_" % (kind, mode, seed)
    if tag + "points" in G:
        np.testing.assert_allclose(r["points"], G[tag + "points"], rtol=2e-6, atol=2e-6)
    else:
        assert len(r["points"]) == int(G[tag + "points_n"])
        np.testing.assert_allclose(r["points"][:1500], G[tag + "points_head"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(r["points"].astype(np.float64).sum(0), G[tag + "points_sum"], rtol=1e-6)
    np.testing.assert_allclose(r["gt_boxes"], G[tag + "gt_boxes"], rtol=2e-6, atol=2e-6)
    assert list(r["gt_names"]) == list(G[tag + "gt_names"])
    for k in ("instance_mask", "semantic_mask"):
        if tag + k in G:
            assert np.array_equal(r[k], G[tag + k])
    assert (np.abs(r["gt_boxes"][:, 6]) <= np.pi + 1e-6).all()            # heading wrapped to [-pi, pi)


def _write_split(root, kind, n=3):
    infos = []
    for i in range(n):
        s = mk.scene(200 + i, n=1200, g=5, yaw=kind == "sunrgbd")
        if i == n - 1:
            s["gt_boxes"], s["gt_names"] = s["gt_boxes"][:0], s["gt_names"][:0]          # a scene without objects
        infos.append(ds.write_processed_scene(
            str(root), i if kind == "sunrgbd" else "scene%04d_00" % i, s["points"], s["gt_boxes"], s["gt_names"], kind=kind,
            instance_mask=s.get("instance_mask"), semantic_mask=s.get("semantic_mask"), axis_align_matrix=s.get("axis_align_matrix")))
    for split in ("train", "val"):
        with open(os.path.join(str(root), "%s_infos_%s.pkl" % (kind, split)), "wb") as f:
            pickle.dump(infos, f)
    return infos


@pytest.mark.parametrize("kind", ["scannet", "sunrgbd"])
def test_on_disk_format_round_trip_and_collate(tmp_path, kind):
    infos = _write_split(tmp_path, kind)
    cfg = _cfg(kind)
    for a in cfg["DATA_AUGMENTOR_TRAIN"]["AUG_CONFIG_LIST"] + cfg["DATA_AUGMENTOR_TEST"]["AUG_CONFIG_LIST"]:
        if a["NAME"] == "indoor_point_sample":
            a["num_points"] = 1000
    val = ds.IndoorDataset(cfg, mk.CLASSES, training=False, root_path=tmp_path)
    trn = ds.IndoorDataset(cfg, mk.CLASSES, training=True, root_path=tmp_path)
    assert len(val) == 3 and len(trn) == 3 * cfg["REPEAT"]["train"]
    s0 = val[0]
    assert s0["points"].shape[1] == 6 and s0["gt_boxes"].shape[1] == 8 and "gt_names" not in s0
    assert set(np.unique(s0["gt_boxes"][:, 7]).astype(int)) <= set(range(len(mk.CLASSES)))
    if kind == "scannet":
        assert s0["points"].shape[0] == 1200 and len(s0["instance_mask"]) == 1200
        assert (s0["gt_boxes"][:, 6] == 0).all()
        np.random.seed(3)
        assert trn[0]["semantic_mask"].max() <= 18         # NYU40 ids -> 18 classes + background, train-time only (the yaml)
    else:
        assert s0["points"].shape[0] == 1000 and "instance_mask" not in s0
    assert len(val[2]["gt_boxes"]) == 0                      # evaluation keeps the empty scene ...
    np.random.seed(0)
    assert len(trn[2]["gt_boxes"]) > 0                       # ... training redraws another index (filter_empty_gt)
    batch = ds.IndoorDataset.collate_batch([val[0], val[1]])
    assert batch["batch_size"] == 2 and batch["points"].shape[1] == 7
    assert set(np.unique(batch["points"][:, 0])) == {0.0, 1.0}
    assert batch["gt_boxes"].shape[0] == 2 and batch["gt_boxes"].shape[2] == 8
    if kind == "scannet":
        assert isinstance(batch["semantic_mask"], list) and len(batch["instance_mask"]) == 2
    annos = val.gt_annos()
    assert annos[0]["gt_num"] == infos[0]["annos"]["gt_num"]
    assert annos[0]["gt_boxes_upright_depth"].shape[1] == (7 if kind == "sunrgbd" else 6)


def test_loader_batch_trains_the_detector(oracle, tmp_path):
    """A batch that went disk -> loader -> augmentation -> collate drives one step of the real detector."""
    import torch
    from cagroup3d_amd import _lib, build_model, synthetic
    from cagroup3d_amd.pcdet.models import load_data_to_gpu
    infos = []
    names = build_model.load_cfg("scannet").CLASS_NAMES
    inv = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]          # class index -> NYU40 id on disk
    for i in range(2):
        s = synthetic.make_scene("S5k", i)
        sem = np.array([inv[c] if c < 18 else 1 for c in s["semantic_mask"]], dtype=np.int64)
        infos.append(ds.write_processed_scene(str(tmp_path), "scene%04d_00" % i, s["points"], s["gt_boxes"][:, :7],
                                              [names[int(c)] for c in s["gt_boxes"][:, 7]], instance_mask=s["instance_mask"],
                                              semantic_mask=sem, axis_align_matrix=np.eye(4, dtype=np.float32)))
    for split in ("train", "val"):
        pickle.dump(infos, open(os.path.join(str(tmp_path), "scannet_infos_%s.pkl" % split), "wb"))
    data = ds.IndoorDataset(_cfg("scannet"), names, training=True, root_path=tmp_path)
    np.random.seed(1)
    batch = ds.IndoorDataset.collate_batch([data[0], data[1]])
    batch["cur_epoch"] = 0
    model, _ = build_model.build_cagroup3d("scannet", seed=0)
    model.train()
    with _lib.use_library(oracle):
        load_data_to_gpu(batch, "cpu")
        ret, tb, _ = model(batch)
        ret["loss"].backward()
    assert torch.isfinite(ret["loss"]) and tb["loss_vote"] > 0


def test_train_driver_on_a_processed_folder(oracle, tmp_path):
    from cagroup3d_amd import _lib, build_model, synthetic, train
    names = build_model.load_cfg("scannet").CLASS_NAMES
    inv = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]
    infos = []
    for i in range(2):
        s = synthetic.make_scene("S5k", i)
        sem = np.array([inv[c] if c < 18 else 1 for c in s["semantic_mask"]], dtype=np.int64)
        infos.append(ds.write_processed_scene(str(tmp_path), "scene%04d_00" % i, s["points"], s["gt_boxes"][:, :7],
                                              [names[int(c)] for c in s["gt_boxes"][:, 7]], instance_mask=s["instance_mask"],
                                              semantic_mask=sem, class_ids=s["gt_boxes"][:, 7], axis_align_matrix=np.eye(4, dtype=np.float32)))
    for split in ("train", "val"):
        pickle.dump(infos, open(os.path.join(str(tmp_path), "scannet_infos_%s.pkl" % split), "wb"))
    d = train.DiskIndoorDataset("scannet", str(tmp_path), names, 2, True, workers=0)
    d.data.infos = d.data.infos[:2]                          # REPEAT 10 -> one pass for the test
    assert len(d) == 1
    with _lib.use_library(oracle):
        model, cfg = build_model.build_cagroup3d("scannet", seed=0)
        opt = train.build_optimizer(model, cfg.OPTIMIZATION)
        sched = train.build_scheduler(opt, len(d), cfg.OPTIMIZATION)
        it = train.train_one_epoch(model, opt, sched, d, 0, 0, cfg.OPTIMIZATION.GRAD_NORM_CLIP, log=lambda *a: None)
        assert it == 1
        res = train.eval_one_epoch(model, train.DiskIndoorDataset("scannet", str(tmp_path), names, 2, False, workers=0), names,
                                   "cpu", log=lambda *a: None)
    assert "mAP_0.25" in res


def _write_synthetic_split(root, config, n):
    """`n` synthetic scenes of `config` in the processed-folder layout of scannet_dataset.py:62-75,223-273."""
    from cagroup3d_amd import build_model, synthetic
    names = build_model.load_cfg("scannet").CLASS_NAMES
    inv = [3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 14, 16, 24, 28, 33, 34, 36, 39]          # class index -> NYU40 id on disk
    infos = []
    for i in range(n):
        s = synthetic.make_scene(config, i)
        sem = np.array([inv[c] if c < 18 else 1 for c in s["semantic_mask"]], dtype=np.int64)
        infos.append(ds.write_processed_scene(str(root), "scene%04d_00" % i, s["points"], s["gt_boxes"][:, :7],
                                              [names[int(c)] for c in s["gt_boxes"][:, 7]], instance_mask=s["instance_mask"],
                                              semantic_mask=sem, class_ids=s["gt_boxes"][:, 7],
                                              axis_align_matrix=np.eye(4, dtype=np.float32)))
    for split in ("train", "val"):
        pickle.dump(infos, open(os.path.join(str(root), "scannet_infos_%s.pkl" % split), "wb"))
    return names


@pytest.mark.gpu
def test_loader_batch_drives_the_hip_path(oracle, hip, tmp_path):
    """SURVEY 8(f) rank 2 on the device: scenes written in the processed-folder layout -> DiskIndoorDataset (worker
    processes, train-time augmentation, collate_batch: dataset.py:159-230) -> load_data_to_gpu -> one training step of the
    detector through the HIP library, in the bench precision.  The voxelisation of the loader's batch is compared bit
    for bit with the oracle's on the same batch, the step's loss with the oracle's step (fp32), and the evaluation path
    (test-time augmentation, gt_annos from the info files) runs on the device too."""
    import torch
    from cagroup3d_amd import _lib, build_model, me, train
    from cagroup3d_amd.pcdet.models import load_data_to_gpu
    names = _write_synthetic_split(tmp_path, "S5k", 4)
    d = train.DiskIndoorDataset("scannet", str(tmp_path), names, 2, True, workers=2, seed=7)
    d.data.infos = d.data.infos[:4]                          # REPEAT 10 -> one pass
    batches = list(d.batches(epoch=0, shuffle=False))
    assert len(batches) == 2
    batch = batches[0]
    assert batch["points"].shape[1] == 7 and batch["batch_size"] == 2 and isinstance(batch["instance_mask"], list)
    ref_batch = copy.deepcopy(batch)

    def step(dev, lib, b, prec):
        model, cfg = build_model.build_cagroup3d("scannet", seed=0)
        model = model.to(dev).train()
        b["cur_epoch"] = 0
        me.PRECISION = prec
        try:
            with _lib.use_library(lib):
                load_data_to_gpu(b, dev)
                vox = model.voxelization(b["points"].clone())
                torch.manual_seed(3)
                np.random.seed(3)
                ret, tb, _ = model(b)
                ret["loss"].backward()
        finally:
            me.PRECISION = 0
        gn = torch.stack([p.grad.float().norm() for p in model.parameters() if p.grad is not None])
        assert torch.isfinite(ret["loss"]) and torch.isfinite(gn).all() and float(gn.sum()) > 0
        return vox.C.cpu(), tb

    c_hip, tb_hip = step("cuda", hip, copy.deepcopy(batch), 0)
    c_or, tb_or = step("cpu", oracle, ref_batch, 0)
    assert torch.equal(c_hip, c_or), "voxel rows of the loader's batch differ from the oracle's"
    assert set(tb_hip) == set(tb_or) and "loss_all" in tb_or
    for k in tb_or:
        assert abs(tb_hip[k] - tb_or[k]) <= 1e-2 * max(1.0, abs(tb_or[k])), (k, tb_hip[k], tb_or[k])
    _, tb_bf16 = step("cuda", hip, copy.deepcopy(batch), 1)               # the bench precision on the loader's batch
    assert abs(tb_bf16["loss_all"] - tb_hip["loss_all"]) <= 2e-2 * abs(tb_hip["loss_all"]), (tb_bf16["loss_all"], tb_hip["loss_all"])
    # the driver's own epoch functions on the folder, on the device (train-time and test-time pipelines)
    with _lib.use_library(hip):
        model, cfg = build_model.build_cagroup3d("scannet", seed=0)
        model = model.cuda()
        opt = train.build_optimizer(model, cfg.OPTIMIZATION)
        sched = train.build_scheduler(opt, len(d), cfg.OPTIMIZATION)
        it = train.train_one_epoch(model, opt, sched, d, 0, 0, cfg.OPTIMIZATION.GRAD_NORM_CLIP, log=lambda *a: None)
        assert it == 2
        val = train.DiskIndoorDataset("scannet", str(tmp_path), names, 2, False, workers=0)
        res = train.eval_one_epoch(model, val, names, "cuda", log=lambda *a: None)
    assert "mAP_0.25" in res
