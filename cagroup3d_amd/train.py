"""Train / evaluate drivers for the CAGroup3D path: this build's counterparts of the reference's `tools/train.py`
(:59-202), `tools/train_utils/train_utils.py` (train_one_epoch :12-90, train_model :111-196, checkpoint_state /
save_checkpoint :199-228), `tools/train_utils/optimization/__init__.py` (build_optimizer :11-40, build_scheduler
:43-65) and `tools/test.py` + `eval_utils.eval_one_epoch` (prediction dicts -> `indoor_eval`).  The data side is the
committed synthetic scene generator (no datasets in this environment); the loop, the optimiser / step-decay
schedule / gradient clipping, the semantic-threshold schedule input (`cur_epoch`), the checkpoint layout
(`model_state`, `optimizer_state`, `epoch`, `it`) and the evaluation protocol follow the reference.

    python -m cagroup3d_amd.train --dataset scannet --config S5k --scenes 8 --epochs 2 --ckpt /tmp/ck.pth --eval
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m cagroup3d_amd.train ...     # data parallel over RCCL (grad_sync.py)
"""
import argparse
import os
import time

# one host thread for torch's CPU-side ops: every host tensor of a step is tiny, and the default OpenMP teams only
# add 50-200 ms stalls of the launching thread; torchrun sets the same default for its workers
os.environ.setdefault("OMP_NUM_THREADS", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from . import build_model, synthetic
from .pcdet.datasets.indoor_eval import indoor_eval
from .pcdet.models import load_data_to_gpu, model_fn_decorator
from .pcdet.models.detectors.detector3d_template import CHECKPOINT_VERSION


def shard_ids(ids, rank, world):
    """This rank's share of `ids`, as torch's DistributedSampler deals it (the reference's sampler,
    pcdet/datasets/__init__.py:66-71): the list is padded with its own head to a multiple of the world size, then
    rank r takes every world-th entry from r.  Every rank gets the SAME number of scenes -- hence the same number of
    batches and the same last-batch size -- so the per-step collectives (gradient buckets, the (B,3) loss-normaliser
    all-reduce) line up on every rank up to the end of the epoch."""
    ids = list(ids)
    if world <= 1 or not ids:
        return ids
    total = -(-len(ids) // world) * world
    while len(ids) < total:
        ids += ids[:total - len(ids)]
    return ids[rank:total:world]


class SyntheticIndoorDataset:
    """Scenes 0..n-1 of a synthetic configuration, sharded like DistributedSampler (scene i -> rank i mod W, padded to
    equal shares) and collated into the reference's batch_dict (dataset.py:159-230)."""

    def __init__(self, config, n_scenes, batch_size, rank=0, world=1, first_scene=0):
        self.config, self.batch_size, self.rank, self.world = config, batch_size, rank, world
        self.all_ids = list(range(first_scene, first_scene + n_scenes))
        self.scene_ids = shard_ids(self.all_ids, rank, world)

    def __len__(self):
        return (len(self.scene_ids) + self.batch_size - 1) // self.batch_size

    def n_total(self):
        return len(self.all_ids)

    def eval_positions(self):
        """Dataset positions of the scenes `batches()` (unshuffled) yields on this rank, in order (padding included)."""
        return shard_ids(range(len(self.all_ids)), self.rank, self.world)

    _cache = {}

    def _scene(self, s):
        """Scenes are deterministic in (config, index): generated once per process (0.3 s each at 50 k points)."""
        key = (self.config, s)
        if key not in SyntheticIndoorDataset._cache:
            if len(SyntheticIndoorDataset._cache) > 512:
                SyntheticIndoorDataset._cache.clear()
            SyntheticIndoorDataset._cache[key] = synthetic.make_scene(self.config, s)
        return SyntheticIndoorDataset._cache[key]

    def batches(self, epoch=0, shuffle=False):
        ids = list(self.all_ids)
        if shuffle:
            np.random.RandomState(epoch).shuffle(ids)           # epoch-seeded, like sampler.set_epoch: shuffle, THEN deal
        ids = shard_ids(ids, self.rank, self.world)
        for i in range(0, len(ids), self.batch_size):
            chunk = ids[i:i + self.batch_size]
            scenes = [self._scene(s) for s in chunk]
            pts = np.concatenate([np.c_[np.full(len(s["points"]), j, np.float32), s["points"]] for j, s in enumerate(scenes)])
            gmax = max(len(s["gt_boxes"]) for s in scenes)
            gt = np.zeros((len(scenes), gmax, 8), np.float32)
            for j, s in enumerate(scenes):
                gt[j, :len(s["gt_boxes"])] = s["gt_boxes"]
            yield {"points": pts.astype(np.float32), "gt_boxes": gt, "batch_size": len(scenes), "frame_id": np.array(chunk),
                   "instance_mask": [s["instance_mask"] for s in scenes], "semantic_mask": [s["semantic_mask"] for s in scenes]}

    @staticmethod
    def gt_annos(batch):
        """info['annos'] of the reference's info files: gt_num / gt_boxes_upright_depth / class (scannet_dataset.py:145)."""
        out = []
        for g in batch["gt_boxes"]:
            g = np.asarray(g.cpu() if torch.is_tensor(g) else g)
            g = g[~(g == 0).all(-1)]
            out.append({"gt_num": len(g), "gt_boxes_upright_depth": g[:, :7].astype(np.float32), "class": g[:, 7].astype(np.int64)})
        return out


class DiskIndoorDataset:
    """The reference's processed ScanNet / SUN RGB-D folder behind the same `.batches()` interface: IndoorDataset +
    torch DataLoader workers (per-worker numpy seeding like common_utils.worker_init_fn) + rank sharding."""

    def __init__(self, kind, root, class_names, batch_size, training, rank=0, world=1, workers=4, seed=666):
        import yaml
        from .pcdet.datasets.indoor_dataset import IndoorDataset
        cfg = yaml.safe_load(open(os.path.join(build_model.CFG_DIR, "dataset_configs", "%s_dataset.yaml" % kind)))
        self.data = IndoorDataset(cfg, class_names, training=training, root_path=root, kind=kind)
        self.batch_size, self.training, self.rank, self.world, self.workers, self.seed = batch_size, training, rank, world, workers, seed

    def __len__(self):
        return (len(shard_ids(range(len(self.data)), self.rank, self.world)) + self.batch_size - 1) // self.batch_size

    def gt_annos(self):
        return [self.data.gt_annos()[i] for i in shard_ids(range(len(self.data)), self.rank, self.world)]

    def n_total(self):
        return len(self.data)

    def eval_positions(self):
        return shard_ids(range(len(self.data)), self.rank, self.world)

    def batches(self, epoch=0, shuffle=False):
        ids = np.arange(len(self.data))
        if shuffle:
            np.random.RandomState(self.seed + epoch).shuffle(ids)
        ids = shard_ids(ids.tolist(), self.rank, self.world)
        seed = self.seed

        def init(worker_id):
            np.random.seed(seed + epoch * 1000 + worker_id)
        loader = torch.utils.data.DataLoader(torch.utils.data.Subset(self.data, ids), batch_size=self.batch_size, shuffle=False,
                                             num_workers=self.workers, collate_fn=self.data.collate_batch, worker_init_fn=init,
                                             drop_last=False)
        yield from loader

    @staticmethod
    def batch_gt_annos(batch):
        return SyntheticIndoorDataset.gt_annos(batch)


def build_optimizer(model, optim_cfg):
    params = [p for p in model.parameters() if p.requires_grad]
    name = optim_cfg.OPTIMIZER
    if name == "adam":
        return torch.optim.Adam(params, lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY)
    if name == "adamW":
        from .optim import ClippedAdamW                         # torch's fused AdamW + clip_grad_norm_ on cached tensor lists
        return ClippedAdamW(params, lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY)
    if name == "sgd":
        return torch.optim.SGD(params, lr=optim_cfg.LR, weight_decay=optim_cfg.WEIGHT_DECAY, momentum=optim_cfg.MOMENTUM)
    raise NotImplementedError(name)       # adam_onecycle (fastai wrapper) is not used by the CAGroup3D configurations


def build_scheduler(optimizer, iters_per_epoch, optim_cfg, last_it=-1):
    """LR x LR_DECAY at every DECAY_STEP_LIST epoch boundary, stepped per ITERATION, floored at LR_CLIP."""
    steps = [e * iters_per_epoch for e in optim_cfg.DECAY_STEP_LIST]
    floor = optim_cfg.get("LR_CLIP", 0.0) / optim_cfg.LR

    def factor(it):
        f = 1.0
        for s in steps:
            if it >= s:
                f *= optim_cfg.LR_DECAY
        return max(f, floor)
    return torch.optim.lr_scheduler.LambdaLR(optimizer, factor, last_epoch=last_it)


def checkpoint_state(model, optimizer, epoch, it):
    m = model.module if hasattr(model, "module") else model
    if hasattr(optimizer, "finish_late"):
        optimizer.finish_late()                 # (the late parameters' update may still be on its own stream: optim.ClippedAdamW.set_early)
    return {"epoch": epoch, "it": it, "model_state": {k: v.cpu() for k, v in m.state_dict().items()},
            "optimizer_state": optimizer.state_dict() if optimizer is not None else None, "version": CHECKPOINT_VERSION}


_FROZEN = [False]


def train_one_epoch(model, optimizer, scheduler, dataset, epoch, it, clip, rank=0, log=print, losses=None):
    model.train()
    model_func = model_fn_decorator()
    params = [p for p in model.parameters() if p.requires_grad]
    device = params[0].device
    core = model.module if hasattr(model, "module") else model

    def staged(it_batches):
        """Look one batch ahead: the next batch is moved to the device and its coordinate structures / data-only
        targets are built on the side stream right after the current step's backward has been queued."""
        prev = None
        for b in it_batches:
            b["cur_epoch"] = epoch                               # drives the semantic threshold (cagroup3d.py:31)
            load_data_to_gpu(b, device)
            if prev is not None:
                yield prev, b
            prev = b
        if prev is not None:
            yield prev, None
    prepared = None
    for batch, nxt in staged(dataset.batches(epoch, shuffle=True)):
        if prepared is not None:
            batch["prepared"] = prepared.result()
        # the next batch's coordinate structures: worker thread + side stream, under this whole step
        prepared = core.prefetch_coordinates_async(nxt) if (nxt is not None and device.type == "cuda") else None
        optimizer.zero_grad(set_to_none=True)
        loss, tb, disp = model_func(model, batch)
        loss.backward()
        if getattr(core, "grad_sync", None) is not None:
            core.grad_sync.finish()                 # head/backbone buckets are already in flight; stem bucket here
        if hasattr(optimizer, "clip_and_step"):
            optimizer.clip_and_step(clip)
        else:
            torch.nn.utils.clip_grad_norm_(params, clip)
            optimizer.step()
        scheduler.step()
        it += 1
        if not _FROZEN[0] and device.type == "cuda":
            # the model, the optimizer state and the cached tables are permanent: out of the cyclic collector's
            # generations after the first iteration, or every gen-2 pass walks them again (a 75 ms pause each)
            import gc
            gc.collect()
            gc.freeze()
            _FROZEN[0] = True
        if losses is not None:
            losses.append(float(loss.detach()))
        if rank == 0:
            log("epoch %d it %d lr %.2e loss %.4f (%s)" % (epoch, it, optimizer.param_groups[0]["lr"], float(loss.detach()),
                                                          ", ".join("%s %.3f" % (k, v) for k, v in sorted(tb.items()) if k.startswith("loss_"))))
    return it


def merge_eval_shards(det_annos, gt_annos, indices, n_total, group=None):
    """Collect every rank's per-scene results on all ranks and put them back in dataset order
    (pcdet/utils/common_utils.py:202-223 `merge_results_dist`, tools/eval_utils/eval_utils.py:44-48: the reference gathers
    through pickle files in a shared tmpdir, interleaves the ranks' lists and cuts the sampler's padding off; here the
    lists travel through the process group and every scene is placed by its own index, so the padding duplicates of
    `shard_ids` simply overwrite themselves).  EVERY rank must call this (a collective)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world <= 1:
        parts = [(list(indices), det_annos, gt_annos)]
    else:
        parts = [None] * world
        dist.all_gather_object(parts, (list(indices), det_annos, gt_annos), group=group)
    det, gt = [None] * n_total, [None] * n_total
    for idx, d, g in parts:
        assert len(idx) == len(d) == len(g), "a rank evaluated %d scenes for %d indices" % (len(d), len(idx))
        for i, di, gi in zip(idx, d, g):
            det[i], gt[i] = di, gi
    missing = [i for i in range(n_total) if det[i] is None]
    assert not missing, "scenes %s were evaluated by no rank" % missing[:8]
    return det, gt


@torch.no_grad()
def eval_one_epoch(model, dataset, class_names, device, metric=(0.25, 0.5), log=print, rank=0, world=1):
    """Detections of every scene of THIS rank's shard, merged over the ranks, -> indoor_eval (test.py / eval_utils.py /
    scannet_dataset.py:88-150).  With world > 1 every rank evaluates its own shard of `dataset` (scene i -> rank
    i mod W, as in training) and all of them must call this; the metric dict is returned on rank 0, None elsewhere."""
    model.eval()
    det_annos, gt_annos = [], []
    for batch in dataset.batches():
        batch["cur_epoch"] = 0
        gt_annos += SyntheticIndoorDataset.gt_annos(batch)
        load_data_to_gpu(batch, device)
        pred_dicts, _ = model(batch)
        for p in pred_dicts:
            det_annos.append({"boxes_3d": p["pred_boxes"].cpu().numpy(), "scores_3d": p["pred_scores"].cpu().numpy(),
                              "labels_3d": p["pred_labels"].cpu().numpy().astype(np.int64)})
    if world > 1:
        det_annos, gt_annos = merge_eval_shards(det_annos, gt_annos, dataset.eval_positions(), dataset.n_total())
        if rank != 0:
            return None
    return indoor_eval(gt_annos, det_annos, list(metric), {i: c for i, c in enumerate(class_names)},
                       logger=type("L", (), {"info": staticmethod(log)}))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="scannet")
    ap.add_argument("--config", default="S50k")
    ap.add_argument("--scenes", type=int, default=16)
    ap.add_argument("--data-root", default=None, help="processed ScanNet / SUN RGB-D folder (default: synthetic scenes)")
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--epochs", type=int, default=None)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--precision", choices=["bf16", "fp32"], default="fp32")
    ap.add_argument("--head-precision", choices=["split", "bf16", "fp32"], default="split",
                    help="with --precision bf16 (\"bf16 backbone\", BASELINE.json configs[1]): arithmetic of the two heads' convolutions -- "
                         "split = fp32-accurate products from three bf16 MFMA passes (me.PREC_SPLIT), fp32 = fp32 MFMA operands, bf16 = "
                         "bf16 operands in the heads too")
    ap.add_argument("--sync_bn", action="store_true", help="BatchNorm statistics over every rank's rows (reference tools/train.py:33,118-119)")
    args = ap.parse_args(argv)
    from . import me
    me.PRECISION = 1 if args.precision == "bf16" else 0
    me.HEAD_PRECISION = me.HEAD_MODES[args.head_precision] if args.precision == "bf16" else None
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    dev = torch.device(args.device, int(os.environ.get("LOCAL_RANK", "0"))) if args.device == "cuda" else torch.device("cpu")
    if dev.type == "cuda":
        from .hostpin import pin_host_threads
        pin_host_threads(int(os.environ.get("LOCAL_RANK", "0")))     # the launching thread stays on a few cores
    if world > 1:
        dist.init_process_group(backend="nccl" if dev.type == "cuda" else "gloo")
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    model, cfg = build_model.build_cagroup3d(args.dataset, seed=0)
    model = model.to(dev)
    if args.sync_bn and world > 1:
        # every nn.BatchNorm1d (the ones inside ME.MinkowskiBatchNorm included) becomes a SyncBatchNorm: me.fused_bn_act then
        # all-reduces its statistics tables (2 C floats forward, 2 C backward per layer); the RoI head's plain FC BatchNorms
        # run torch's SyncBatchNorm itself
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    oc = cfg.OPTIMIZATION
    epochs = args.epochs if args.epochs is not None else oc.NUM_EPOCHS
    bs = args.batch or oc.BATCH_SIZE_PER_GPU
    if args.data_root:
        ds = DiskIndoorDataset(args.dataset, args.data_root, cfg.CLASS_NAMES, bs, True, rank, world, args.workers)
    else:
        ds = SyntheticIndoorDataset(args.config, args.scenes, bs, rank, world)
    if dev.type == "cuda" and os.environ.get("CG3D_AUTOGRAD_ST", "1") != "0":
        torch.autograd.set_multithreading_enabled(False)      # backward nodes on the issuing thread (bench.py: -0.2 ms per step)
    optimizer = build_optimizer(model, oc)
    if hasattr(model, "split_late_parameters"):
        model.split_late_parameters(optimizer)
    start_epoch, it = 0, 0
    if args.resume:
        it, start_epoch = model.load_params_with_optimizer(args.resume, to_cpu=dev.type == "cpu", optimizer=optimizer)
    scheduler = build_scheduler(optimizer, len(ds), oc, last_it=it - 1 if it else -1)
    net = model
    if world > 1:
        # flat gradient buckets over RCCL sent from inside the backward pass (grad_sync.py; torch DDP's per-parameter
        # hooks cost this 432-parameter model 8 ms/step on one MI355X before a byte moves)
        from .grad_sync import TwoBucketGradSync
        model.grad_sync = TwoBucketGradSync(model)
    t0 = time.time()
    for epoch in range(start_epoch, epochs):
        it = train_one_epoch(net, optimizer, scheduler, ds, epoch, it, oc.GRAD_NORM_CLIP, rank)
        if args.ckpt and rank == 0:
            torch.save(checkpoint_state(net, optimizer, epoch + 1, it), args.ckpt)
    if rank == 0:
        print("trained %d iterations in %.1f s" % (it, time.time() - t0))
    result = None
    if args.eval:
        # every rank evaluates its shard of the validation set; the detections meet on rank 0 (tools/test.py with
        # --launcher pytorch: eval_utils.py:44-48)
        val = (DiskIndoorDataset(args.dataset, args.data_root, cfg.CLASS_NAMES, bs, False, rank, world, workers=args.workers)
               if args.data_root else SyntheticIndoorDataset(args.config, args.scenes, bs, rank, world))
        result = eval_one_epoch(model, val, cfg.CLASS_NAMES, dev, rank=rank, world=world)
    if world > 1:
        dist.barrier()                       # nobody tears the group down while a peer is still inside a collective
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
