"""CAGroup3D detector (mirror of pcdet/models/detectors/cagroup3d.py:8-157): colour normalisation,
voxelisation into a sparse tensor (hash build), backbone -> dense head -> RoI head, loss / prediction
packaging.  Same forward(batch_dict) contract as the reference."""
import torch

from .... import me as ME
from ...utils.common_utils import DeferredLog
from ..dense_heads.cagroup_head import split_gt_boxes
from .detector3d_template import Detector3DTemplate


class _Done:
    def __init__(self, value):
        self.value = value

    def result(self):
        return self.value


class _PrefetchWorker:
    """One daemon thread per detector; jobs are served in order, one result handle per job."""

    def __init__(self, detector, device):
        import queue
        import threading
        import weakref
        self.jobs = queue.Queue()
        det = weakref.ref(detector)          # the thread must not keep the model alive

        def loop():
            torch.cuda.set_device(device)    # the current device is per thread
            while True:
                job = self.jobs.get()
                if job is None:
                    return
                batch, handle = job
                d = det()
                try:
                    handle.value = d.prefetch_coordinates(batch) if d is not None else None
                except BaseException as e:  # noqa: BLE001  (handed to the thread that asks for the result)
                    handle.error = e
                finally:
                    del d
                    handle.done.set()
        self.thread = threading.Thread(target=loop, name=ME.PREFETCH_THREAD_NAME, daemon=True)
        self.thread.start()
        # the thread must be OUT of the HIP runtime and of torch before the interpreter tears them down (a daemon thread
        # still inside a launch at exit corrupted the heap: "corrupted size vs. prev_size" after the last bench line)
        import atexit
        atexit.register(self.close)

    def close(self):
        """Stop the worker: pending jobs are dropped (their handles report the shutdown), the sentinel is queued and the
        thread joined.  A worker that does not come back within the timeout is still inside a launch or a device wait --
        tearing the runtime down under it is what corrupted the heap -- so that is reported loudly instead of ignored."""
        if not self.thread.is_alive():
            return
        import queue
        try:
            while True:
                job = self.jobs.get_nowait()
                if job is not None:
                    job[1].error = RuntimeError("coordinate prefetch worker closed before this job ran")
                    job[1].done.set()
        except queue.Empty:
            pass
        self.jobs.put(None)
        self.thread.join(timeout=30)
        if self.thread.is_alive():
            import sys
            print("cagroup3d_amd: the coordinate-prefetch worker did not stop within 30 s (a hung device wait?); "
                  "the interpreter may not exit cleanly", file=sys.stderr, flush=True)

    def submit(self, batch):
        h = _Pending()
        self.jobs.put((batch, h))
        return h


class _Pending:
    def __init__(self):
        import threading
        self.done, self.value, self.error = threading.Event(), None, None

    def result(self):
        self.done.wait()
        if self.error is not None:
            raise self.error
        return self.value


class CAGroup3D(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset)
        self.module_list = self.build_networks()
        self.voxel_size = self.model_cfg.VOXEL_SIZE
        self.semantic_min_threshold = self.model_cfg.SEMANTIC_MIN_THR
        self.semantic_iter_value = self.model_cfg.SEMANTIC_ITER_VALUE
        self.semantic_value = self.model_cfg.SEMANTIC_THR

    def split_late_parameters(self, optimizer=None):
        """The class branches' parameters -- 107 of the detector's 126.5 M, the per-class 9^3 / 5^3 / transposed / fuse
        convolutions and their BatchNorms (`cagroup_head.py:227-282`) -- are first read behind the dense head's first blocking read,
        i.e. after the device-bound half of the step.  Their optimizer rows (optim.ClippedAdamW.set_early) and their bf16 / split
        copies (me.set_early_weights) are kept out of that half: deferred until `me.run_late()` (called by the head right after
        that read; me.LATE_MODE).  Whoever reads parameters outside a detector forward after an optimizer step calls
        `optimizer.finish_late()` first (train.checkpoint_state does)."""
        if not ME.LATE_WEIGHTS:
            return                       # the feature is off: no process-wide state is touched
        names = ("cls_individual_out", "cls_individual_expand_out", "cls_individual_up", "cls_individual_fuse")
        late = {id(p) for n, p in self.dense_head.named_parameters() if n.startswith(names)}
        early = [p for p in self.parameters() if id(p) not in late]
        ME.set_early_weights(early)
        if optimizer is not None and hasattr(optimizer, "set_early"):
            optimizer.set_early(early)

    def voxelization(self, points, prepared=None):
        """points (N,7) = (b,x,y,z,r,g,b) -> sparse tensor on the 0.02 m grid; one (the first) point's
        colour per voxel (cagroup3d.py:18-25).  `prepared`: the coordinate side from `prefetch_coordinates`."""
        if prepared is not None:
            mgr, key, uniq, n_in, event, keep, targets = prepared[:7]
            object.__setattr__(self, "_engine_program", prepared[7] if len(prepared) > 7 else None)
            assert n_in == points.shape[0], "prefetched coordinates belong to another batch"
            torch.cuda.current_stream().wait_event(event)
            ME.release_to_stream(mgr, [keep, targets], torch.cuda.current_stream())
            if len(prepared) > 7 and prepared[7] is not None:
                ME.record_cached(prepared[7].keep_cached, torch.cuda.current_stream())
            if targets is not None and self.training:
                object.__setattr__(self.dense_head, "_data_targets", targets.get("loss"))
                object.__setattr__(self.dense_head, "_forced_pre", targets.get("forced"))
            feats = points[:, 4:][uniq.long()]          # rows are in (batch, Morton) order: always re-index (a fresh tensor)
            return ME.SparseTensor(features=feats, coordinate_map_key=key, coordinate_manager=mgr)
        coordinates = points[:, :4].clone()
        coordinates[:, 1:] /= self.voxel_size
        return ME.SparseTensor(coordinates=coordinates, features=points[:, 4:].clone())

    def prefetch_coordinates(self, batch_dict):
        """Everything of a training / inference step that depends only on the point COORDINATES of a batch -- the
        voxel hash, every strided map, kernel map, pair list and segment table of the backbone, the per-scene row
        lists -- built on a side stream, with its data-dependent host reads, while the main stream is still busy (with
        the previous step's backward).  Pass the result as batch_dict['prepared'] to the forward of THAT batch.  The
        forward then has no host sync before the head, so the host runs ahead of the GPU through the backbone."""
        points = batch_dict["points"]
        if not points.is_cuda:
            return None
        if getattr(self, "_side_stream", None) is None:
            # normal priority on purpose: a high-priority side stream next to a stream pair that waits on each other
            # (main <-> the RCCL stream of any in-step collective) cost 8 ms/step on MI355X
            import os as _os
            prio = int(_os.environ.get("CG3D_SIDE_PRIORITY", "0"))
            self._side_stream = torch.cuda.Stream(device=points.device, priority=prio)
        side = self._side_stream
        with torch.cuda.stream(side), torch.no_grad():
            coordinates = points[:, :4].clone()
            coordinates[:, 1:] /= self.voxel_size
            ME.set_coords_only(True)
            program = None
            try:
                sp = ME.SparseTensor(coordinates=coordinates, features=points[:, 4:])
                out = None
                from .... import engine
                ME.set_coords_only(False)
                if engine.applicable(self.backbone_3d, compiling=True):
                    # the backbone's launch program for this batch (engine.py): compiling it builds every map, plan and table
                    # the pass needs -- it IS the dry run -- and the step then runs the backbone with one call each way
                    try:
                        gs = getattr(self.backbone_3d, "grad_sync", None)
                        feat = ME._fake(sp.F.shape[0], sp.F.shape[1], sp.F)
                        program = engine.compile_backbone(self.backbone_3d, sp._like(feat), mid_mark=bool(gs is not None and gs.mid))
                        m = sp.coordinate_manager.get(program.out_key)
                        out = ME.SparseTensor(features=ME._fake(m.n, self.backbone_3d.num_point_features, sp.F),
                                              coordinate_map_key=program.out_key, coordinate_manager=sp.coordinate_manager)
                    except engine.NotReady:
                        program = None
                ME.set_coords_only(True)
                if out is None:
                    out = self.backbone_3d({"sp_tensor": sp, "batch_size": batch_dict["batch_size"]})["sp_tensor"]
                _ = out.decomposition_permutations            # the head's first host read
            finally:
                ME.set_coords_only(False)
            targets = None
            if self.training and "gt_boxes" in batch_dict and getattr(self.dense_head, "batched", False):
                # training targets that depend on the data only (semantic labels, vote targets, the bench's forced mask)
                head, dev, bs = self.dense_head, points.device, batch_dict["batch_size"]
                gt_b, gt_l = split_gt_boxes(batch_dict["gt_boxes"], torch.long)
                if all(len(g) > 0 for g in gt_b) and "instance_mask" in batch_dict:
                    def masks(key):
                        return [x.to(dev) if torch.is_tensor(x) else torch.from_numpy(x).to(dev) for x in batch_dict[key]]
                    targets = {"loss": head.data_targets(out.C, gt_b, gt_l, self.convert2list(points, bs),
                                                         masks("semantic_mask"), masks("instance_mask"))}
                    if head.force_gt_selection:
                        targets["forced"] = head._forced_selection(batch_dict, out, out.C[:, 1:].float() * head.voxel_size)
            event = torch.cuda.Event()
            event.record(side)
        return (sp.coordinate_manager, sp.coordinate_map_key, sp.unique_index, points.shape[0], event,
                [sp.unique_index, sp.inverse_mapping, program.keep if program is not None else None], targets, program)

    def prefetch_coordinates_async(self, batch_dict):
        """`prefetch_coordinates` on a worker thread: returns a handle whose `.result()` is what `prefetch_coordinates`
        returns.  What the dry run may touch (and nothing else): the backbone's modules READ-ONLY (no parameter, buffer or
        attribute is written: `COORDS_ONLY` is a per-thread flag and BatchNorm / convolutions return placeholders under it),
        `dense_head.data_targets` / `_forced_selection` (pure functions of the batch), its own side stream, and the
        host-side caches of `me.py` (`_offset_cache`, `_chunk_cache`, `_ident_cache` ...: every get-or-build and every
        size-triggered `clear()` runs under `me._CACHE_LOCK`, and the one cache both threads read -- the kernel-offset tables --
        publishes an entry only after its upload has completed on the building thread's stream) and the pinned staging rings
        (keyed per stream).  A worker exception surfaces at `.result()`, i.e. one step late.  Submit the NEXT batch before starting the current step: the dry run is Python glue, short launches and
        ~30 host reads of device counters (each a wait for the side stream); on its own thread those waits and every
        GIL-free stretch (launches, ATen calls, the autograd engine's C++ side) overlap with the issue path of the
        current step, which is what bounds the step once the kernels are fast (host path)."""
        if batch_dict is None or not batch_dict["points"].is_cuda:
            return _Done(None)
        w = getattr(self, "_prefetch_worker", None)
        if w is None or not w.thread.is_alive():
            w = self._prefetch_worker = _PrefetchWorker(self, batch_dict["points"].device)
        return w.submit(batch_dict)

    def forward(self, batch_dict):
        cur_epoch = batch_dict.get("cur_epoch", None)
        assert cur_epoch is not None
        ME._ROWS16.clear()
        ME._ROWS48.clear()
        ME._STATS.clear()
        if not self.training:
            ME.run_late()                           # (an optimizer step's late rows may still be deferred: evaluation reads everything)
        ME.zero_arena().reset()                     # a fresh zero block for this step's statistics tables
        ME.WANT_BN_STATS = bool(self.training)      # evaluation: no BatchNorm takes the conv epilogue's partial sums
        # the bf16 copies of every conv weight in one launch; the layers of THIS forward take them from the arena
        ME.prepare_weights(self.training, split=True)   # (late rows: ME.late_weights_ready() behind the backbone, below)
        try:
            return self._forward(batch_dict, cur_epoch)
        finally:
            ME.finish_weights()
            ME._STATS.clear()                       # partial sums no BatchNorm asked for must not keep conv outputs alive
            ME.WANT_BN_STATS = True

    def _forward(self, batch_dict, cur_epoch):
        object.__setattr__(self.module_list[1], "semantic_threshold",
                           max(self.semantic_value - int(cur_epoch) * self.semantic_iter_value, self.semantic_min_threshold))
        batch_dict["points"][:, -3:] = batch_dict["points"][:, -3:] / 255.
        object.__setattr__(self, "_engine_program", None)
        batch_dict["sp_tensor"] = self.voxelization(batch_dict["points"], batch_dict.pop("prepared", None))
        batch_dict["engine_program"] = self._engine_program          # the backbone's launch program compiled by the dry run
        for i, module in enumerate(self.module_list):
            # the two heads (and their losses) under ME.HEAD_PRECISION when that is set ("bf16 backbone", fp32 heads)
            with ME.precision_scope(ME.HEAD_PRECISION if i > 0 else None):
                batch_dict.update(module(batch_dict))
            if i == 0:
                ME.late_weights_ready()                 # (stream mode: the late stream's work is behind us; defer mode: see head_stage.class_rows)
            if i == 0 and self.training:
                # the per-scene views of the raw points (the losses' scene_points) cost one host read of the scene sizes.  The
                # dense head takes it where its own first blocking read is -- after the backbone AND its coordinate-independent
                # layers have been issued, the host ahead of the device -- instead of between the heads and the backward pass,
                # where every blocking read leaves the device idle until the next launch arrives
                def early_reads(bd=batch_dict):
                    bd["scene_points"] = self.convert2list(bd["points"], bd["batch_size"])
                batch_dict["early_host_reads"] = early_reads
            if i == 0 and self.training and getattr(self, "grad_sync", None) is not None:
                self.grad_sync.attach(batch_dict["sp_tensor"].F)      # heads' gradients are complete when this one is
        if self.training:
            with ME.precision_scope(ME.HEAD_PRECISION):
                loss, tb_dict, disp_dict = self.get_training_loss(batch_dict)
            disp_dict["cur_semantic_value"] = self.module_list[1].semantic_threshold
            return {"loss": loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def post_processing(self, batch_dict):
        pred_dicts = [{"pred_boxes": batch_dict["batch_box_preds"][i], "pred_scores": batch_dict["batch_score_preds"][i],
                       "pred_labels": batch_dict["batch_cls_preds"][i]} for i in range(batch_dict["batch_size"])]
        recall_dict = {}
        if "gt_boxes" in batch_dict:
            recall_dict = {"gt": 0}
            for t in self.model_cfg.POST_PROCESSING.RECALL_THRESH_LIST:
                recall_dict["roi_%s" % str(t)] = 0
                recall_dict["rcnn_%s" % str(t)] = 0
        return pred_dicts, recall_dict

    @staticmethod
    def convert2list(points, batch_size=None):
        """(N, 1 + C) rows with a leading scene index -> per-scene (n_i, C) tensors (reference cagroup3d.py:76-80).  Collated
        batches are scene-major, so the scenes are row RANGES: views, and one host read for their sizes instead of one
        boolean-mask selection (a host sync each) per scene; a column that is not sorted takes the reference's masks."""
        if batch_size is None:
            batch_size = int(points[:, 0].max().int()) + 1
        if points.is_cuda and points.shape[0] > 0:
            counts = ME.sorted_batch_counts(points[:, 0], batch_size)
            if counts is not None:
                out, r0 = [], 0
                for c in counts:
                    out.append(points[r0:r0 + c, 1:])
                    r0 += c
                return out
        return [points[points[:, 0] == i, 1:] for i in range(batch_size)]

    def get_training_loss(self, batch_dict):
        bs = batch_dict["batch_size"]
        dev = batch_dict["points"].device

        def masks(key):
            if key not in batch_dict:
                return None
            return [x.to(dev) if torch.is_tensor(x) else torch.from_numpy(x).to(dev) for x in batch_dict[key]]
        if batch_dict.get("gt_bboxes_3d", None) is None:
            batch_dict["gt_bboxes_3d"], batch_dict["gt_labels_3d"] = split_gt_boxes(batch_dict["gt_boxes"], torch.long)
        x, semantic_scores, voxel_offset = batch_dict["one_stage_results"]
        centernesses, bbox_preds, cls_scores, voxel_points = x
        loss_one, tb_dict = self.dense_head.loss(
            centernesses, bbox_preds, cls_scores, voxel_points, semantic_scores, voxel_offset,
            batch_dict["gt_bboxes_3d"], batch_dict["gt_labels_3d"],
            batch_dict["scene_points"] if batch_dict.get("scene_points") is not None else self.convert2list(batch_dict["points"], bs),
            [None] * bs, masks("semantic_mask"), masks("instance_mask"))
        loss_two, tb_two = self.roi_head.loss(batch_dict)
        loss_all = loss_one + loss_two
        # every number of the log stays on the device until somebody reads the dict (common_utils.DeferredLog)
        tb_all = DeferredLog(("loss_all",), loss_all.view(1)).absorb(tb_dict).absorb(tb_two)
        disp_dict = DeferredLog().absorb(tb_dict).absorb(tb_two)
        return loss_all, tb_all, disp_dict
