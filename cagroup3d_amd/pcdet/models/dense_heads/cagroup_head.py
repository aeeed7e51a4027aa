"""CAGroup3DHead: semantic + vote branch, class-aware local regrouping, proposal decode + NMS,
stage-1 losses (mirror of pcdet/models/dense_heads/cagroup_head.py:14-797) on the gfx950 engine.

Data flow of one class branch: SURVEY.md Appendix B."""
import collections.abc

import numpy as np
import torch
from torch import nn

from .... import me as ME
from ....ops.iou3d_nms_utils import nms_batched_sorted, nms_gpu, nms_normal_gpu
from ....ops.knn import knn
from ...config import AttrDict
from ...utils.common_utils import DeferredLog
from ...utils.iou3d_loss import IoU3DLoss
from ...utils.loss_utils import CrossEntropy, FocalLoss, SmoothL1Loss
from ..model_utils.cagroup_utils import Scale, bias_init_with_prob, parse_params, reduce_mean
from .target_assigner.cagroup3d_assigner import CAGroup3DAssigner, find_points_in_boxes

# mean box size per class (the reference hard-codes them, cagroup_head.py:75-104); the class voxel
# size is half of it clipped to [0.04, 1.0] m (:105-106)
SCANNET_CLASS_SIZES = [[0.2309, 0.2435, 0.2777], [0.5631, 0.5528, 0.3579], [0.1840, 0.1845, 0.2155],
                       [0.4187, 0.4536, 0.2503], [0.2938, 0.3203, 0.1899], [0.1595, 0.1787, 0.5250],
                       [0.2887, 0.2174, 0.3445], [0.2497, 0.3147, 0.5063], [0.0634, 0.1262, 0.1612],
                       [0.4332, 0.5691, 0.0810], [0.3088, 0.4212, 0.2627], [0.4130, 0.1966, 0.5044],
                       [0.1995, 0.2133, 0.3897], [0.1260, 0.1137, 0.5254], [0.1781, 0.1774, 0.2218],
                       [0.1526, 0.1520, 0.0904], [0.3453, 0.3164, 0.1491], [0.1426, 0.1477, 0.1741]]
SUNRGBD_CLASS_SIZES = [[0.6343, 0.4861, 0.2782], [0.2373, 0.3839, 0.2155], [0.2771, 0.5602, 0.2536],
                       [0.1776, 0.1659, 0.2482], [0.2097, 0.1363, 0.2269], [0.2086, 0.4039, 0.2209],
                       [0.1586, 0.3008, 0.3519], [0.1502, 0.1896, 0.2050], [0.1214, 0.3213, 0.5067],
                       [0.2298, 0.4195, 0.1418]]


FUSED_LOSSES = __import__("os").environ.get("CG3D_FUSED_LOSSES", "1") != "0"
FUSED_HEAD = __import__("os").environ.get("CG3D_FUSED_HEAD", "1") != "0"


def _conv_bn_elu(cin, cout, k):
    return ME.Sequential(ME.MinkowskiConvolution(cin, cout, kernel_size=k, dimension=3),
                         ME.MinkowskiBatchNorm(cout), ME.MinkowskiELU())


import time as _time
TICKS = None         # list -> (name, host s, after-sync s) records


def _tick(name):
    if TICKS is not None:
        a = _time.perf_counter()
        torch.cuda.synchronize()
        TICKS.append((name, a, _time.perf_counter()))


class _LazyClassLists:
    """The class branches' outputs as the reference returns them -- four lists (centre-ness, box, class score, points) of
    n_classes lists of one tensor per scene -- cut from the (class, scene)-major merged rows on first use."""

    def __init__(self, merged, per_scene, n_classes, n_batch):
        self._args, self._lists = (list(merged), list(per_scene), n_classes, n_batch), None

    def lists(self):
        if self._lists is None:
            merged, per_scene, C, B = self._args
            pieces = [torch.split(t, per_scene) for t in merged]
            self._lists = [[list(pieces[j][c * B:(c + 1) * B]) for c in range(C)] for j in range(4)]
        return self._lists

    def view(self, j):
        return _LazyClassList(self, j)


class _LazyClassList(collections.abc.Sequence):
    def __init__(self, owner, j):
        self._owner, self._j = owner, j

    def __getitem__(self, i):
        return self._owner.lists()[self._j][i]

    def __len__(self):
        return self._owner._args[2]


class CAGroup3DHead(nn.Module):
    def __init__(self, model_cfg, yaw_parametrization="fcaf3d", predict_boxes=True, **kwargs):
        super().__init__()
        cfg = model_cfg
        self.n_classes = n_classes = cfg.N_CLASSES
        out_channels = cfg.OUT_CHANNELS
        n_reg_outs = cfg.N_REG_OUTS
        self.voxel_size = cfg.VOXEL_SIZE
        self.semantic_threshold = cfg.SEMANTIC_THR
        self.expand = cfg.EXPAND_RATIO
        self.with_yaw = cfg.WITH_YAW
        self.use_sem_score = cfg.USE_SEM_SCORE
        self.cls_kernel = cfg.CLS_KERNEL
        self.yaw_parametrization = yaw_parametrization
        self.predict_boxes = predict_boxes
        self.gt_per_seed = 3  # SUN RGB-D only
        # benchmark aid (SURVEY.md 8(d) "forced-selection"): an untrained net selects ~nothing per class;
        # when set, class c selects the voxels inside its GT boxes so the class branches see trained-like sizes
        self.force_gt_selection = False
        self.force_class_logit_boost = 0.0   # added to class c's own logit on class c's map (trained-like scores)
        # batched = all class branches share one coordinate space (class folded into the batch index) and
        # run as grouped launches; False = the reference's 18-iteration loop (kept for the equivalence test)
        self.batched = True
        self._data_targets = None        # data-only targets / forced mask handed over by CAGroup3D.prefetch_coordinates
        self._forced_pre = None

        def sub(name, default):
            return cfg.get(name, AttrDict(default))
        self.assigner = CAGroup3DAssigner(cfg.ASSIGNER)
        self.loss_centerness = CrossEntropy(**parse_params(sub("LOSS_CENTERNESS", dict(NAME="CrossEntropyLoss", USE_SIGMOID=True, LOSS_WEIGHT=1.0))))
        self.loss_bbox = IoU3DLoss(**parse_params(sub("LOSS_BBOX", dict(NAME="IoU3DLoss", LOSS_WEIGHT=1.0))))
        focal = dict(NAME="FocalLoss", USE_SIGMOID=True, GAMMA=2.0, ALPHA=0.25, LOSS_WEIGHT=1.0)
        self.loss_cls = FocalLoss(**parse_params(sub("LOSS_CLS", focal)))
        self.loss_sem = FocalLoss(**parse_params(sub("LOSS_SEM", focal)))
        self.loss_offset = SmoothL1Loss(**parse_params(sub("LOSS_OFFSET", dict(NAME="SmoothL1Loss", BETA=0.04, REDUCTION="sum", LOSS_WEIGHT=1.0))))
        self.nms_cfg = sub("NMS_CONFIG", dict(SCORE_THR=0.01, NMS_PRE=1000, IOU_THR=0.5))

        sizes = SCANNET_CLASS_SIZES if n_classes == 18 else SUNRGBD_CLASS_SIZES
        self.voxel_size_list = np.clip(np.array(sizes) / 2., 0.04, 1.0).tolist()

        c = out_channels
        n_vote = 3 if self.with_yaw else 1
        self.offset_block = ME.Sequential(
            ME.MinkowskiConvolution(c, c, kernel_size=1, dimension=3), ME.MinkowskiBatchNorm(c), ME.MinkowskiELU(),
            ME.MinkowskiConvolution(c, c, kernel_size=1, dimension=3), ME.MinkowskiBatchNorm(c), ME.MinkowskiELU(),
            ME.MinkowskiConvolution(c, 3 * n_vote, kernel_size=1, dimension=3))
        self.feature_offset = _conv_bn_elu(c, c * n_vote, 3)
        self.semantic_conv = ME.MinkowskiConvolution(c, n_classes, kernel_size=1, bias=True, dimension=3)
        self.centerness_conv = ME.MinkowskiConvolution(c, 1, kernel_size=1, dimension=3)
        self.reg_conv = ME.MinkowskiConvolution(c, n_reg_outs, kernel_size=1, dimension=3)
        self.cls_conv = ME.MinkowskiConvolution(c, n_classes, kernel_size=1, bias=True, dimension=3)
        self.scales = nn.ModuleList([Scale(1.) for _ in range(n_classes)])
        self.cls_individual_out = nn.ModuleList([_conv_bn_elu(c, c, self.cls_kernel) for _ in range(n_classes)])
        self.cls_individual_up = nn.ModuleList([nn.ModuleList([
            ME.MinkowskiGenerativeConvolutionTranspose(c, c, kernel_size=self.expand, stride=self.expand, dimension=3),
            ME.Sequential(ME.MinkowskiBatchNorm(c), ME.MinkowskiELU())]) for _ in range(n_classes)])
        self.cls_individual_fuse = nn.ModuleList([_conv_bn_elu(c * 2, c, 1) for _ in range(n_classes)])
        self.cls_individual_expand_out = nn.ModuleList([_conv_bn_elu(c, c, 5) for _ in range(n_classes)])
        self.init_weights()

    def init_weights(self):
        """(cagroup_head.py:190-198)"""
        for conv in (self.centerness_conv, self.reg_conv, self.cls_conv, self.semantic_conv):
            nn.init.normal_(conv.kernel, std=.01)
        nn.init.constant_(self.cls_conv.bias, bias_init_with_prob(.01))
        nn.init.constant_(self.semantic_conv.bias, bias_init_with_prob(.01))
        for blk in self.cls_individual_out:
            nn.init.normal_(blk[0].kernel, std=.01)

    # ------------------------------------------------------------------ forward
    def forward(self, input_dict, return_middle_feature=True):
        batch_size = input_dict["batch_size"]
        out = input_dict["sp_tensor"]
        semantic_scores = self.semantic_conv(out)
        ts = out.coordinate_map_key.get_key()[0][0]
        pre = None
        if FUSED_HEAD:
            from .... import engine
            if engine.head_pre_applicable(self):
                # the vote-offset block and the feature-offset block as one launch program each way (engine.compile_head_pre):
                # two foreign calls and one autograd node for seven layers
                try:
                    pre = engine.run_head_pre(self, out)
                except engine.NotReady:
                    pre = None
        if pre is not None:
            voxel_offsets = ME.SparseTensor(features=pre[0], coordinate_map_key=out.coordinate_map_key,
                                            coordinate_manager=out.coordinate_manager)
            offset_features = pre[1]
        else:
            voxel_offsets = self.offset_block(out)
            offset_features = self.feature_offset(out).F
        n_vote = 3 if self.with_yaw else 1
        offset_features = offset_features.view(offset_features.shape[0], n_vote, -1)
        sem_prob = semantic_scores.F.detach().sigmoid()
        forced = None
        if self.force_gt_selection:
            forced = self._forced_pre
            object.__setattr__(self, "_forced_pre", None)     # (plain attribute: skip nn.Module.__setattr__'s registration checks)
            if forced is None or forced.shape[0] != out.F.shape[0]:
                forced = self._forced_selection(input_dict, out, out.C[:, 1:].float() * self.voxel_size)

        # the ground-truth lists need one host read of the padding mask (and the detector's scene sizes another): taken right
        # behind the class-row counting launch, inside the step's first wait for the device -- not after the class branches,
        # where every blocking read leaves the device idle until the next launch arrives
        gt_holder = []

        def early_reads():
            if self.predict_boxes and "gt_boxes" in input_dict and "gt_bboxes_3d" not in input_dict:
                gt_holder.append(split_gt_boxes(input_dict["gt_boxes"], torch.int))
            early = input_dict.pop("early_host_reads", None)    # (the detector's own data-only host reads: same place, same reason)
            if early is not None:
                early()
        object.__setattr__(self, "_merged", None)
        fused = self.batched and FUSED_HEAD and out.F.shape[0] > 0
        if fused:
            # selection, pad voxels, voted positions (with the scene-bound clamp) and both quantisations: ops/head_stage.class_rows
            outs = self._class_branches_batched(out, sem_prob, forced, None, None, None, None, offset_features, n_vote, batch_size,
                                                votes=voxel_offsets.F.detach(), ts=ts, before_read=early_reads)
        else:
            early_reads()
            pad_id = torch.stack([p[0] for p in semantic_scores.decomposition_permutations]).long()  # first row of every scene
            xyz_vox = out.C[:, 1:]
            # (column reductions of the strided [N, 3] view run one workgroup chain per column: 100 + 52 us on 156 k rows;
            # the same numbers as row reductions of a contiguous [3, N] copy: 3 x 5 us)
            xyz_t = xyz_vox.t().contiguous()
            max_bound = (xyz_t.amax(1) + ts) * self.voxel_size
            min_bound = (xyz_t.amin(1) - ts) * self.voxel_size
            ori_xyz = xyz_vox.float() * self.voxel_size
            voted = ori_xyz.view(-1, 1, 3) + voxel_offsets.F.detach().view(-1, n_vote, 3)
            voted = torch.max(torch.min(voted, max_bound.view(1, 1, 3)), min_bound.view(1, 1, 3))
            batch_col = out.C[:, :1].float()
            branch = self._class_branches_batched if self.batched else self._class_branches_loop
            outs = branch(out, sem_prob, forced, pad_id, batch_col, voted, ori_xyz, offset_features, n_vote, batch_size)
        if isinstance(outs, _LazyClassLists):
            centernesses, bbox_preds, cls_scores, voxel_points = (outs.view(j) for j in range(4))
        else:
            centernesses, bbox_preds, cls_scores, voxel_points = [list(x) for x in zip(*outs)]
        out_dict = {"one_stage_results": [[centernesses, bbox_preds, cls_scores, voxel_points], semantic_scores, voxel_offsets],
                    "middle_feature_list": [None, None, None, out] if return_middle_feature else None}
        if self.predict_boxes:
            object.__setattr__(self, "_flat_props", None)
            if self._merged is not None and not self.use_sem_score and \
                    not (self.training and self.nms_cfg.get("SCORE_THR_AGNOSTIC", None) is not None):
                out_dict["pred_bbox_list"] = self.get_bboxes_batched(self._merged, batch_size)
                out_dict["pred_bbox_flat"] = self._flat_props
            else:
                out_dict["pred_bbox_list"] = self.get_bboxes(centernesses, bbox_preds, cls_scores, voxel_points,
                                                             [None] * batch_size, rescale=False)
            if gt_holder:
                out_dict["gt_bboxes_3d"], out_dict["gt_labels_3d"] = gt_holder[0]
        return out_dict

    def _vs_table(self, device):
        """Per-class voxel sizes [C,3] on `device` (cached: building it from a Python list is a blocking copy)."""
        key = str(device)
        if getattr(self, "_vs_cache", None) is None or self._vs_cache[0] != key:
            self._vs_cache = (key, ME.h2d(self.voxel_size_list, torch.float32, device))
        return self._vs_cache[1]

    def _forced_selection(self, input_dict, out, ori_xyz):
        """bool [N, n_classes]: voxel lies inside a GT box of that class (all scenes in one pass)."""
        gt = input_dict["gt_boxes"]                                    # [B, Gmax, 8], zero-padded
        B, Gmax = gt.shape[0], gt.shape[1]
        flat = gt.reshape(B * Gmax, -1)
        real = ~(flat == 0).all(dim=-1)
        box_scene = torch.arange(B, device=gt.device).repeat_interleave(Gmax)
        inside = find_points_in_boxes(ori_xyz, flat[:, :7], point_seg=out.C[:, 0], box_seg=box_scene)   # padded rows: zero size
        inside = inside & real.view(1, -1)
        onehot = torch.nn.functional.one_hot(flat[:, 7].long().clamp(0, self.n_classes - 1), self.n_classes).float()
        return (inside.float() @ onehot) > 0

    def _class_branches_loop(self, out, sem_prob, forced, pad_id, batch_col, voted, ori_xyz, offset_features, n_vote,
                             batch_size):
        """The reference's per-class loop (cagroup_head.py:227-282), one class at a time."""
        outs = []
        for cls_id in range(self.n_classes):
            with torch.no_grad():
                hit = sem_prob[:, cls_id] > self.semantic_threshold
                if forced is not None:
                    hit = forced[:, cls_id]          # the forced mode REPLACES the net's selection (sizes independent of the weights)
                sel = torch.nonzero(hit).squeeze(1)
                sel = torch.cat([sel, pad_id])
            b = batch_col[sel]
            vote_rows = torch.cat([b.view(-1, 1, 1).expand(-1, n_vote, 1), voted[sel]], dim=2).reshape(-1, 4)
            fuse_xyz = torch.cat([vote_rows, torch.cat([b, ori_xyz[sel]], dim=1)], dim=0)       # (n_vote+1)*n rows
            fuse_feat = torch.cat([offset_features[sel].reshape(-1, offset_features.shape[-1]), out.F[sel]], dim=0)

            vsize = fuse_xyz.new_tensor(self.voxel_size_list[cls_id])
            fine = fuse_xyz.clone()
            fine[:, 1:] = torch.floor(fuse_xyz[:, 1:] / vsize)
            cls_map = ME.SparseTensor(coordinates=fine, features=fuse_feat,
                                      quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
            fine_C = cls_map.C
            cls_map = self.cls_individual_out[cls_id](cls_map)

            coarse = fuse_xyz.clone()
            coarse[:, 1:] = torch.floor(fuse_xyz[:, 1:] / (vsize * self.expand)) * self.expand
            cls_exp = ME.SparseTensor(coordinates=coarse, features=fuse_feat, tensor_stride=self.expand,
                                      quantization_mode=ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
            cls_exp = self.cls_individual_expand_out[cls_id](cls_exp)
            up = self.cls_individual_up[cls_id][0](cls_exp, fine_C)
            up = self.cls_individual_up[cls_id][1](up)
            fused = ME.SparseTensor(coordinates=fine_C, features=torch.cat([up.F, cls_map.F], dim=-1))
            fused = self.cls_individual_fuse[cls_id](fused)
            outs.append(self.forward_single(fused, self.scales[cls_id], self.voxel_size_list[cls_id], cls_id))

        return outs

    @staticmethod
    def _grouped_bn_act(feats, bounds, bns, act):
        """Per-class BatchNorm + ELU over contiguous row groups: one fused launch pair for all classes."""
        return ME.fused_bn_act(feats, [b.bn for b in bns], bounds, ME.ACT_ELU)

    @torch.no_grad()
    def _class_rows_torch(self, hit, pad_id, batch_col, voted, ori_xyz, n_vote, B):
        """The class-major row construction as tensor expressions (the form the fused stage op replaces and is tested
        against): -> (src int64 [T] rows of the [votes ; originals] table, fine / coarse float [T,4] quantised coordinates)."""
        N, C = hit.shape
        dev = hit.device
        vs_tab = self._vs_table(dev)
        sel_cls, sel_row = torch.nonzero(hit.t(), as_tuple=True)                 # class-major, rows ascending
        ar_c = torch.arange(C, device=dev)
        all_cls = torch.cat([sel_cls, ar_c.repeat_interleave(B)])
        all_pos = torch.cat([sel_row, N + torch.arange(B, device=dev).repeat(C)])  # pads go last in a class
        all_row = torch.cat([sel_row, pad_id.repeat(C)])
        order = torch.argsort(all_cls * (N + B) + all_pos)
        e_cls, e_row = all_cls[order], all_row[order]
        E = e_cls.shape[0]
        n_c = ME.count_ids(e_cls, C)
        start_c = torch.cumsum(n_c, 0) - n_c
        j = torch.arange(E, device=dev) - start_c[e_cls]
        base = start_c[e_cls] * (n_vote + 1)
        dest_vote = (base + j * n_vote).unsqueeze(1) + torch.arange(n_vote, device=dev).unsqueeze(0)
        dest_ori = base + n_c[e_cls] * n_vote + j
        total = E * (n_vote + 1)
        src = torch.empty(total, dtype=torch.long, device=dev)          # row of the [votes ; originals] table
        src[dest_vote.reshape(-1)] = (e_row.unsqueeze(1) * n_vote + torch.arange(n_vote, device=dev).unsqueeze(0)).reshape(-1)
        src[dest_ori] = N * n_vote + e_row
        row_cls = torch.empty(total, dtype=torch.long, device=dev)
        row_cls[dest_vote.reshape(-1)] = e_cls.unsqueeze(1).expand(-1, n_vote).reshape(-1)
        row_cls[dest_ori] = e_cls
        src_vox = torch.where(src < N * n_vote, src // n_vote, src - N * n_vote)      # backbone row of each fused row
        xyz_tab = torch.cat([voted.reshape(-1, 3), ori_xyz], dim=0)
        fuse_xyz = xyz_tab[src]
        bprime = row_cls.float() * B + batch_col[src_vox, 0]
        vs = vs_tab[row_cls]
        fine = torch.cat([bprime.unsqueeze(1), torch.floor(fuse_xyz / vs)], dim=1)
        coarse = torch.cat([bprime.unsqueeze(1), torch.floor(fuse_xyz / (vs * self.expand)) * self.expand], dim=1)

        return src, fine, coarse

    def _class_branches_batched(self, out, sem_prob, forced, pad_id, batch_col, voted, ori_xyz, offset_features, n_vote,
                                batch_size, votes=None, ts=None, before_read=None):
        """All class branches at once: rows of class c live at batch index c*B + b of ONE coordinate map,
        class-major, so every hash build / kernel map / convolution / NMS is a single (grouped) launch.
        Row order inside a class equals the loop version's, so results match it up to fp32 summation order.
        votes / ts given: the row construction runs as the fused stage op (ops/head_stage.class_rows) on the predicted votes
        themselves; otherwise as the tensor expressions below on the precomputed voted positions."""
        C, B, dev = self.n_classes, batch_size, out.F.device
        N, ch = out.F.shape
        elu = torch.nn.functional.elu
        vs_tab = self._vs_table(dev)
        _tick("enter")
        if votes is not None:
            from ....ops import head_stage as HS
            with torch.no_grad():
                hit = forced if forced is not None else sem_prob > self.semantic_threshold
                starts = out.batch_row_starts(B)
                if starts is None:
                    pad_row = torch.stack([p[0] for p in out.decomposition_permutations]).to(torch.int32)
                else:
                    pad_row = ME.h2d(starts, torch.int32, dev)
                src, fine, coarse, _ = HS.class_rows(hit, out.C, pad_row, votes, n_vote, self.voxel_size, int(ts),
                                                     vs_tab, self.expand, B, before_read=before_read)
            fuse_feat = HS.gather_rows2(offset_features.reshape(N * n_vote, -1), out.F, src)
        else:
            with torch.no_grad():
                hit = sem_prob > self.semantic_threshold
                if forced is not None:
                    hit = forced                         # the forced mode REPLACES the net's selection (sizes independent of the weights)
                src, fine, coarse = self._class_rows_torch(hit, pad_id, batch_col, voted, ori_xyz, n_vote, B)
            feat_tab = torch.cat([offset_features.reshape(N * n_vote, -1), out.F], dim=0)
            fuse_feat = ME.gather_rows(feat_tab, src)

        _tick("class_rows+gather")
        avg = ME.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE
        cls_map, cls_exp = ME.SparseTensor.build_many([          # (one host read for the two maps' row counts)
            dict(coordinates=fine, features=fuse_feat, quantization_mode=avg),
            dict(coordinates=coarse, features=fuse_feat, tensor_stride=self.expand, quantization_mode=avg)])
        _tick("fine + coarse tensor")
        fine_C = cls_map.C
        with torch.no_grad():                                             # ONE host read for all group sizes
            fb, cb = fine_C[:, 0].long(), cls_exp.C[:, 0].long()
            per = ME.count_ids(fb, C * B)                                 # (class, scene) sizes; a class's size is their sum
            sizes = torch.cat([per.view(C, B).sum(1), ME.count_ids(cb // B, C), per]).cpu().numpy()
        fine_bounds = (0,) + tuple(np.cumsum(sizes[:C]).tolist())
        coarse_bounds = (0,) + tuple(np.cumsum(sizes[C:2 * C]).tolist())
        per_scene = sizes[2 * C:].tolist()
        _tick("sizes")

        mgr, emgr = cls_map.coordinate_manager, cls_exp.coordinate_manager
        km9 = mgr.kernel_map(cls_map.coordinate_map_key, cls_map.coordinate_map_key, self.cls_kernel, 1, False)
        km5 = emgr.kernel_map(cls_exp.coordinate_map_key, cls_exp.coordinate_map_key, 5, 1, False)
        tgt_key, _, _ = emgr.insert(fine_C, 1, assume_unique=True)         # generative transposed conv onto the fine voxels (rows of cls_map: distinct, no host read)
        km_up = emgr.kernel_map(cls_exp.coordinate_map_key, tgt_key, self.expand, 1, True)
        ident = ME.KernelMap.identity(fine_C.shape[0], dev)
        _tick("maps")
        f = None
        from .... import engine
        if engine.class_branches_applicable(self):
            # the four (convolution, BatchNorm, ELU) stages of all classes as one launch program each way (engine.py): one
            # autograd node and two foreign calls instead of ~16 nodes and ~60 calls issued from here
            try:
                f = engine.run_class_branches(self, cls_map.F, cls_exp.F, km9, km5, km_up, ident, fine_bounds, coarse_bounds)
            except engine.NotReady:
                f = None
        _tick("program")
        if f is None:
            a = ME.grouped_conv(cls_map.F, [m[0].kernel for m in self.cls_individual_out], km9, fine_bounds, closed=True)
            a = self._grouped_bn_act(a, fine_bounds, [m[1] for m in self.cls_individual_out], elu)
            e = ME.grouped_conv(cls_exp.F, [m[0].kernel for m in self.cls_individual_expand_out], km5, coarse_bounds, closed=True)
            e = self._grouped_bn_act(e, coarse_bounds, [m[1] for m in self.cls_individual_expand_out], elu)
            u = ME.grouped_conv(e, [m[0].kernel for m in self.cls_individual_up], km_up, fine_bounds)
            u = self._grouped_bn_act(u, fine_bounds, [m[1][0] for m in self.cls_individual_up], elu)
            f = ME.grouped_conv(torch.cat([u, a], dim=1), [m[0].kernel for m in self.cls_individual_fuse], ident, fine_bounds)
            f = self._grouped_bn_act(f, fine_bounds, [m[1] for m in self.cls_individual_fuse], elu)
        _tick("layers")

        with torch.no_grad():
            rc = fb // B                                                   # class of every fine row
        centerness = ME.linear(f, self.centerness_conv.kernel)
        cls_score = ME.linear(f, self.cls_conv.kernel, self.cls_conv.bias.view(-1))
        reg = ME.linear(f, self.reg_conv.kernel)
        scale_vec = torch.stack([sc.scale for sc in self.scales])
        if FUSED_HEAD and fine_C.dtype == torch.int32 and fine_C.is_contiguous() and cls_score.is_contiguous():
            # Scale + exp, the points and the bench's logit boost in one launch (ops/head_stage.head_outputs)
            from ....ops.head_stage import head_outputs
            bbox_pred, points = head_outputs(reg, scale_vec, fine_C, vs_tab, B, cls_score, self.force_class_logit_boost)
        else:
            if self.force_class_logit_boost:
                cls_score = cls_score + torch.nn.functional.one_hot(rc, C).float() * self.force_class_logit_boost
            bbox_pred = torch.cat((torch.exp(reg[:, :6] * scale_vec[rc].unsqueeze(1)), reg[:, 6:]), dim=1)
            points = fine_C[:, 1:].float() * vs_tab[rc]
        if ME.MORTON_ROWS and cls_map.rows_batch_major:
            # rows of the class map are already (class, scene)-major (SparseTensor inserts in (batch, Morton) order): no sort,
            # no four gathers -- and none of their sort-based index_put backward passes
            merged, seg = [centerness, bbox_pred, cls_score, points], fb
        else:
            perm = torch.sort(fb, stable=True)[1]                          # (class, scene)-major, rows ascending inside
            merged, seg = [t[perm] for t in (centerness, bbox_pred, cls_score, points)], fb[perm]
        object.__setattr__(self, "_merged", {"centerness": merged[0], "bbox_pred": merged[1], "cls_score": merged[2],
                                             "points": merged[3], "seg": seg, "per_scene": per_scene})
        _tick("linears+outputs")
        assert min(per_scene) > 0, "forward empty"
        # the reference's per-class, per-scene lists are cut from the merged rows when somebody reads them: the batched loss and
        # the batched proposal stage work on the merged rows themselves (288 views and their Python lists per step otherwise)
        outs = _LazyClassLists(merged, per_scene, C, B)
        _tick("split")
        return outs

    def forward_single(self, x, scale, voxel_size, cls_id=None):
        """Per-class prediction heads (cagroup_head.py:627-652); returns per-scene lists."""
        centerness = self.centerness_conv(x).F
        cls_score = self.cls_conv(x).F
        if self.force_class_logit_boost and cls_id is not None:
            cls_score = cls_score.clone()
            cls_score[:, cls_id] += self.force_class_logit_boost
        reg = self.reg_conv(x).F
        bbox_pred = torch.cat((torch.exp(scale(reg[:, :6])), reg[:, 6:]), dim=1)
        perms = x.decomposition_permutations
        vs = cls_score.new_tensor(voxel_size)
        points = [c * vs for c in x.decomposed_coordinates]
        for p in points:
            assert len(p) > 0, "forward empty"
        return ([centerness[p] for p in perms], [bbox_pred[p] for p in perms], [cls_score[p] for p in perms], points)

    # ------------------------------------------------------------------ losses
    def loss(self, centernesses, bbox_preds, cls_scores, points, semantic_scores, voxel_offset, gt_bboxes, gt_labels,
             scene_points, img_metas, pts_semantic_mask, pts_instance_mask):
        nb = len(img_metas)
        if pts_semantic_mask is None:
            pts_semantic_mask = pts_instance_mask = [None] * nb
        has_gt = all(len(g) > 0 for g in gt_bboxes)
        if not has_gt and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size() > 1:
            # the two loss paths issue different collectives (one [B,3] all-reduce vs three scalars per scene): which one
            # runs must not depend on rank-local data.  The reference cannot train on such a scene either -- its proposal
            # target layer indexes an empty label tensor (cagroup_proposal_target_layer.py:120) -- so this is an input error.
            raise ValueError("a training scene without ground-truth boxes reached the dense head in a data-parallel run")
        if self.batched and self._merged is not None and has_gt:
            return self._loss_batched(semantic_scores, voxel_offset, gt_bboxes, gt_labels, scene_points,
                                      pts_semantic_mask, pts_instance_mask)
        assert len(centernesses[0]) == len(bbox_preds[0]) == len(cls_scores[0]) == len(points[0]) == nb \
            == len(gt_bboxes) == len(gt_labels) == len(pts_instance_mask) == len(pts_semantic_mask) == len(scene_points)
        sem_perms = semantic_scores.decomposition_permutations
        off_perms = voxel_offset.decomposition_permutations
        terms = []
        for i in range(nb):
            terms.append(self._loss_single(
                centernesses=[x[i] for x in centernesses], bbox_preds=[x[i] for x in bbox_preds],
                cls_scores=[x[i] for x in cls_scores], points=[x[i] for x in points],
                voxel_offset_preds=voxel_offset.F[off_perms[i]],
                original_points=voxel_offset.C[off_perms[i], 1:] * self.voxel_size,
                semantic_scores=semantic_scores.F[sem_perms[i]],
                semantic_points=semantic_scores.C[sem_perms[i], 1:] * self.voxel_size,
                img_meta=img_metas[i], gt_bboxes=gt_bboxes[i], gt_labels=gt_labels[i], scene_points=scene_points[i],
                pts_semantic_mask=pts_semantic_mask[i], pts_instance_mask=pts_instance_mask[i]))
        names = ("loss_centerness", "loss_bbox", "loss_cls", "loss_sem", "loss_vote")
        means = [torch.mean(torch.stack([t[j] for t in terms])) for j in range(5)]
        loss = means[0] + means[1] + means[2] + means[3] + means[4]
        return loss, DeferredLog(names + ("one_stage_loss",), torch.stack(means + [loss]))      # ONE device->host copy, when the log is read

    @torch.no_grad()
    def data_targets(self, vox_C, gt_bboxes, gt_labels, scene_points, sem_masks, ins_masks):
        """The training targets that depend on the DATA only (stride-2 voxel coordinates, boxes, raw points and their
        masks), not on the network: semantic label of every backbone voxel (assigner.assign_semantic,
        cagroup3d_assigner.py:132-152) and the vote targets (cagroup_head.py:418-498).  `CAGroup3D.prefetch_coordinates`
        computes them ahead of the step on its side stream; `_loss_batched` falls back to calling this itself."""
        from .target_assigner.cagroup3d_assigner import FLOAT_MAX, volume
        B, dev, vs = len(gt_bboxes), vox_C.device, self.voxel_size
        n_gt = [len(g) for g in gt_bboxes]
        gt = torch.cat([g.to(dev) for g in gt_bboxes])
        gl = torch.cat([l.to(dev).long() for l in gt_labels])
        gt_scene = torch.repeat_interleave(torch.arange(B, device=dev), ME.h2d(n_gt, torch.long, dev), output_size=sum(n_gt))
        vox_scene = vox_C[:, 0].long()
        vox_xyz = vox_C[:, 1:] * vs
        inside = find_points_in_boxes(vox_xyz, gt, point_seg=vox_scene, box_seg=gt_scene)      # same-scene pairs only
        vols = torch.where(inside, volume(gt).view(1, -1).expand(inside.shape), torch.full((1, 1), FLOAT_MAX, device=dev))
        min_vol, min_ind = vols.min(dim=1)
        semantic_labels = torch.where(min_vol == FLOAT_MAX, torch.full_like(min_ind, -1), gl[min_ind])
        # ---- vote targets
        N = vox_C.shape[0]
        equal_pts = len({sp.shape[0] for sp in scene_points}) == 1
        if not self.with_yaw:
            n_ins = (torch.stack(list(ins_masks)).amax(1) if equal_pts else torch.stack([im.max() for im in ins_masks])).cpu().numpy() + 1      # one host read for all scenes
        if not self.with_yaw and equal_pts:
            info = {}
            perms = _Perms(ME.rows_by_batch(vox_scene, B, info=info))
            perms.sorted = bool(info.get("sorted"))
            perms.starts = [0] + np.cumsum([p.shape[0] for p in perms]).tolist()
            counts = ME.h2d([p.shape[0] for p in perms], torch.long, dev)
            t, mk = self._vote_targets_masks_batched(vox_xyz, vox_scene, perms, gt_bboxes, scene_points, sem_masks, ins_masks, n_ins)
            off_t, off_m = t, mk.float()
            n_vox = counts.float()[vox_scene]
        else:
            off_t = torch.zeros((N, 3 * (self.gt_per_seed if self.with_yaw else 1)), device=dev)
            off_m, n_vox = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
            for b in range(B):
                rows = torch.nonzero(vox_scene == b).squeeze(1)
                op = vox_xyz[rows]
                if self.with_yaw:
                    t, mk = self._vote_targets_yaw(op, gt_bboxes[b], gt_labels[b])
                else:
                    t, mk = self._vote_targets_masks(op, gt_bboxes[b], scene_points[b], sem_masks[b], ins_masks[b], int(n_ins[b]))
                off_t[rows] = t
                off_m[rows] = mk.float()
                n_vox[rows] = float(len(rows))
        return {"n": N, "semantic_labels": semantic_labels, "vox_scene": vox_scene, "off_t": off_t, "off_m": off_m, "n_vox": n_vox}

    def _loss_batched(self, semantic_scores, voxel_offset, gt_bboxes, gt_labels, scene_points, sem_masks, ins_masks):
        """All scenes of the batch in one pass: the same five loss terms as `_loss_single` averaged over scenes
        (cagroup_head.py:322-396,399-555), with per-row normalisers instead of a Python loop over scenes and
        ONE all-reduce for every scene's three cross-rank means."""
        m = self._merged
        B = len(gt_bboxes)
        dev = m["points"].device
        vs = self.voxel_size
        n_gt = [len(g) for g in gt_bboxes]
        gt = torch.cat([g.to(dev) for g in gt_bboxes])
        gl = torch.cat([l.to(dev).long() for l in gt_labels])
        gt_scene = torch.repeat_interleave(torch.arange(B, device=dev), ME.h2d(n_gt, torch.long, dev), output_size=sum(n_gt))
        with torch.no_grad():
            pre = self._data_targets
            object.__setattr__(self, "_data_targets", None)
            if pre is None or pre["n"] != semantic_scores.C.shape[0]:
                pre = self.data_targets(semantic_scores.C, gt_bboxes, gt_labels, scene_points, sem_masks, ins_masks)
            semantic_labels, vox_scene, off_t, off_m, n_vox = (pre[k] for k in ("semantic_labels", "vox_scene", "off_t", "off_m", "n_vox"))
            vox_scene_off = vox_scene
            # ---- FCOS-style assignment of the class-map voxels (assigner.assign_all_classes, + same-scene mask)
            seg = m["seg"]
            pt_cls, pt_scene = seg // B, seg % B
            per = ME.h2d(m["per_scene"], torch.long, dev)                      # points of map (class c, scene b) at c*B+b
            centerness_targets, bbox_targets, labels = self.assigner.assign_all_classes(
                [m["points"]], gt, gl, pt_cls=pt_cls, pt_scene=pt_scene, gt_scene=gt_scene,
                n_map=per[gl.clamp(max=self.n_classes - 1) * B + gt_scene])
            # ---- per-scene normalisers, one all-reduce (the reference: 3 per scene, cagroup_head.py:523,530,538)
            pos = labels >= 0
            stats = torch.zeros((B, 3), device=dev)
            stats[:, 0].index_add_(0, vox_scene, (semantic_labels >= 0).float())
            stats[:, 1].index_add_(0, pt_scene, pos.float())
            stats[:, 2].index_add_(0, pt_scene, torch.where(pos, centerness_targets, torch.zeros_like(centerness_targets)))
            stats = reduce_mean(stats)
            sem_n_pos, n_pos = stats[:, 0].clamp(min=1.), stats[:, 1].clamp(min=1.)
            ctr_denorm = stats[:, 2].clamp(min=1e-6)

        from ....ops.focal_loss import sigmoid_focal_loss_rows
        C = self.n_classes

        def focal(pred, lab, row_w, loss_mod):      # FocalLoss + the per-scene avg_factor as a row weight, one pass
            return loss_mod.loss_weight * sigmoid_focal_loss_rows(pred, lab, row_w, loss_mod.gamma, loss_mod.alpha)
        loss_sem = focal(semantic_scores.F, semantic_labels, 1.0 / (sem_n_pos[vox_scene] * B), self.loss_sem)
        loss_cls = focal(m["cls_score"], labels, 1.0 / (n_pos[pt_scene] * B), self.loss_cls)
        # vote loss: smooth-L1 'sum' per scene, then mean over scenes
        if self.with_yaw:
            msum = torch.zeros(B, device=dev).index_add_(0, vox_scene_off, off_m)
            w = (off_m / (msum[vox_scene_off] + 1e-6)).unsqueeze(1)
            base = (voxel_offset.C[:, 1:] * vs).repeat(1, self.gt_per_seed)
            pred_v, tgt_v = base + voxel_offset.F, base + off_t
        else:
            w = (off_m / n_vox + 1e-6).unsqueeze(1)         # the reference's precedence quirk (:518): +1e-6 on every weight
            pred_v, tgt_v = voxel_offset.F, off_t
        beta = self.loss_offset.beta
        pos_inds = torch.nonzero(pos).squeeze(1)
        eps = torch.finfo(torch.float32).eps
        nd = m["bbox_pred"].shape[1]
        if FUSED_LOSSES and ((not self.with_yaw and nd == 6) or (self.with_yaw and nd == 8 and self.yaw_parametrization == "fcaf3d"
                                                                  and bbox_targets.shape[1] >= 7)):
            # vote loss, centerness loss and box loss as fused ops (one pass forward, one backward each: ops/fused_losses.py)
            from ....ops.fused_losses import positives_loss, smooth_l1_rows
            loss_vote = self.loss_offset.loss_weight * smooth_l1_rows(pred_v, tgt_v, w, beta) / B
            lp = positives_loss(m["centerness"], m["bbox_pred"], m["points"], centerness_targets, bbox_targets, pt_scene, n_pos,
                                ctr_denorm, pos_inds, self.loss_centerness.loss_weight / B, self.loss_bbox.loss_weight / B, eps)
            loss_centerness, loss_bbox = lp[0], lp[1]
        else:
            d = torch.abs(pred_v - tgt_v)
            el = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
            loss_vote = self.loss_offset.loss_weight * (el * w).sum() / B
            # centerness + box losses over the positives
            ps = pt_scene[pos_inds]
            pc, pb = m["centerness"][pos_inds], m["bbox_pred"][pos_inds]
            ct = centerness_targets[pos_inds].unsqueeze(1)
            bce = torch.nn.functional.binary_cross_entropy_with_logits(pc, ct, reduction="none")
            loss_centerness = self.loss_centerness.loss_weight * (bce.squeeze(1) / ((n_pos[ps] + eps) * B)).sum()
            boxes = self._bbox_pred_to_bbox(m["points"][pos_inds], pb)
            # IoU3DLoss returns pred.sum()*weight.sum() (== 0) for a scene whose weights are all zero (iou3d_loss.py:31);
            # with positives the weights (centerness targets) are > 0, so the plain weighted form is identical
            iou_el = self.loss_bbox.loss_function(boxes, bbox_targets[pos_inds], None, reduction="none")
            loss_bbox = self.loss_bbox.loss_weight * (iou_el * ct.squeeze(1) / (ctr_denorm[ps] * B)).sum()
        losses = [loss_centerness, loss_bbox, loss_cls, loss_sem, loss_vote]
        loss = losses[0] + losses[1] + losses[2] + losses[3] + losses[4]
        names = ("loss_centerness", "loss_bbox", "loss_cls", "loss_sem", "loss_vote")
        # (the terms stay on the device until the log is read: common_utils.DeferredLog)
        return loss, DeferredLog(names + ("one_stage_loss",), torch.stack(losses + [loss]))

    def _vote_targets_yaw(self, original_points, gt_bboxes, gt_labels):
        """SUN RGB-D: up to gt_per_seed box-centre votes per voxel (cagroup_head.py:418-452)."""
        n = original_points.shape[0]
        vote_targets = original_points.new_zeros([n, 3 * self.gt_per_seed])
        vote_masks = original_points.new_zeros([n], dtype=torch.long)
        vote_idx = original_points.new_zeros([n], dtype=torch.long)
        inside_all = find_points_in_boxes(points=original_points, gt_bboxes=gt_bboxes)
        for i in range(gt_labels.shape[0]):
            ind = torch.nonzero(inside_all[:, i], as_tuple=False).squeeze(-1)
            pts = original_points[ind]
            vote_masks[ind] = 1
            tmp = vote_targets[ind]
            votes = gt_bboxes[i, :3].unsqueeze(0).to(pts.device) - pts[:, :3]
            for j in range(self.gt_per_seed):
                col = torch.nonzero(vote_idx[ind] == j, as_tuple=False).squeeze(-1)
                tmp[col, int(j * 3):int(j * 3 + 3)] = votes[col]
                if j == 0:
                    tmp[col] = votes[col].repeat(1, self.gt_per_seed)
            vote_targets[ind] = tmp
            vote_idx[ind] = torch.clamp(vote_idx[ind] + 1, max=2)
        return vote_targets, vote_masks

    def _vote_targets_masks(self, original_points, gt_bboxes, scene_points, sem_mask, ins_mask, n_ins=None):
        """ScanNet: nearest raw point (kNN k=1) gives each voxel its instance; the target is the offset to
        the centre of the GT box nearest to that instance's bbox centre (cagroup_head.py:454-498).
        Instance statistics are segment reductions (the reference loops over torch.unique with a host
        sync per instance)."""
        dev = scene_points.device
        if n_ins is None:
            n_ins = int(ins_mask.max()) + 1
        xyz = scene_points[:, :3]
        # dense masked reductions over the few instances (atomic scatter_reduce contends on ~20 slots)
        member = ins_mask.view(-1, 1) == torch.arange(n_ins, device=dev).view(1, -1)            # (n_pts, n_ins)
        big = torch.full((1, 1, 1), float("inf"), device=dev)
        lo = torch.where(member.unsqueeze(2), xyz.unsqueeze(1), big).amin(0)
        hi = torch.where(member.unsqueeze(2), xyz.unsqueeze(1), -big).amax(0)
        ar = torch.arange(ins_mask.shape[0], device=dev).view(-1, 1)
        first = torch.where(member, ar, torch.full_like(ar, ins_mask.shape[0])).amin(0)
        present = first < ins_mask.shape[0]
        sem_first = sem_mask[first.clamp(max=ins_mask.shape[0] - 1)]
        is_obj = present & (sem_first < self.n_classes)
        center = 0.5 * (lo + hi)
        center = torch.where(is_obj.unsqueeze(1), center, torch.zeros_like(center))
        gt_ctr = gt_bboxes[:, :3].to(dev)
        match = torch.argmin(torch.cdist(center.unsqueeze(0), gt_ctr.unsqueeze(0)).squeeze(0), dim=1)
        instance_center = torch.where(is_obj.unsqueeze(1), gt_ctr[match],
                                      torch.where(present.unsqueeze(1), torch.full_like(center, -10000.),
                                                  torch.zeros_like(center)))
        idx = knn(1, xyz[None].contiguous(), original_points[None, ::].contiguous())[0].long()  # (1, n)
        instance_idx = ins_mask[idx.view(-1)].view(idx.shape[0], idx.shape[1])
        valid = (instance_idx == instance_idx[0]).all(0)
        if idx.shape[0] == 1:
            major = instance_idx[0]
        else:  # majority instance over the k neighbours
            votes = (instance_idx[None] == torch.arange(n_ins, device=dev).view(-1, 1, 1)).sum(1)
            major = torch.argmax(votes, dim=0)
        offset_t = instance_center[major] - original_points
        offset_m = torch.where(offset_t < -100., torch.zeros_like(offset_t), torch.ones_like(offset_t)).all(1)
        offset_t = torch.where(offset_t < -100., torch.zeros_like(offset_t), offset_t)
        return offset_t, offset_m * valid

    def _vote_targets_masks_batched(self, vox_xyz, vox_scene, perms, gt_bboxes, scene_points, sem_masks, ins_masks, n_ins):
        """`_vote_targets_masks` for all scenes of the batch in one pass (scenes with equal raw point counts; the
        kNN stays one launch per scene -- the raw point sets are separate search spaces).  Returns
        (offset_t [N,3], offset_m [N]) in voxel row order."""
        dev = vox_xyz.device
        B, P = len(scene_points), scene_points[0].shape[0]
        xyz = torch.stack([sp[:, :3] for sp in scene_points])                          # [B, P, 3]
        ins, sem = torch.stack(list(ins_masks)), torch.stack(list(sem_masks))          # [B, P]
        I = int(n_ins.max())
        if FUSED_HEAD and ins.dtype == torch.int64 and sem.dtype == torch.int64 and all(len(g) > 0 for g in gt_bboxes):
            # instance bounding boxes, the box each instance votes for and the masked offsets as stage ops (ops/head_stage.py):
            # the tensor form below materialises [B, P, I, 3] masked copies of the points three times over
            from ....ops import head_stage as HS
            n_gt = [len(g) for g in gt_bboxes]
            gt_ctr = torch.nn.utils.rnn.pad_sequence([g[:, :3].to(dev) for g in gt_bboxes], batch_first=True)
            contiguous = all(int(p.shape[0]) > 0 for p in perms) and getattr(perms, "sorted", False)
            if contiguous:          # batch-major rows: scene b is a row range
                nearest = torch.cat([knn(1, xyz[b:b + 1], vox_xyz[perms.starts[b]:perms.starts[b + 1]][None].contiguous())[0, 0] for b in range(B)]).long()
            else:
                nearest = torch.empty(vox_xyz.shape[0], dtype=torch.long, device=dev)
                for b in range(B):
                    nearest[perms[b]] = knn(1, xyz[b:b + 1], vox_xyz[perms[b]][None].contiguous())[0, 0].long()
            return HS.vote_targets(xyz, ins, sem, gt_ctr, ME.h2d(n_gt, torch.int32, dev), self.n_classes, I, vox_xyz, vox_scene, nearest)
        member = ins.unsqueeze(2) == torch.arange(I, device=dev).view(1, 1, -1)        # [B, P, I]
        big = torch.full((1, 1, 1, 1), float("inf"), device=dev)
        lo = torch.where(member.unsqueeze(3), xyz.unsqueeze(2), big).amin(1)           # [B, I, 3]
        hi = torch.where(member.unsqueeze(3), xyz.unsqueeze(2), -big).amax(1)
        ar = torch.arange(P, device=dev).view(1, -1, 1)
        first = torch.where(member, ar, torch.full_like(ar, P)).amin(1)                # [B, I]
        present = first < P
        sem_first = sem.gather(1, first.clamp(max=P - 1))
        is_obj = present & (sem_first < self.n_classes)
        center = 0.5 * (lo + hi)
        center = torch.where(is_obj.unsqueeze(2), center, torch.zeros_like(center))
        n_gt = [len(g) for g in gt_bboxes]
        G = max(n_gt)
        gt_ctr = torch.zeros((B, G, 3), device=dev)
        for b, g in enumerate(gt_bboxes):
            gt_ctr[b, :n_gt[b]] = g[:, :3].to(dev)
        gt_ok = torch.arange(G, device=dev).view(1, -1) < ME.h2d(n_gt, torch.long, dev).view(-1, 1)
        d = torch.where(gt_ok.unsqueeze(1), torch.cdist(center, gt_ctr), torch.full((1, 1, 1), float("inf"), device=dev))
        match = torch.argmin(d, dim=2)                                                  # [B, I]
        instance_center = torch.where(is_obj.unsqueeze(2), gt_ctr.gather(1, match.unsqueeze(2).expand(-1, -1, 3)),
                                      torch.where(present.unsqueeze(2), torch.full_like(center, -10000.),
                                                  torch.zeros_like(center)))
        nearest = torch.empty(vox_xyz.shape[0], dtype=torch.long, device=dev)
        for b in range(B):
            nearest[perms[b]] = knn(1, xyz[b:b + 1], vox_xyz[perms[b]][None].contiguous())[0, 0].long()
        major = ins.view(-1)[vox_scene * P + nearest]
        offset_t = instance_center.view(-1, 3)[vox_scene * I + major] - vox_xyz
        offset_m = torch.where(offset_t < -100., torch.zeros_like(offset_t), torch.ones_like(offset_t)).all(1)
        offset_t = torch.where(offset_t < -100., torch.zeros_like(offset_t), offset_t)
        return offset_t, offset_m

    def _loss_single(self, centernesses, bbox_preds, cls_scores, points, voxel_offset_preds, original_points,
                     semantic_scores, semantic_points, img_meta, gt_bboxes, gt_labels, scene_points,
                     pts_semantic_mask, pts_instance_mask):
        with torch.no_grad():
            semantic_labels, _ = self.assigner.assign_semantic(semantic_points, gt_bboxes, gt_labels, self.n_classes)
            assign = self.assigner.assign_all_classes if self.batched else self.assigner.assign
            centerness_targets, bbox_targets, labels = assign(points, gt_bboxes, gt_labels)
            if self.with_yaw:
                offset_targets, offset_masks = self._vote_targets_yaw(original_points, gt_bboxes, gt_labels)
            elif pts_semantic_mask is not None and pts_instance_mask is not None:
                offset_targets, offset_masks = self._vote_targets_masks(original_points, gt_bboxes, scene_points,
                                                                        pts_semantic_mask, pts_instance_mask)
            else:
                raise NotImplementedError
        centerness = torch.cat(centernesses)
        bbox_preds = torch.cat(bbox_preds)
        cls_scores = torch.cat(cls_scores)
        points = torch.cat(points)

        if self.with_yaw:
            w = (offset_masks.float() / (offset_masks.float().sum() + 1e-6)).unsqueeze(1).repeat(1, 9)
            base = original_points.repeat(1, self.gt_per_seed)
            loss_offset = self.loss_offset(base + voxel_offset_preds, base + offset_targets, weight=w)
        else:
            # the reference's operator precedence (cagroup_head.py:518): every weight gets +1e-6
            w = (offset_masks.float() / torch.ones_like(offset_masks).float().sum() + 1e-6).unsqueeze(1).repeat(1, 3)
            loss_offset = self.loss_offset(voxel_offset_preds, offset_targets, weight=w)

        pos = labels >= 0
        # the three cross-rank means of the reference (:523,530,538) in ONE all-reduce
        stats = torch.stack([(semantic_labels >= 0).sum().float(), pos.sum().float(),
                             centerness_targets[pos].sum().detach()])
        stats = reduce_mean(stats)
        sem_n_pos, n_pos = stats[0].clamp(min=1.), stats[1].clamp(min=1.)
        centerness_denorm = stats[2].clamp(min=1e-6)

        loss_sem = self.loss_sem(semantic_scores, semantic_labels, avg_factor=sem_n_pos)
        loss_cls = self.loss_cls(cls_scores, labels, avg_factor=n_pos)
        pos_inds = torch.nonzero(pos).squeeze(1)
        pos_centerness, pos_bbox_preds = centerness[pos_inds], bbox_preds[pos_inds]
        pos_ctr_targets = centerness_targets[pos_inds].unsqueeze(1)
        if len(pos_inds) > 0:
            loss_centerness = self.loss_centerness(pos_centerness, pos_ctr_targets, avg_factor=n_pos)
            loss_bbox = self.loss_bbox(self._bbox_pred_to_bbox(points[pos_inds], pos_bbox_preds), bbox_targets[pos_inds],
                                       weight=pos_ctr_targets.squeeze(1), avg_factor=centerness_denorm)
        else:
            loss_centerness, loss_bbox = pos_centerness.sum(), pos_bbox_preds.sum()
        return loss_centerness, loss_bbox, loss_cls, loss_sem, loss_offset

    # ------------------------------------------------------------------ proposals
    def get_bboxes(self, centernesses, bbox_preds, cls_scores, points, img_metas, rescale=False):
        assert len(centernesses[0]) == len(bbox_preds[0]) == len(cls_scores[0]) == len(points[0]) == len(img_metas)
        return [self._get_bboxes_single([x[i] for x in centernesses], [x[i] for x in bbox_preds],
                                        [x[i] for x in cls_scores], [x[i] for x in points], img_metas[i])
                for i in range(len(img_metas))]

    @staticmethod
    def _sort_seg_desc(seg, score):
        """Permutation ordering entries by (segment ascending, score descending)."""
        o1 = torch.sort(score, descending=True, stable=True)[1]
        o2 = torch.sort(seg[o1], stable=True)[1]
        return o1[o2]

    def get_bboxes_batched(self, m, batch_size):
        """_get_bboxes_single for every scene at once (cagroup_head.py:579-624,747-797): per class map the
        top NMS_PRE candidates, then ONE batched NMS launch over all (scene, class) problems and three host
        reads in total (the loop version syncs twice per class per scene)."""
        C, B = self.n_classes, batch_size
        dev = m["points"].device
        nd = m["bbox_pred"].shape[1]
        if FUSED_HEAD and m["seg"].shape[0] > 0 and (nd == 6 or (nd == 8 and self.yaw_parametrization == "fcaf3d")) \
                and B * C < 512 and len(m["per_scene"]) == B * C:
            return self._get_bboxes_fused(m, B)
        scores = m["cls_score"].detach().sigmoid() * m["centerness"].detach().sigmoid()
        seg = m["seg"]                                                     # c*B + b, non-decreasing
        per = ME.h2d(m["per_scene"], torch.long, dev)
        pre = int(self.nms_cfg.NMS_PRE)
        if pre > 0:
            order = self._sort_seg_desc(seg, scores.max(dim=1)[0])
            sseg = seg[order]
            rank = torch.arange(sseg.shape[0], device=dev) - (torch.cumsum(per, 0) - per)[sseg]
            cand = order[rank < pre]
        else:
            cand = torch.arange(seg.shape[0], device=dev)
        c_scores = scores[cand]
        c_scene = seg[cand] % B
        boxes = self._bbox_pred_to_bbox(m["points"][cand].detach(), m["bbox_pred"][cand].detach())
        yaw_flag = boxes.shape[1] == 7
        if not yaw_flag:
            boxes = torch.cat((boxes, torch.zeros_like(boxes[:, :1])), dim=1)
        j, i = torch.nonzero(c_scores > self.nms_cfg.SCORE_THR, as_tuple=True)       # host read 1
        e_seg = c_scene[j] * C + i
        e_score = c_scores[j, i]
        o = self._sort_seg_desc(e_seg, e_score)
        j, i, e_seg, e_score = j[o], i[o], e_seg[o], e_score[o]
        counts = ME.count_ids(e_seg, B * C).cpu().numpy()                  # host read 2
        seg_off = np.zeros(B * C + 1, dtype=np.int64)
        seg_off[1:] = np.cumsum(counts)
        e_boxes = boxes[j].contiguous()
        nms_boxes = e_boxes
        if yaw_flag:
            nms_boxes = e_boxes.clone()
            nms_boxes[:, 6] *= -1                                                      # heading sign fix (:770)
        keep, num = nms_batched_sorted(nms_boxes, seg_off, float(self.nms_cfg.IOU_THR), yaw_flag)
        num = num.cpu().numpy().astype(np.int64)                                       # host read 3
        off = seg_off
        idx = np.concatenate([np.arange(off[g], off[g] + num[g]) for g in range(B * C)] + [np.zeros(0, np.int64)])
        gof = np.repeat(np.arange(B * C), num)
        idx_d = ME.h2d(torch.from_numpy(idx), torch.long, dev)
        sel = keep[idx_d] + ME.h2d(torch.from_numpy(off[gof]), torch.long, dev)
        out_boxes, out_scores = e_boxes[sel], e_score[sel]
        out_labels = ME.h2d(torch.from_numpy(gof % C), torch.long, dev)
        per_scene = num.reshape(B, C).sum(1).tolist()
        # the same proposals FLAT (scene-major): the RoI head's fused training path reads them in place instead of padding them
        object.__setattr__(self, "_flat_props", (out_boxes, out_scores, out_labels, per_scene))
        return list(zip(torch.split(out_boxes, per_scene), torch.split(out_scores, per_scene),
                        torch.split(out_labels, per_scene)))

    @torch.no_grad()
    def _get_bboxes_fused(self, m, B):
        """`get_bboxes_batched` through the stage ops of include/cagroup3d_stages.h: ONE sort for the per-map top NMS_PRE
        (key = segment | inverted score bits), the candidates' (row, class) entries above SCORE_THR emitted with a key that
        already IS their final order (problem | inverted score bits | entry id), ONE sort of those keys, decoding only the
        boxes that enter NMS.  Two host reads (entry counts, kept counts) instead of three; ~25 launches instead of ~70."""
        from ctypes import c_float, c_int32, c_int64
        from .... import _lib
        from ...._lib import ptr
        lib = _lib.get()
        C, dev = self.n_classes, m["points"].device
        scores = (m["cls_score"].detach().sigmoid() * m["centerness"].detach().sigmoid()).contiguous()
        smax = scores.max(dim=1)[0]
        seg = m["seg"].contiguous()
        E = seg.shape[0]
        per = np.asarray(m["per_scene"], dtype=np.int64)
        pre = int(self.nms_cfg.NMS_PRE)
        if pre > 0:
            keys = torch.empty(E, dtype=torch.int64, device=dev)
            lib.call("cg3d_prop_keys", ptr(seg), ptr(smax), c_int64(E), ptr(keys), lib.stream())
            order = torch.sort(keys, stable=True)[1]
            cnt = np.minimum(per, pre)
        else:
            order = torch.arange(E, device=dev)
            cnt = per
        nseg = per.shape[0]
        seg_start = np.cumsum(per) - per
        cand_off = np.concatenate([[0], np.cumsum(cnt)])
        ncand = int(cand_off[-1])
        if ncand * C >= (1 << 22):
            raise NotImplementedError("more than 2^22 (candidate, class) pairs")
        tab = ME.h2d(np.concatenate([seg_start, cand_off]), torch.int32, dev)
        seg_start_d, cand_off_d = tab[:nseg], tab[nseg:]
        points, bbox_pred = m["points"].detach().contiguous(), m["bbox_pred"].detach().contiguous()
        nd = bbox_pred.shape[1]
        ekeys = torch.empty(max(ncand * C, 1), dtype=torch.int64, device=dev)
        counts = torch.empty(B * C + 1, dtype=torch.int32, device=dev)
        lib.call("cg3d_prop_entries", ptr(order), ptr(seg_start_d), ptr(cand_off_d), c_int32(nseg), c_int32(ncand), c_int32(B),
                 ptr(scores), c_int32(C), c_float(float(self.nms_cfg.SCORE_THR)), ptr(ekeys), ptr(counts), lib.stream())
        cnt_host = counts.cpu().numpy().astype(np.int64)                    # host read 1
        total = int(cnt_host[-1])
        seg_off = np.zeros(B * C + 1, dtype=np.int64)
        seg_off[1:] = np.cumsum(cnt_host[:-1])
        skeys = torch.sort(ekeys[:total])[0]
        out3 = torch.empty((max(total, 1), 15), dtype=torch.float32, device=dev)
        flat = out3.view(-1)
        e_boxes, nms_boxes, e_score = flat[:7 * total].view(total, 7), flat[7 * total:14 * total].view(total, 7), flat[14 * total:15 * total]
        lib.call("cg3d_prop_gather", ptr(skeys), c_int64(total), ptr(order), ptr(seg_start_d), ptr(cand_off_d), c_int32(nseg), c_int32(C),
                 ptr(points), ptr(bbox_pred), c_int32(nd), ptr(scores), ptr(e_boxes), ptr(nms_boxes), ptr(e_score), lib.stream())
        keep, num = nms_batched_sorted(nms_boxes, seg_off, float(self.nms_cfg.IOU_THR), nd == 8)
        num = num.cpu().numpy().astype(np.int64)                                       # host read 2
        off = seg_off
        idx = np.concatenate([np.arange(off[g], off[g] + num[g]) for g in range(B * C)] + [np.zeros(0, np.int64)])
        gof = np.repeat(np.arange(B * C), num)
        ht = ME.h2d(np.concatenate([idx, off[gof], gof % C]), torch.long, dev)          # one table for the three index vectors
        n_keep = idx.shape[0]
        sel = keep[ht[:n_keep]] + ht[n_keep:2 * n_keep]
        out_boxes, out_scores = e_boxes[sel], e_score[sel]
        out_labels = ht[2 * n_keep:]
        per_scene = num.reshape(B, C).sum(1).tolist()
        object.__setattr__(self, "_flat_props", (out_boxes, out_scores, out_labels, per_scene))
        return list(zip(torch.split(out_boxes, per_scene), torch.split(out_scores, per_scene),
                        torch.split(out_labels, per_scene)))

    def _get_bboxes_single(self, centernesses, bbox_preds, cls_scores, points, img_meta):
        """score = sigmoid(cls) * sigmoid(centerness); top NMS_PRE per class map; per-class NMS
        (cagroup_head.py:579-624)."""
        all_boxes, all_scores, all_sem = [], [], []
        for centerness, bbox_pred, cls_score, point in zip(centernesses, bbox_preds, cls_scores, points):
            scores = cls_score.sigmoid() * centerness.sigmoid()
            sem = cls_score.sigmoid() if self.use_sem_score else None
            if len(scores) > self.nms_cfg.NMS_PRE > 0:
                _, ids = scores.max(dim=1)[0].topk(self.nms_cfg.NMS_PRE)
                bbox_pred, scores, point = bbox_pred[ids], scores[ids], point[ids]
                sem = sem[ids] if sem is not None else None
            all_boxes.append(self._bbox_pred_to_bbox(point, bbox_pred))
            all_scores.append(scores)
            if sem is not None:
                all_sem.append(sem)
        bboxes, scores = torch.cat(all_boxes), torch.cat(all_scores)
        sem_scores = torch.cat(all_sem) if self.use_sem_score else None
        agnostic = self.training and self.nms_cfg.get("SCORE_THR_AGNOSTIC", None) is not None
        fn = self.class_agnostic_nms if agnostic else self._nms
        return fn(bboxes, scores, img_meta, sem_scores=sem_scores) if self.use_sem_score else fn(bboxes, scores, img_meta)

    def _bbox_pred_to_bbox(self, points, bbox_pred):
        """(dx-,dx+,dy-,dy+,dz-,dz+[, yaw params]) at a point -> (x,y,z,w,l,h[,alpha]) (cagroup_head.py:654-703)."""
        if bbox_pred.shape[0] == 0:
            return bbox_pred
        if bbox_pred.shape[1] == 6:
            # the same arithmetic on strided column slices (two views instead of twelve selects, each of which costs a
            # zero-fill + copy in backward): centre = p + (d+ - d-) / 2, size = d- + d+
            lo, hi = bbox_pred[:, 0:6:2], bbox_pred[:, 1:6:2]
            return torch.cat([points[:, :3] + (hi - lo) / 2, lo + hi], dim=1)
        xc = points[:, 0] + (bbox_pred[:, 1] - bbox_pred[:, 0]) / 2
        yc = points[:, 1] + (bbox_pred[:, 3] - bbox_pred[:, 2]) / 2
        zc = points[:, 2] + (bbox_pred[:, 5] - bbox_pred[:, 4]) / 2
        base = torch.stack([xc, yc, zc, bbox_pred[:, 0] + bbox_pred[:, 1], bbox_pred[:, 2] + bbox_pred[:, 3],
                            bbox_pred[:, 4] + bbox_pred[:, 5]], -1)
        if self.yaw_parametrization == "naive":
            return torch.cat((base, bbox_pred[:, 6:7]), -1)
        if self.yaw_parametrization == "sin-cos":
            norm = torch.pow(torch.pow(bbox_pred[:, 6:7], 2) + torch.pow(bbox_pred[:, 7:8], 2), 0.5)
            return torch.cat((base, torch.atan2(bbox_pred[:, 6:7] / norm, bbox_pred[:, 7:8] / norm)), -1)
        # 'fcaf3d': (sin(2a) ln q, cos(2a) ln q)
        scale = bbox_pred[:, 0] + bbox_pred[:, 1] + bbox_pred[:, 2] + bbox_pred[:, 3]
        q = torch.exp(torch.sqrt(torch.pow(bbox_pred[:, 6], 2) + torch.pow(bbox_pred[:, 7], 2)))
        alpha = 0.5 * torch.atan2(bbox_pred[:, 6], bbox_pred[:, 7])
        return torch.stack((xc, yc, zc, scale / (1 + q), scale / (1 + q) * q, bbox_pred[:, 5] + bbox_pred[:, 4], alpha), dim=-1)

    def _finish_nms(self, bboxes, n_classes, yaw_flag, parts, sem_scores):
        if len(parts[0]):
            cat = [torch.cat(p, dim=0) for p in parts[:3]]
            sem = torch.cat(parts[3], dim=0) if sem_scores is not None else None
        else:
            cat = [bboxes.new_zeros((0, 7 if not yaw_flag else bboxes.shape[1])), bboxes.new_zeros((0,)), bboxes.new_zeros((0,))]
            sem = bboxes.new_zeros((0, n_classes)) if sem_scores is not None else None
        nb = cat[0]
        if not yaw_flag:
            nb = torch.cat([nb[:, :6], nb.new_zeros(nb.shape[0], 1)], dim=1)
        return (nb, cat[1], cat[2], sem) if sem_scores is not None else (nb, cat[1], cat[2])

    def _nms(self, bboxes, scores, img_meta, sem_scores=None):
        """Per-class NMS above SCORE_THR (cagroup_head.py:747-797); no yaw -> axis-aligned BEV NMS."""
        n_classes = scores.shape[1]
        yaw_flag = bboxes.shape[1] == 7
        parts = ([], [], [], [])
        above = scores > self.nms_cfg.SCORE_THR
        has_any = above.any(dim=0).tolist()          # one host sync for all classes
        for i in range(n_classes):
            if not has_any[i]:
                continue
            ids = above[:, i]
            class_scores, class_bboxes = scores[ids, i], bboxes[ids]
            if yaw_flag:
                nms_boxes = class_bboxes.clone()
                nms_boxes[..., 6] *= -1              # heading sign fix before NMS (:770)
                keep, _ = nms_gpu(nms_boxes, class_scores, self.nms_cfg.IOU_THR)
            else:
                class_bboxes = torch.cat((class_bboxes, torch.zeros_like(class_bboxes[:, :1])), dim=1)
                keep, _ = nms_normal_gpu(class_bboxes, class_scores, self.nms_cfg.IOU_THR)
            parts[0].append(class_bboxes[keep])
            parts[1].append(class_scores[keep])
            parts[2].append(bboxes.new_full(class_scores[keep].shape, i, dtype=torch.long))
            if sem_scores is not None:
                parts[3].append(sem_scores[ids][keep])
        return self._finish_nms(bboxes, n_classes, yaw_flag, parts, sem_scores)

    def class_agnostic_nms(self, bboxes, scores, img_meta, sem_scores=None):
        """(cagroup_head.py:705-745); unused by the shipped configs (no SCORE_THR_AGNOSTIC)."""
        n_classes = scores.shape[1]
        yaw_flag = bboxes.shape[1] == 7
        max_scores, labels = scores.max(dim=1)
        if not yaw_flag:
            bboxes = torch.cat((bboxes, torch.zeros_like(bboxes[:, :1])), dim=1)
        ids = max_scores > self.nms_cfg.SCORE_THR_AGNOSTIC
        parts = ([], [], [], [])
        if ids.any():
            cb, cs, cl = bboxes[ids], max_scores[ids], labels[ids]
            nb = cb.clone()
            if yaw_flag:
                nb[..., 6] *= -1
            keep, _ = (nms_gpu if yaw_flag else nms_normal_gpu)(nb, cs, self.nms_cfg.IOU_THR)
            parts[0].append(cb[keep]); parts[1].append(cs[keep]); parts[2].append(cl[keep])
            if sem_scores is not None:
                parts[3].append(sem_scores[ids][keep])
        return self._finish_nms(bboxes, n_classes, yaw_flag, parts, sem_scores)


def split_gt_boxes(gt_boxes, label_dtype=torch.long):
    """Zero-padded [B, Gmax, 8] -> per-scene (boxes [G,7], labels [G]) lists
    (cagroup_head.py:298-318, cagroup3d.py:118-135)."""
    boxes, labels = GtList(), []
    valid = ~(gt_boxes == 0.).all(dim=-1)
    v = valid.cpu().numpy()                        # ONE host read (a boolean-mask selection per scene is a sync per scene)
    counts = v.sum(1)
    prefix = all(int(n) == 0 or bool(v[b, :int(n)].all()) for b, n in enumerate(counts))
    if prefix:
        # the padding rows come last (collate_batch): every scene is a prefix slice of ONE contiguous copy / ONE cast
        b7, l7 = gt_boxes[..., :7].contiguous(), gt_boxes[..., 7].to(label_dtype)
        for b, n in enumerate(counts):
            boxes.append(b7[b, :int(n)])
            labels.append(l7[b, :int(n)])
        boxes.prefix_counts = [int(x) for x in counts]
        return boxes, labels
    for b in range(gt_boxes.shape[0]):
        g = gt_boxes[b][ME.h2d(np.nonzero(v[b])[0], torch.long, gt_boxes.device)]
        boxes.append(g[:, :7].contiguous())
        labels.append(g[:, 7].to(label_dtype))
    return boxes, labels


class _Perms(list):
    """Per-scene row lists + whether they are consecutive row ranges (`sorted`, `starts`)."""
    sorted, starts = False, None


class GtList(list):
    """Per-scene ground-truth boxes; `prefix_counts` = real boxes per scene when every scene's boxes are the leading rows of
    the zero-padded batch tensor (then that tensor itself can be handed to the stage kernels), else None."""
    prefix_counts = None
