"""CAGroup3DRoIHead: fully sparse RoI grid pooling + box refinement (mirror of
pcdet/models/roi_heads/cagroup_roi_head.py:14-620) on the gfx950 engine.

RoI pooling = (a) a k5 sparse conv of the backbone tensor evaluated at the UNIQUE 0.04 m grid
voxels touched by the RoIs' 7^3 grid points, (b) a per-RoI 7^3 -> centre contraction.  (b) is a
kernel-7 convolution whose only output is the grid centre, i.e. exactly one dense GEMM
[R, 343*C] x [343*C, C]; it is run as such (the reference builds a fake 7^3 sparse tensor per RoI,
cagroup_roi_head.py:74-91)."""
import numpy as np
import torch
import torch.nn as nn

from .... import me as ME
from ....ops.iou3d_nms_utils import nms_gpu, nms_normal_gpu
from ...utils import common_utils
from ...utils.common_utils import DeferredLog
from ...utils.iou3d_loss import IoU3DLoss
from ...utils.loss_utils import WeightedSmoothL1Loss
from ..model_utils.cagroup_utils import CAGroupResidualCoder as ResidualCoder
from .target_assigner.cagroup_proposal_target_layer import ProposalTargetLayer


FUSED_ROI = __import__("os").environ.get("CG3D_FUSED_ROI", "1") != "0"


class SimplePoolingLayer(nn.Module):
    def __init__(self, channels=(128, 128, 128), grid_kernel_size=5, grid_num=7, voxel_size=0.04, coord_key=2,
                 point_cloud_range=(-5.12 * 3, -5.12 * 3, -5.12 * 3, 5.12 * 3, 5.12 * 3, 5.12 * 3),
                 corner_offset_emb=False, pooling=False):
        super().__init__()
        self.voxel_size, self.coord_key, self.grid_num, self.pooling = voxel_size, coord_key, grid_num, pooling
        pcr = point_cloud_range
        self.grid_size = [int((pcr[3 + i] - pcr[i]) / voxel_size) for i in range(3)]
        self.grid_conv = ME.MinkowskiConvolution(channels[0], channels[1], kernel_size=grid_kernel_size, dimension=3)
        self.grid_bn = ME.MinkowskiBatchNorm(channels[1])
        self.grid_relu = ME.MinkowskiELU()
        if pooling:
            self.pooling_conv = ME.MinkowskiConvolution(channels[1], channels[2], kernel_size=grid_num, dimension=3)
            self.pooling_bn = ME.MinkowskiBatchNorm(channels[1])
        self.init_weights()

    def init_weights(self):
        nn.init.normal_(self.grid_conv.kernel, std=.01)
        if self.pooling:
            nn.init.normal_(self.pooling_conv.kernel, std=.01)

    def forward(self, sp_tensor, grid_points, grid_corners=None, box_centers=None, batch_size=None):
        """grid_points (B*R*G^3, 4) = (b, x, y, z) in metres, RoI-major then grid index (ix, iy, iz)."""
        half = self.grid_size[0] // 2
        vox = torch.floor(grid_points[:, 1:4] / self.voxel_size)
        vox = torch.clamp(vox, min=-self.grid_size[0] / 2 + 1, max=self.grid_size[0] / 2 - 1).long() + half
        gs = self.grid_size
        lin = ((grid_points[:, 0].long() * gs[0] + vox[:, 0]) * gs[1] + vox[:, 1]) * gs[2] + vox[:, 2]
        unq, inv = torch.unique(lin, return_inverse=True)             # sorted -> coords already unique
        uc = torch.stack((unq // (gs[0] * gs[1] * gs[2]), (unq // (gs[1] * gs[2])) % gs[0] - half,
                          (unq // gs[2]) % gs[1] - half, unq % gs[2] - half), dim=1)
        uc[:, 1:4] *= self.coord_key
        feat = self.grid_bn(self.grid_conv(sp_tensor, uc.int()), act=ME.ACT_ELU).F      # BN + ELU fused
        return self._pool(feat, inv)

    def forward_rois(self, sp_tensor, rois, rois_per_scene, with_yaw):
        """The same layer from the RoIs themselves, float32 [n,7]: the grid points are generated, quantised and clamped by one
        launch (ops/roi_stage.roi_grid_coords) and de-duplicated by the coordinate map the convolution is evaluated on
        (cg3d_coord_map_build: hash insert, no sort) -- in place of ~25 tensor launches and a sort-based torch.unique.  Rows of
        the de-duplicated map come in first-occurrence order instead of sorted order; nothing downstream depends on it."""
        from ....ops.roi_stage import roi_grid_coords
        gs = self.grid_size
        assert gs[0] == gs[1] == gs[2]
        coords = roi_grid_coords(rois, rois_per_scene, self.grid_num, with_yaw, self.voxel_size, -gs[0] / 2 + 1, gs[0] / 2 - 1,
                                 self.coord_key)
        out, inv = self.grid_conv(sp_tensor, coords, return_inverse=True)
        return self._pool(self.grid_bn(out, act=ME.ACT_ELU).F, inv)

    def _pool(self, feat, inv):
        if not self.pooling:
            return ME.gather_rows(feat, inv)       # scatter-add backward (atomics), not torch's sort-based index_put
        # one row per RoI, (grid, channel) order, times the kernel as a [G C, C2] matrix (ME.roi_contract)
        pooled = ME.roi_contract(feat, inv, self.pooling_conv.kernel)
        return ME.fused_bn_act(pooled, [self.pooling_bn.bn])       # (the fused rows form: also the cross-rank statistics of --sync_bn)


class CAGroup3DRoIHead(nn.Module):
    def __init__(self, model_cfg, cls_loss_type="BinaryCrossEntropy", reg_loss_type="smooth-l1", **kwargs):
        super().__init__()
        cfg = model_cfg
        self.middle_feature_source = cfg.MIDDLE_FEATURE_SOURCE
        self.num_class = cfg.NUM_CLASSES
        self.code_size = cfg.CODE_SIZE
        self.grid_size = cfg.GRID_SIZE
        self.voxel_size = cfg.VOXEL_SIZE
        self.enlarge_ratio = cfg.ENLARGE_RATIO
        self.mlps = cfg.MLPS
        self.reg_fc = cfg.get("REG_FC", [256, 256])
        dp_ratio = cfg.get("DP_RATIO", 0.3)
        self.test_score_thr = cfg.get("TEST_SCORE_THR", 0.01)
        self.test_iou_thr = cfg.get("TEST_IOU_THR", 0.5)
        self.encode_angle_by_sincos = cfg.get("ENCODE_SINCOS", False)
        self.use_iou_loss = cfg.get("USE_IOU_LOSS", False)
        self.use_simple_pooling = cfg.get("USE_SIMPLE_POOLING", True)
        self.use_center_pooling = cfg.get("USE_CENTER_POOLING", True)
        self.loss_weight = cfg.LOSS_WEIGHTS
        self.cls_loss_type, self.reg_loss_type = cls_loss_type, reg_loss_type
        assert self.use_simple_pooling and self.use_center_pooling, "only the shipped pooling mode is built"
        if self.use_iou_loss:
            self.iou_loss_computer = IoU3DLoss(loss_weight=1.0, with_yaw=self.code_size > 6)
        self.proposal_target_layer = ProposalTargetLayer(roi_per_image=cfg.get("ROI_PER_IMAGE", 128),
                                                         fg_ratio=cfg.get("ROI_FG_RATIO", 0.9),
                                                         reg_fg_thresh=cfg.get("REG_FG_THRESH", 0.3))
        self.box_coder = ResidualCoder(code_size=self.code_size, encode_angle_by_sincos=self.encode_angle_by_sincos)
        self.reg_loss_func = WeightedSmoothL1Loss(code_weights=self.loss_weight.CODE_WEIGHT)
        self.roi_grid_pool_layers = nn.ModuleList([
            SimplePoolingLayer(channels=mlp, grid_kernel_size=cfg.get("ROI_CONV_KERNEL", 5), grid_num=self.grid_size,
                               voxel_size=self.voxel_size * cfg.COORD_KEY, coord_key=cfg.COORD_KEY, pooling=True)
            for mlp in self.mlps])
        pre = sum(x[-1] for x in self.mlps)
        fc = []
        for k, width in enumerate(self.reg_fc):
            fc += [nn.Linear(pre, width, bias=False), nn.BatchNorm1d(width), nn.ReLU()]
            pre = width
            if k != len(self.reg_fc) - 1 and dp_ratio > 0:
                fc.append(nn.Dropout(dp_ratio))
        self.reg_fc_layers = nn.Sequential(*fc)
        self.reg_pred_layer = nn.Linear(pre, self.code_size + (1 if self.encode_angle_by_sincos else 0), bias=True)
        self.init_weights()

    def init_weights(self):
        for m in self.reg_fc_layers.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        nn.init.normal_(self.reg_pred_layer.weight, mean=0, std=0.001)
        nn.init.constant_(self.reg_pred_layer.bias, 0)

    # ------------------------------------------------------------------ RoI grid pooling
    @staticmethod
    def get_dense_grid_points(rois, batch_size_rcnn, grid_size):
        """(R,7) -> (R, G^3, 3) cell centres in the box frame, index order (ix, iy, iz)."""
        g = torch.arange(grid_size, device=rois.device, dtype=rois.dtype)
        idx = torch.stack(torch.meshgrid(g, g, g, indexing="ij"), dim=-1).view(1, -1, 3)
        size = rois.view(batch_size_rcnn, -1)[:, 3:6].unsqueeze(1)
        return (idx + 0.5) / grid_size * size - size / 2

    def get_global_grid_points_of_roi(self, rois, grid_size):
        rois = rois.view(-1, rois.shape[-1])
        local = self.get_dense_grid_points(rois, rois.shape[0], grid_size)
        glob = local
        if self.code_size > 6:
            glob = common_utils.rotate_points_along_z(local.clone(), rois[:, 6]).squeeze(dim=1)
        return glob + rois[:, 0:3].unsqueeze(dim=1), local

    def roi_grid_pool(self, input_dict):
        rois, bs = input_dict["rois"], input_dict["batch_size"]
        feats = [input_dict["middle_feature_list"][i] for i in self.middle_feature_source]
        if FUSED_ROI and rois.dtype == torch.float32 and rois.shape[-1] == 7 and rois.numel() > 0:
            flat = rois.reshape(-1, 7)
            return torch.cat([layer.forward_rois(t, flat, rois.shape[1], self.code_size > 6)
                              for layer, t in zip(self.roi_grid_pool_layers, feats)], dim=-1)
        xyz, _ = self.get_global_grid_points_of_roi(rois, grid_size=self.grid_size)
        xyz = xyz.view(bs, -1, 3)
        bidx = torch.arange(bs, device=xyz.device, dtype=xyz.dtype).view(bs, 1, 1).expand(-1, xyz.shape[1], 1)
        grid_points = torch.cat([bidx, xyz], dim=-1).reshape(-1, 4)
        return torch.cat([layer(t, grid_points=grid_points) for layer, t in zip(self.roi_grid_pool_layers, feats)], dim=-1)

    # ------------------------------------------------------------------ train / test
    def reoder_rois_for_refining(self, pred_boxes_3d):
        """List[(boxes, scores, labels[, sem])] -> zero-padded batch tensors (cagroup_roi_head.py:328-362)."""
        bs = len(pred_boxes_3d)
        n_max = max(1, max(len(p[0]) for p in pred_boxes_3d))
        ref = pred_boxes_3d[0][0]
        use_sem = len(pred_boxes_3d[0]) == 4
        rois = ref.new_zeros((bs, n_max, ref.shape[-1]))
        scores = ref.new_zeros((bs, n_max))
        labels = ref.new_zeros((bs, n_max)).long()
        sem = ref.new_zeros((bs, n_max, pred_boxes_3d[0][3].shape[-1])) if use_sem else None
        for i, p in enumerate(pred_boxes_3d):
            n = len(p[0])
            rois[i, :n], scores[i, :n], labels[i, :n] = p[0], p[1], p[2]
            if use_sem:
                sem[i, :n] = p[3]
        rois[..., 6] *= -1                                      # to pcdet heading (:358)
        return (rois, scores, labels, sem, bs) if use_sem else (rois, scores, labels, bs)

    def _refine(self, input_dict):
        pooled = self.roi_grid_pool(input_dict)
        pooled = pooled.view(pooled.shape[0], -1)
        return self.reg_pred_layer(self._fc(pooled))

    def _fc(self, x):
        """reg_fc_layers (Linear, BatchNorm1d, ReLU[, Dropout]) with BatchNorm + ReLU as the fused rows form (ME.fused_bn_act):
        two launches instead of torch's chain, and the same cross-rank statistics path as every other BatchNorm of the model
        under --sync_bn (torch's own SyncBatchNorm refuses host tensors, so the gloo tests could not run it)."""
        mods = list(self.reg_fc_layers)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.modules.batchnorm._BatchNorm) and x.shape[1] % 4 == 0 and (x.shape[0] > 0 or isinstance(m, nn.SyncBatchNorm)):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = ME.fused_bn_act(x, [m], act=ME.ACT_RELU if relu else ME.ACT_NONE)
                i += 2 if relu else 1
            elif isinstance(m, nn.Linear):
                x = ME.linear_t(x, m.weight, m.bias)
                i += 1
            else:
                x = m(x)
                i += 1
        return x

    def _forward_train_fused(self, input_dict, flat):
        """forward_train on the FLAT proposals of the dense head (cagroup_head.get_bboxes_batched): matching, the padded index
        space of reoder_rois_for_refining, the sampled gathers, the canonical-frame targets and the regression targets are
        three launches around the one host read the sampling needs (ops/roi_stage.py); the reference's path (below) is ~120."""
        from .... import _lib
        from ....ops import roi_stage as RS
        boxes, scores, labels, per_scene = flat
        ptl = self.proposal_target_layer
        bs = len(per_scene)
        gt = input_dict["gt_boxes"].contiguous()
        dev = gt.device
        rin = max(1, max(per_scene))
        tab = ME.h2d(np.concatenate([np.cumsum([0] + list(per_scene)), np.asarray(input_dict["gt_bboxes_3d"].prefix_counts)]),
                     torch.int32, dev)
        roi_off, n_gt = tab[:bs + 1], tab[bs + 1:]
        enlarge = float(self.enlarge_ratio) if self.enlarge_ratio else 1.0
        with torch.no_grad():
            max_ov, assign = RS.roi_match(boxes, labels, roi_off, bs, rin, enlarge, gt, n_gt)
            ov = max_ov.view(bs, rin).cpu().numpy()                         # the only host read of the stage
            keep = np.concatenate([RS.subsample_rois_host(ov[i], ptl.roi_per_image, ptl.fg_ratio, ptl.reg_fg_thresh,
                                                          ptl.cls_fg_thresh, ptl.cls_bg_thresh_l0, ptl.hard_bg_ratio)
                                   for i in range(bs)])
            t = RS.roi_targets(boxes, scores, labels, roi_off, bs, rin, enlarge, gt, max_ov, assign,
                               ME.h2d(keep, torch.int32, dev), ptl.roi_per_image,
                               self.code_size + (1 if self.encode_angle_by_sincos else 0), ptl.reg_fg_thresh,
                               ptl.cls_fg_thresh, ptl.cls_bg_thresh)
        input_dict.update(batch_size=bs)
        input_dict.update(t)
        input_dict["rcnn_reg"] = self._refine(input_dict)
        return input_dict

    def _fused_train_ok(self, input_dict):
        flat, gtl = input_dict.get("pred_bbox_flat"), input_dict.get("gt_bboxes_3d")
        return (FUSED_ROI and flat is not None and getattr(gtl, "prefix_counts", None) is not None and "gt_boxes" in input_dict
                and all(n > 0 for n in gtl.prefix_counts) and (self.code_size == 6 and not self.encode_angle_by_sincos or self.code_size == 7)
                and flat[0].shape[0] > 0 and flat[0].shape[1] == 7 and input_dict["gt_boxes"].shape[-1] >= 8
                and input_dict["gt_boxes"].dtype == torch.float32)

    def forward_train(self, input_dict):
        if self._fused_train_ok(input_dict):
            return self._forward_train_fused(input_dict, input_dict["pred_bbox_flat"])
        res = self.reoder_rois_for_refining(input_dict["pred_bbox_list"])
        rois, roi_scores, roi_labels, bs = res[0], res[1], res[2], res[-1]
        if self.enlarge_ratio:
            rois[..., 3:6] *= self.enlarge_ratio
        input_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels, batch_size=bs)
        input_dict.update(self.assign_targets(input_dict))
        input_dict["rcnn_reg"] = self._refine(input_dict)
        return input_dict

    def assign_targets(self, input_dict):
        """Sample RoIs and express their GT boxes in the RoI's canonical frame (:291-326)."""
        with torch.no_grad():
            t = self.proposal_target_layer(input_dict)
        bs = input_dict["batch_size"]
        rois, gt = t["rois"], t["gt_of_rois"]
        t["gt_of_rois_src"] = gt.clone().detach()
        roi_ry = rois[:, :, 6] % (2 * np.pi)
        gt[:, :, 6] = gt[:, :, 6] % (2 * np.pi)
        gt[:, :, 0:3] = gt[:, :, 0:3] - rois[:, :, 0:3]
        gt[:, :, 6] = gt[:, :, 6] - roi_ry
        if self.code_size > 6:
            gt = common_utils.rotate_points_along_z(points=gt.view(-1, 1, gt.shape[-1]), angle=-roi_ry.view(-1)
                                                    ).view(bs, -1, gt.shape[-1])
            h = gt[:, :, 6] % (2 * np.pi)
            opp = (h > np.pi * 0.5) & (h < np.pi * 1.5)
            h[opp] = (h[opp] + np.pi) % (2 * np.pi)
            over = h > np.pi
            h[over] = h[over] - np.pi * 2
            gt[:, :, 6] = torch.clamp(h, min=-np.pi / 2, max=np.pi / 2)
        t["gt_of_rois"] = gt
        return t

    def simple_test(self, input_dict):
        pred = input_dict["pred_bbox_list"]
        if len(pred[0]) == 4:
            rois, roi_scores, roi_labels, sem, bs = self.reoder_rois_for_refining(pred)
            input_dict["roi_sem_scores"] = sem
        else:
            rois, roi_scores, roi_labels, bs = self.reoder_rois_for_refining(pred)
        input_dict.update(rois=rois, roi_scores=roi_scores, roi_labels=roi_labels, batch_size=bs)
        input_dict["rcnn_reg"] = self._refine(input_dict)
        results = self.get_boxes(input_dict, [None] * bs)
        input_dict.update(batch_box_preds=[r[0] for r in results], batch_score_preds=[r[1] for r in results],
                          batch_cls_preds=[r[2] for r in results])
        return input_dict

    def get_boxes(self, input_dict, img_meta):
        bs = input_dict["batch_size"]
        _, box_preds = self.generate_predicted_boxes(batch_size=bs, rois=input_dict["rois"], cls_preds=None,
                                                     box_preds=input_dict["rcnn_reg"], roi_labels=input_dict["roi_labels"])
        input_dict["cls_preds_normalized"] = False
        # final score = stage-1 score; stage 2 only refines the box (cagroup_roi_head.py:425)
        return [self._nms(box_preds[b], input_dict["roi_scores"][b], input_dict["roi_labels"][b], img_meta[b])
                for b in range(bs)]

    def _nms(self, bboxes, scores, labels, img_meta):
        """Per-class NMS of the refined boxes (cagroup_roi_head.py:433-475)."""
        yaw_flag = bboxes.shape[1] == 7
        out_b, out_s, out_l = [], [], []
        nonzero = bboxes.sum() != 0          # scalar over ALL boxes, as in the reference (:440,442)
        cand = (scores > self.test_score_thr) & nonzero if scores.ndim == 1 else None
        present = torch.unique(labels[cand]).tolist() if cand is not None else range(self.num_class)
        for i in present:
            if scores.ndim == 2:
                ids = (labels == i) & (scores[:, i] > self.test_score_thr) & nonzero
                if not ids.any():
                    continue
                cs = scores[ids, i]
            else:
                ids = (labels == i) & cand
                cs = scores[ids]
            cb = bboxes[ids]
            if yaw_flag:
                keep, _ = nms_gpu(cb, cs, self.test_iou_thr)
            else:
                cb = torch.cat((cb, torch.zeros_like(cb[:, :1])), dim=1)
                keep, _ = nms_normal_gpu(cb, cs, self.test_iou_thr)
            out_b.append(cb[keep]); out_s.append(cs[keep])
            out_l.append(bboxes.new_full(cs[keep].shape, i, dtype=torch.long))
        if out_b:
            nb, ns, nl = torch.cat(out_b, dim=0), torch.cat(out_s, dim=0), torch.cat(out_l, dim=0)
        else:
            nb, ns, nl = bboxes.new_zeros((0, bboxes.shape[1])), bboxes.new_zeros((0,)), bboxes.new_zeros((0,))
        if yaw_flag:
            nb[..., 6] *= -1                 # back to the original heading convention (:470)
        else:
            nb = torch.cat([nb[:, :6], nb.new_zeros(nb.shape[0], 1)], dim=1)
        return nb, ns, nl

    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds, roi_labels=None, gt_bboxes_3d=None,
                                 gt_labels_3d=None, roi_sem_scores=None):
        """Decode residuals against the RoI (size-only anchor), rotate/translate back (:477-510)."""
        cs = self.code_size
        enc = box_preds.view(batch_size, -1, cs + 1 if self.encode_angle_by_sincos else cs)
        local_rois = rois.clone().detach()[..., :cs]
        local_rois[:, :, 0:3] = 0
        boxes = self.box_coder.decode_torch(enc, local_rois).view(-1, cs)
        if cs > 6:
            boxes = common_utils.rotate_points_along_z(boxes.unsqueeze(dim=1), rois[:, :, 6].view(-1)).squeeze(dim=1)
        boxes[:, 0:3] += rois[:, :, 0:3].reshape(-1, 3)
        return None, boxes.view(batch_size, -1, cs)

    # ------------------------------------------------------------------ loss
    def loss(self, input_dict):
        if not self.use_iou_loss:
            reg, _ = self.get_box_reg_layer_loss(input_dict)
            parts = {"rcnn_loss_reg": reg}
        else:
            reg, iou, _ = self.get_box_reg_layer_loss(input_dict)
            parts = {"rcnn_loss_iou": iou}
            if self.loss_weight.RCNN_REG_WEIGHT > 0:
                parts = {"rcnn_loss_reg": reg, "rcnn_loss_iou": iou}
        total = sum(parts.values())
        keys = list(parts)
        return total, DeferredLog(keys + ["loss_two_stage"], torch.stack([parts[k] for k in keys] + [total]))

    def get_box_reg_layer_loss(self, d):
        """Smooth-L1 on encoded residuals of foreground RoIs (+ rotated IoU loss) (:551-615)."""
        cs = self.code_size
        assert self.reg_loss_type == "smooth-l1"
        fused_reg = None
        if FUSED_ROI and d.get("reg_targets") is not None:
            # targets encoded by cg3d_roi_targets; mask, code weights, smooth-L1, sum and normalisation in one launch each way
            from ....ops.roi_stage import roi_reg_loss
            cw = self.reg_loss_func.code_weights
            if cw is not None and cw.device != d["rcnn_reg"].device:
                cw = cw.to(d["rcnn_reg"].device)
            fused_reg = roi_reg_loss(d["rcnn_reg"].view(-1, d["rcnn_reg"].shape[-1]), d["reg_targets"], d["reg_valid_mask"].view(-1),
                                     cw, self.reg_loss_func.beta, float(self.loss_weight.RCNN_REG_WEIGHT))
            if not self.use_iou_loss:
                return fused_reg, {}
        fg = d["reg_valid_mask"].view(-1) > 0
        gt_ct = d["gt_of_rois"][..., 0:cs]
        gt_src = d["gt_of_rois_src"][..., 0:cs].view(-1, cs)
        rcnn_reg = d["rcnn_reg"]
        roi_boxes = d["rois"][..., 0:cs]
        n = gt_ct.view(-1, cs).shape[0]
        fg_sum = fg.long().sum()
        assert self.reg_loss_type == "smooth-l1"
        if fused_reg is not None:
            loss_reg = fused_reg
        else:
            anchors = roi_boxes.clone().detach().view(-1, cs)
            anchors[:, 0:3] = 0
            if cs > 6:
                anchors[:, 6] = 0
            targets = self.box_coder.encode_torch(gt_ct.view(n, cs), anchors)
            l = self.reg_loss_func(rcnn_reg.view(n, -1).unsqueeze(dim=0), targets.unsqueeze(dim=0))
            loss_reg = (l.view(n, -1) * fg.unsqueeze(dim=-1).float()).sum() / fg_sum.clamp(min=1)
            loss_reg = loss_reg * self.loss_weight.RCNN_REG_WEIGHT
        if not self.use_iou_loss:
            return loss_reg, {}
        loss_iou = torch.tensor(0., device=fg.device)
        if int(fg_sum) > 0:
            fg_reg = rcnn_reg.view(n, -1)[fg]
            fg_rois = roi_boxes.reshape(-1, cs)[fg].view(1, -1, cs)
            anc = fg_rois.clone().detach()
            xyz = fg_rois[:, :, 0:3].view(-1, 3)
            anc[:, :, 0:3] = 0
            boxes = self.box_coder.decode_torch(fg_reg.view(1, -1, cs + 1 if self.encode_angle_by_sincos else cs), anc).view(-1, cs)
            if cs > 6:
                boxes = common_utils.rotate_points_along_z(boxes.unsqueeze(dim=1), anc[:, :, 6].view(-1)).squeeze(dim=1)
            boxes[:, 0:3] += xyz
            loss_iou = self.iou_loss_computer(boxes[:, 0:cs], gt_src[fg][:, 0:cs]) * self.loss_weight.RCNN_IOU_WEIGHT
        return loss_reg, loss_iou, {}

    def forward(self, input_dict):
        return self.forward_train(input_dict) if self.training else self.simple_test(input_dict)
