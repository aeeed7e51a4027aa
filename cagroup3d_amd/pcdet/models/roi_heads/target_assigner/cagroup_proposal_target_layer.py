"""RoI sampling for the second stage (mirror of
pcdet/models/roi_heads/target_assigner/cagroup_proposal_target_layer.py:8-237).

Uses the same two host RNG streams as the reference (np.random.permutation / np.random.rand for the
foreground draw, torch.randint for the background draw; :144,155,181-197), so a seeded run samples
the same RoIs."""
import numpy as np
import torch
import torch.nn as nn

from .....ops.iou3d_nms_utils import boxes_iou3d_gpu


class ProposalTargetLayer(nn.Module):
    def __init__(self, roi_per_image=128, fg_ratio=0.5, reg_fg_thresh=0.3, cls_fg_thresh=0.55, cls_bg_thresh=0.15,
                 cls_bg_thresh_l0=0.1, hard_bg_ratio=0.8):
        super().__init__()
        self.roi_per_image, self.fg_ratio = roi_per_image, fg_ratio
        self.reg_fg_thresh, self.cls_fg_thresh = reg_fg_thresh, cls_fg_thresh
        self.cls_bg_thresh, self.cls_bg_thresh_l0, self.hard_bg_ratio = cls_bg_thresh, cls_bg_thresh_l0, hard_bg_ratio

    def forward(self, batch_dict):
        rois, gt_of_rois, gt_label_of_rois, ious, scores, labels = self.sample_rois_for_rcnn(batch_dict)
        reg_valid_mask = (ious > self.reg_fg_thresh).long()
        fg, bg = ious > self.cls_fg_thresh, ious < self.cls_bg_thresh
        mid = (fg == 0) & (bg == 0)
        cls_labels = (fg > 0).float()
        cls_labels[mid] = (ious[mid] - self.cls_bg_thresh) / (self.cls_fg_thresh - self.cls_bg_thresh)
        return {"rois": rois, "gt_of_rois": gt_of_rois, "gt_label_of_rois": gt_label_of_rois, "gt_iou_of_rois": ious,
                "roi_scores": scores, "roi_labels": labels, "reg_valid_mask": reg_valid_mask,
                "rcnn_cls_labels": cls_labels}

    def sample_rois_for_rcnn(self, batch_dict):
        """All scenes at once: one pairwise-IoU launch (same-scene, same-class masked), ONE device->host read of
        the per-RoI best overlaps, the fg/bg draw on the host with the reference's RNG call sequence, one gather."""
        bs = batch_dict["batch_size"]
        rois, roi_scores, roi_labels = batch_dict["rois"], batch_dict["roi_scores"], batch_dict["roi_labels"]
        gt_boxes, gt_labels = batch_dict["gt_bboxes_3d"], batch_dict["gt_labels_3d"]
        if any(len(g) == 0 for g in gt_boxes):
            return self._sample_rois_per_scene(batch_dict)
        from .....me import h2d
        dev, R, Rin = rois.device, self.roi_per_image, rois.shape[1]
        n_gt = [len(g) for g in gt_boxes]
        gt_all = torch.cat([g for g in gt_boxes]).clone()
        gt_all[..., 6] *= -1                                      # mmdet3d heading -> pcdet heading (:97)
        gl_all = torch.cat([l for l in gt_labels])
        gt_scene = torch.repeat_interleave(torch.arange(bs, device=dev), h2d(n_gt, torch.long, dev), output_size=sum(n_gt))
        gt_first = h2d(np.cumsum([0] + n_gt[:-1]), torch.long, dev)
        roi_scene = torch.arange(bs, device=dev).repeat_interleave(Rin)
        flat_rois, flat_labels = rois.reshape(bs * Rin, -1), roi_labels.reshape(-1)
        iou = boxes_iou3d_gpu(flat_rois[:, :7].contiguous(), gt_all[:, 0:7].contiguous())          # (B*Rin, sum G)
        same = (flat_labels.view(-1, 1) == gl_all.long().view(1, -1)) & (roi_scene.view(-1, 1) == gt_scene.view(1, -1))
        best, arg = torch.max(torch.where(same, iou, torch.full_like(iou, -1.0)), dim=1)
        has = best >= 0
        max_ov = torch.where(has, best, torch.zeros_like(best))
        assign = torch.where(has, arg, gt_first[roi_scene])        # no GT of the RoI's class: the scene's first box (:204-238)
        ov_host = max_ov.view(bs, Rin).cpu()                       # the only host read
        keep = torch.stack([self.subsample_rois(ov_host[i]) for i in range(bs)])            # host, reference RNG order
        flat_keep = (h2d(keep, torch.long, dev) + torch.arange(bs, device=dev).view(-1, 1) * Rin).view(-1)
        a = assign[flat_keep]
        return (flat_rois[flat_keep].view(bs, R, -1), gt_all[a].view(bs, R, -1), gl_all[a].view(bs, R).to(rois.dtype),
                max_ov[flat_keep].view(bs, R), roi_scores.reshape(-1)[flat_keep].view(bs, R), flat_labels[flat_keep].view(bs, R))

    def _sample_rois_per_scene(self, batch_dict):
        bs = batch_dict["batch_size"]
        rois, roi_scores, roi_labels = batch_dict["rois"], batch_dict["roi_scores"], batch_dict["roi_labels"]
        gt_boxes, gt_labels = batch_dict["gt_bboxes_3d"], batch_dict["gt_labels_3d"]
        R = self.roi_per_image
        b_rois = rois.new_zeros(bs, R, rois.shape[-1])
        b_gt = rois.new_zeros(bs, R, gt_boxes[0].shape[-1])
        b_gt_label = rois.new_zeros(bs, R)
        b_iou, b_score = rois.new_zeros(bs, R), rois.new_zeros(bs, R)
        b_label = rois.new_zeros((bs, R), dtype=torch.long)
        for i in range(bs):
            cur_gt = gt_boxes[i].clone()
            cur_gt[..., 6] *= -1                                  # mmdet3d heading -> pcdet heading (:97)
            if len(cur_gt) == 0:
                cur_gt = cur_gt.new_zeros((1, cur_gt.shape[1]))
            cur_labels = gt_labels[i]
            max_ov, assign = self.get_max_iou_with_same_class(rois[i], roi_labels[i], cur_gt[:, 0:7], cur_labels.long())
            keep = self.subsample_rois(max_ov)
            b_rois[i], b_label[i], b_iou[i], b_score[i] = rois[i][keep], roi_labels[i][keep], max_ov[keep], roi_scores[i][keep]
            b_gt[i] = cur_gt[assign[keep]]
            b_gt_label[i] = cur_labels[assign[keep]]
        return b_rois, b_gt, b_gt_label, b_iou, b_score, b_label

    def subsample_rois(self, max_overlaps):
        fg_per_image = int(np.round(self.fg_ratio * self.roi_per_image))
        fg_thresh = min(self.reg_fg_thresh, self.cls_fg_thresh)
        fg_inds = (max_overlaps >= fg_thresh).nonzero().view(-1)
        easy_bg = (max_overlaps < self.cls_bg_thresh_l0).nonzero().view(-1)
        hard_bg = ((max_overlaps < self.reg_fg_thresh) & (max_overlaps >= self.cls_bg_thresh_l0)).nonzero().view(-1)
        n_fg, n_bg = fg_inds.numel(), hard_bg.numel() + easy_bg.numel()
        if n_fg > 0 and n_bg > 0:
            take = min(fg_per_image, n_fg)
            perm = torch.from_numpy(np.random.permutation(n_fg)).type_as(max_overlaps).long()
            fg_inds = fg_inds[perm[:take]]
            bg_inds = self.sample_bg_inds(hard_bg, easy_bg, self.roi_per_image - take, self.hard_bg_ratio)
        elif n_fg > 0:
            rnd = torch.from_numpy(np.floor(np.random.rand(self.roi_per_image) * n_fg)).type_as(max_overlaps).long()
            fg_inds = fg_inds[rnd]
            bg_inds = fg_inds[fg_inds < 0]
        elif n_bg > 0:
            bg_inds = self.sample_bg_inds(hard_bg, easy_bg, self.roi_per_image, self.hard_bg_ratio)
        else:
            raise NotImplementedError("no RoIs to sample: FG=%d BG=%d" % (n_fg, n_bg))
        return torch.cat((fg_inds, bg_inds), dim=0)

    @staticmethod
    def sample_bg_inds(hard_bg_inds, easy_bg_inds, n_bg, hard_bg_ratio):
        def draw(pool, k):
            return pool[torch.randint(low=0, high=pool.numel(), size=(k,)).long()]
        if hard_bg_inds.numel() > 0 and easy_bg_inds.numel() > 0:
            n_hard = min(int(n_bg * hard_bg_ratio), len(hard_bg_inds))
            hard = draw(hard_bg_inds, n_hard)
            return torch.cat([hard, draw(easy_bg_inds, n_bg - n_hard)], dim=0)
        if hard_bg_inds.numel() > 0:
            return draw(hard_bg_inds, n_bg)
        if easy_bg_inds.numel() > 0:
            return draw(easy_bg_inds, n_bg)
        raise NotImplementedError

    @staticmethod
    def get_max_iou_with_same_class(rois, roi_labels, gt_boxes, gt_labels):
        """Per RoI: best 3D IoU against GT boxes of ITS class and that GT's index (:204-238).
        One pairwise-overlap launch for all classes; RoIs whose class has no GT get (0, 0) like the
        reference's zero-initialised outputs."""
        iou = boxes_iou3d_gpu(rois.contiguous(), gt_boxes.contiguous())            # (R, G)
        same = roi_labels.view(-1, 1) == gt_labels.view(1, -1)
        best, arg = torch.max(torch.where(same, iou, torch.full_like(iou, -1.0)), dim=1)
        has = best >= 0
        return torch.where(has, best, torch.zeros_like(best)), torch.where(has, arg, torch.zeros_like(arg)).to(roi_labels.dtype)
