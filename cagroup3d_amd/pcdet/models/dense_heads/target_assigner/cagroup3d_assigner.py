"""FCOS-style target assignment of CAGroup3D (mirror of
pcdet/models/dense_heads/target_assigner/cagroup3d_assigner.py:1-152; pure torch, training only)."""
import torch

from ...model_utils.cagroup_utils import rotation_3d_in_axis

FLOAT_MAX = 1e8
FUSED_ASSIGN = __import__("os").environ.get("CG3D_FUSED_ASSIGN", "1") != "0"


def volume(boxes):
    return boxes[:, 3] * boxes[:, 4] * boxes[:, 5]


def _face_distances(points, gt_bboxes):
    """points (n,3), gt (m,7) -> (n,m,7): distances to the 6 faces in the box frame + heading
    (the shared body of find_points_in_boxes :9-36 and CAGroup3DAssigner.assign :83-98)."""
    n, m = len(points), len(gt_bboxes)
    gt = gt_bboxes.to(points.device).expand(n, m, 7)
    pts = points.unsqueeze(1).expand(n, m, 3)
    shift = torch.stack((pts[..., 0] - gt[..., 0], pts[..., 1] - gt[..., 1], pts[..., 2] - gt[..., 2]),
                        dim=-1).permute(1, 0, 2)
    shift = rotation_3d_in_axis(shift, -gt[0, :, 6], axis=2).permute(1, 0, 2)
    ctr = gt[..., :3] + shift
    return torch.stack((ctr[..., 0] - gt[..., 0] + gt[..., 3] / 2, gt[..., 0] + gt[..., 3] / 2 - ctr[..., 0],
                        ctr[..., 1] - gt[..., 1] + gt[..., 4] / 2, gt[..., 1] + gt[..., 4] / 2 - ctr[..., 1],
                        ctr[..., 2] - gt[..., 2] + gt[..., 5] / 2, gt[..., 2] + gt[..., 5] / 2 - ctr[..., 2],
                        gt[..., 6]), dim=-1)


def _pairwise_face_distances(points, boxes):
    """points (n,3), boxes (n,7) -> (n,7): `_face_distances` of point i to box i only."""
    shift = (points - boxes[:, :3]).unsqueeze(0)                                # (1, n, 3)
    shift = rotation_3d_in_axis(shift.permute(1, 0, 2), -boxes[:, 6], axis=2).permute(1, 0, 2)[0]
    ctr = boxes[:, :3] + shift
    g = boxes
    return torch.stack((ctr[:, 0] - g[:, 0] + g[:, 3] / 2, g[:, 0] + g[:, 3] / 2 - ctr[:, 0],
                        ctr[:, 1] - g[:, 1] + g[:, 4] / 2, g[:, 1] + g[:, 4] / 2 - ctr[:, 1],
                        ctr[:, 2] - g[:, 2] + g[:, 5] / 2, g[:, 2] + g[:, 5] / 2 - ctr[:, 2], g[:, 6]), dim=-1)


def find_points_in_boxes(points, gt_bboxes, expanded_volumes=None, point_seg=None, box_seg=None):
    """(n,3) x (m,7) -> bool (n,m): strictly inside the (rotated) box -- one fused launch (cg3d_points_in_boxes) of the
    reference's tensor expression `_face_distances(points, gt)[..., :6].min(-1)[0] > 0`."""
    from .....ops.iou3d_nms_utils import points_in_boxes
    return points_in_boxes(points, gt_bboxes.to(points.device), point_seg, box_seg)


def compute_centerness(bbox_targets):
    """sqrt of the product over axes of min/max face distance (:39-46)."""
    x, y, z = bbox_targets[..., 0:2], bbox_targets[..., 2:4], bbox_targets[..., 4:6]
    c = x.min(dim=-1)[0] / x.max(dim=-1)[0] * y.min(dim=-1)[0] / y.max(dim=-1)[0] * z.min(dim=-1)[0] / z.max(dim=-1)[0]
    return torch.sqrt(c)


class CAGroup3DAssigner(object):
    def __init__(self, cfg):
        self.limit = cfg.LIMIT
        self.topk = cfg.TOPK
        self.n_scales = cfg.N_SCALES
        self.return_ins_label = cfg.get("RETURN_INS_LABEL", True)

    def assign(self, points_list, gt_bboxes_ori, gt_labels_ori):
        """Per class: a point is positive for the smallest-volume same-class GT box it lies in,
        restricted to the top-k most central points of that box (:62-130)."""
        ctr_all, box_all, lab_all = [], [], []
        for cls_id, points in enumerate(points_list):
            n = len(points)
            assert n > 0, "empty points in class {}".format(cls_id)
            sel = torch.nonzero(gt_labels_ori == cls_id).squeeze(1)
            if len(sel) == 0:
                lab_all.append(torch.full((n,), -1, dtype=torch.long, device=points.device))
                box_all.append(torch.zeros((n, 7), dtype=torch.float, device=points.device))
                ctr_all.append(torch.zeros((n,), dtype=torch.float, device=points.device))
                continue
            m = len(sel)
            gt = gt_bboxes_ori[sel].clone().to(points.device)
            gt_labels = gt_labels_ori[sel].clone()
            vols = volume(gt_bboxes_ori).to(points.device)[sel].expand(n, m).contiguous()
            targets = _face_distances(points, gt)
            inside = targets[..., :6].min(-1)[0] > 0
            cness = compute_centerness(targets)
            cness = torch.where(inside, cness, torch.ones_like(cness) * -1)
            kth = torch.topk(cness, min(self.topk + 1, len(cness)), dim=0).values[-1]
            in_top = cness > kth.unsqueeze(0)
            vols = torch.where(inside, vols, torch.ones_like(vols) * FLOAT_MAX)
            vols = torch.where(in_top, vols, torch.ones_like(vols) * FLOAT_MAX)
            min_vol, min_ind = vols.min(dim=1)
            labels = gt_labels[min_ind]
            labels = torch.where(min_vol == FLOAT_MAX, -labels.new_ones(labels.shape), labels)
            rows = torch.arange(n, device=points.device)
            ctr_all.append(compute_centerness(targets[rows, min_ind]))
            box_all.append(gt.expand(n, m, 7)[rows, min_ind].clone())
            lab_all.append(labels)
        return torch.cat(ctr_all), torch.cat(box_all), torch.cat(lab_all)

    def _assign_fused(self, points, pt_cls, pt_scene, gt, gt_labels, gt_scene, k):
        """`assign_all_classes` through the C-ABI (cg3d_fcos_centerness -> one top-k over the [n, m] table ->
        cg3d_fcos_assign): three launches instead of ~70."""
        from ctypes import c_int32, c_int64
        from ..... import _lib
        from ....._lib import ptr
        lib = _lib.get()
        dev = points.device
        n, m = points.shape[0], gt.shape[0]
        pts = points[:, :3].to(torch.float32).contiguous()
        gtc = gt[:, :7].to(torch.float32).contiguous()
        pc, gc = pt_cls.to(torch.int64).contiguous(), gt_labels.to(torch.int64).contiguous()
        ps = pt_scene.to(torch.int64).contiguous() if pt_scene is not None else None
        gs = gt_scene.to(torch.int64).contiguous() if gt_scene is not None else None
        lib.check(pts, gtc, pc, gc, ps, gs)
        cness = torch.empty((n, m), dtype=torch.float32, device=dev)
        lib.call("cg3d_fcos_centerness", ptr(pts), ptr(pc), ptr(ps), c_int64(n), ptr(gtc), ptr(gc), ptr(gs), c_int32(m), ptr(cness),
                 lib.stream())
        top = torch.topk(cness, min(self.topk + 1, n), dim=0).values                      # [K, m], descending
        kth = top.gather(0, (k - 1).clamp(max=top.shape[0] - 1).unsqueeze(0)).squeeze(0).contiguous()
        ctr = torch.empty(n, dtype=torch.float32, device=dev)
        box = torch.empty((n, 7), dtype=torch.float32, device=dev)
        labels = torch.empty(n, dtype=torch.int64, device=dev)
        lib.call("cg3d_fcos_assign", ptr(cness), ptr(kth), c_int64(n), ptr(gtc), ptr(gc), c_int32(m), ptr(ctr), ptr(box), ptr(labels),
                 lib.stream())
        return ctr, box, labels

    def assign_all_classes(self, points_list, gt_bboxes_ori, gt_labels_ori, pt_cls=None, same=None, n_map=None, pt_scene=None,
                           gt_scene=None):
        """`assign` for all classes (and, with `same`, all scenes) in one pass: a point of class map c only
        competes for GT boxes of class c (and of its own scene).  Same positives / targets as `assign` for every
        labelled point; rows with label -1 carry unspecified (unused) box / centerness values.
          pt_cls : class of every point (default: list position);  same : bool [n, m] extra pair mask;
          n_map  : [m] number of points on the map each GT box competes on (default: its class's point count);
          pt_scene / gt_scene : int [n] / [m], the pair mask `same` as two id vectors (same = pt_scene[:, None] == gt_scene[None])
          -- the form the fused path (`_assign_fused`) takes."""
        points = torch.cat(points_list)
        dev = points.device
        n, m = len(points), len(gt_bboxes_ori)
        if m == 0:
            return (torch.zeros(n, device=dev), torch.zeros((n, 7), device=dev),
                    torch.full((n,), -1, dtype=torch.long, device=dev))
        gt = gt_bboxes_ori.to(dev)
        gt_labels = gt_labels_ori.to(dev).long()
        if pt_cls is None:
            from ..... import me
            n_per = [len(p) for p in points_list]
            n_per_d = me.h2d(n_per, torch.long, dev)
            pt_cls = torch.repeat_interleave(torch.arange(len(points_list), device=dev), n_per_d, output_size=n)
            n_map = n_per_d[gt_labels.clamp(max=len(n_per) - 1)]
        # GT boxes are independent columns up to the final "smallest box" choice per point, so the (n, m, 7) face-distance
        # tensor and its (n, m) companions are built for at most PAIR_CHUNK pairs at a time (all of them at the bench's
        # sizes: 28 k points x 80 boxes); larger batches walk the boxes in column chunks -- the reference's per-class,
        # per-scene loop bounds its memory the same way (cagroup3d_assigner.py:62-130)
        k = torch.clamp(n_map, max=self.topk + 1).clamp(min=1)
        if FUSED_ASSIGN and gt.shape[1] >= 7 and n * m <= self.PAIR_CHUNK and n > 0 and (same is None or pt_scene is not None):
            return self._assign_fused(points, pt_cls, pt_scene, gt, gt_labels, gt_scene, k)
        if same is None and pt_scene is not None:
            same = pt_scene.view(-1, 1) == gt_scene.view(1, -1)
        vol_all = volume(gt)
        mc = m if n * m <= self.PAIR_CHUNK else max(1, self.PAIR_CHUNK // max(n, 1))
        min_vol = min_ind = None
        for c0 in range(0, m, mc):
            sl = slice(c0, min(c0 + mc, m))
            targets = _face_distances(points, gt[sl])                           # (n, mc, 7)
            inside = (targets[..., :6].min(-1)[0] > 0) & (pt_cls.unsqueeze(1) == gt_labels[sl].unsqueeze(0))
            if same is not None:
                inside = inside & same[:, sl]
            cness = compute_centerness(targets)
            cness = torch.where(inside, cness, torch.ones_like(cness) * -1)
            kth = torch.sort(cness, dim=0, descending=True)[0].gather(0, (k[sl] - 1).unsqueeze(0)).squeeze(0)
            in_top = cness > kth.unsqueeze(0)
            vols = vol_all[sl].unsqueeze(0).expand(n, sl.stop - sl.start)
            vols = torch.where(inside & in_top, vols, torch.ones_like(vols) * FLOAT_MAX)
            mv, mi = vols.min(dim=1)
            if min_vol is None:
                min_vol, min_ind = mv, mi
                first_targets = targets if mc == m else None
            else:
                better = mv < min_vol                                           # ties keep the earlier (lower-index) box, as one min() does
                min_vol = torch.where(better, mv, min_vol)
                min_ind = torch.where(better, mi + c0, min_ind)
        labels = torch.where(min_vol == FLOAT_MAX, -torch.ones_like(min_ind), gt_labels[min_ind])
        rows = torch.arange(n, device=dev)
        if first_targets is not None:
            chosen = first_targets[rows, min_ind]
        else:                                                                   # face distances to every point's own box only
            chosen = _pairwise_face_distances(points, gt[min_ind])
        return compute_centerness(chosen), gt[min_ind].clone(), labels

    PAIR_CHUNK = 1 << 23            # pairs (point, GT box) per pass of assign_all_classes: 8 M x 7 fp32 = 235 MB

    @classmethod
    def assign_semantic(cls, points, gt_bboxes, gt_labels, n_classes):
        """Semantic label of a voxel = class of the smallest GT box containing it (:132-152)."""
        n, m = len(points), len(gt_bboxes)
        vols = volume(gt_bboxes).to(points.device).expand(n, m).contiguous()
        inside = find_points_in_boxes(points, gt_bboxes)
        vols = torch.where(inside, vols, torch.ones_like(vols) * FLOAT_MAX)
        bk = inside.sum(dim=1) != 0
        min_vol, min_ind = vols.min(dim=1)
        labels = gt_labels[min_ind]
        labels = torch.where(min_vol == FLOAT_MAX, -labels.new_ones(labels.shape), labels)
        return labels, (min_ind + 1) * bk
