"""Indoor (ScanNet / SUN RGB-D) detection evaluator: per-class AP / recall at 3D-IoU thresholds and their means.

Mirror of the reference's `pcdet/datasets/scannet/scannet_object_eval_python/eval.py:6-331` (`indoor_eval`,
`eval_map_recall`, `eval_det_cls`, `average_precision`, `d3_box_overlap`) -- same inputs, same result keys, same
matching rules (detections ranked by score over the whole split, greedy one-to-one matching to the GT box of
highest IoU, `>` comparisons, area-under-curve AP) -- restructured around arrays instead of nested Python dicts
and loops: one pairwise-overlap launch per (class, scene) on the bound C-ABI library (the reference runs a
numba-CUDA rotated-IoU kernel, `kitti_object_eval_python/rotate_iou.py:262-330`, then a Python double loop), a
vectorised height/union step in float32 in the reference's operation order, and one ranked pass per class.
"""
import numpy as np
import torch

from ...ops import iou3d_nms_utils
from ... import _lib


def d3_box_overlap(boxes, qboxes, criterion=-1):
    """(N,7) x (K,7) float32 -> (N,K) float32 3D IoU (eval.py:6-42): rotated BEV intersection area
    x height overlap over the union volume (criterion -1), `boxes`' / `qboxes`' volume (0 / 1) or 1 (else)."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 7)
    qboxes = np.ascontiguousarray(qboxes, dtype=np.float32).reshape(-1, 7)
    if len(boxes) == 0 or len(qboxes) == 0:
        return np.zeros((len(boxes), len(qboxes)), dtype=np.float32)
    dev = _lib.get().device_type
    rinc = iou3d_nms_utils.boxes_overlap_bev(torch.from_numpy(boxes).to(dev), torch.from_numpy(qboxes).to(dev)).cpu().numpy()
    rinc = rinc.astype(np.float32)
    half = np.float32(2.)
    top1, bot1 = boxes[:, 2] + boxes[:, 5] / half, boxes[:, 2] - boxes[:, 5] / half
    top2, bot2 = qboxes[:, 2] + qboxes[:, 5] / half, qboxes[:, 2] - qboxes[:, 5] / half
    iw = np.maximum(np.minimum(top1[:, None], top2[None, :]) - np.maximum(bot1[:, None], bot2[None, :]), np.float32(0))
    vol1 = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
    vol2 = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
    inc = iw * rinc
    if criterion == -1:
        ua = vol1 + vol2 - inc
    elif criterion == 0:
        ua = np.broadcast_to(vol1, inc.shape)
    elif criterion == 1:
        ua = np.broadcast_to(vol2, inc.shape)
    else:
        ua = inc
    touched = (rinc > 0) & (iw > 0)              # the reference only rewrites entries with BEV overlap, zeroes those without height overlap
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(touched, inc / ua, np.where(rinc > 0, np.float32(0), rinc))
    return out.astype(np.float32)


def average_precision(recalls, precisions, mode="area"):
    """AP of one or several (rows) precision/recall curves (eval.py:44-87): area under the monotone envelope,
    or the 11-point mean."""
    recalls, precisions = np.atleast_2d(recalls), np.atleast_2d(precisions)
    assert recalls.shape == precisions.shape and recalls.ndim == 2
    n = recalls.shape[0]
    ap = np.zeros(n, dtype=np.float32)
    if mode == "area":
        z, o = np.zeros((n, 1), dtype=recalls.dtype), np.ones((n, 1), dtype=recalls.dtype)
        mrec = np.hstack((z, recalls, o))
        mpre = np.hstack((z, precisions, z))
        mpre = np.maximum.accumulate(mpre[:, ::-1], axis=1)[:, ::-1]       # envelope: running max from the right
        for i in range(n):
            step = np.where(mrec[i, 1:] != mrec[i, :-1])[0]
            ap[i] = np.sum((mrec[i, step + 1] - mrec[i, step]) * mpre[i, step + 1])
    elif mode == "11points":
        for i in range(n):
            for thr in np.arange(0, 1 + 1e-3, 0.1):
                sel = precisions[i, recalls[i, :] >= thr]
                ap[i] += sel.max() if sel.size > 0 else 0
            ap /= 11        # the reference divides inside the loop over curves (eval.py:82); kept for identical numbers
    else:
        raise ValueError('Unrecognized mode, only "area" and "11points" are supported')
    return ap


def eval_det_cls(pred, gt, iou_thr=None):
    """One class.  pred: {scene: [(box[7], score), ...]}, gt: {scene: [box[7], ...]} -> per threshold
    (recall curve, precision curve, AP) (eval.py:90-188)."""
    gt_boxes, npos = {}, 0
    for sid, boxes in gt.items():
        gt_boxes[sid] = np.asarray(boxes, dtype=np.float32).reshape(-1, 7) if len(boxes) else np.zeros((0, 7), np.float32)
        npos += len(gt_boxes[sid])
    scene_of, conf, best_iou, best_gt = [], [], [], []
    for sid, dets in pred.items():
        if len(dets) == 0:
            continue
        boxes = np.stack([np.asarray(b, dtype=np.float32).reshape(-1)[:7] for b, _ in dets]).astype(np.float32)
        g = gt_boxes[sid]
        if len(g) > 0:
            iou = d3_box_overlap(boxes, g)
            arg = iou.argmax(axis=1)                       # first maximum, like the reference's strict '>' scan
            top = iou[np.arange(len(boxes)), arg]
        else:
            arg = np.zeros(len(boxes), dtype=np.int64)     # the reference compares the dummy IoU 0 of a one-entry row
            top = np.full(len(boxes), -np.inf, dtype=np.float32)
        scene_of += [sid] * len(dets)
        conf += [s for _, s in dets]
        best_iou.append(top)
        best_gt.append(arg)
    conf = np.array(conf)
    order = np.argsort(-conf)
    nd = len(order)
    if nd:
        best_iou, best_gt = np.concatenate(best_iou)[order], np.concatenate(best_gt)[order]
    scene_of = [scene_of[i] for i in order]
    out = []
    for thr in iou_thr:
        taken = {sid: np.zeros(len(b), dtype=bool) for sid, b in gt_boxes.items()}
        tp, fp = np.zeros(nd), np.zeros(nd)
        for d in range(nd):
            hit = taken[scene_of[d]]
            if best_iou[d] > thr and not hit[best_gt[d]]:
                tp[d] = 1.
                hit[best_gt[d]] = True
            else:
                fp[d] = 1.
        ctp, cfp = np.cumsum(tp), np.cumsum(fp)
        with np.errstate(divide="ignore", invalid="ignore"):
            recall = ctp / float(npos)
        precision = ctp / np.maximum(ctp + cfp, np.finfo(np.float64).eps)
        out.append((recall, precision, average_precision(recall, precision)))
    return out


def eval_map_recall(pred, gt, ovthresh=None):
    """pred / gt: {class: {scene: [...]}} -> (recall, precision, ap), each a list over thresholds of {class: value}
    (eval.py:191-224).  Classes without predictions get zeros."""
    per_class = {c: eval_det_cls(pred[c], gt[c], ovthresh) for c in gt if c in pred}
    recall, precision, ap = [{} for _ in ovthresh], [{} for _ in ovthresh], [{} for _ in ovthresh]
    for c in gt:
        for i in range(len(ovthresh)):
            if c in pred:
                recall[i][c], precision[i][c], ap[i][c] = per_class[c][i]
            else:
                recall[i][c], precision[i][c], ap[i][c] = np.zeros(1), np.zeros(1), np.zeros(1)
    return recall, precision, ap


def _table(header, columns):
    rows = [header] + [list(r) for r in zip(*columns)]
    width = [max(len(str(r[i])) for r in rows) for i in range(len(header))]
    line = "+" + "+".join("-" * (w + 2) for w in width) + "+"
    fmt = lambda r: "| " + " | ".join(str(v).ljust(w) for v, w in zip(r, width)) + " |"
    body = [line, fmt(rows[0]), line] + [fmt(r) for r in rows[1:-1]] + [line, fmt(rows[-1]), line]
    return "\n".join(body)


def indoor_eval(gt_annos, dt_annos, metric, label2cat, logger=None, box_type_3d=None, box_mode_3d=None):
    """gt_annos[i]: {'gt_num', 'gt_boxes_upright_depth' [G,6|7], 'class' [G]}; dt_annos[i]: {'boxes_3d' [N,7],
    'scores_3d' [N], 'labels_3d' [N]}; metric: IoU thresholds -> {'<cat>_AP_<thr>', '<cat>_rec_<thr>', 'mAP_<thr>',
    'mAR_<thr>'} (eval.py:227-331)."""
    assert len(dt_annos) == len(gt_annos)
    pred, gt = {}, {}
    for sid, (det, ann) in enumerate(zip(dt_annos, gt_annos)):
        labels = np.asarray(torch.as_tensor(det["labels_3d"]).cpu()).reshape(-1)
        boxes = np.asarray(torch.as_tensor(det["boxes_3d"]).cpu(), dtype=np.float32).reshape(len(labels), 7)
        scores = np.asarray(torch.as_tensor(det["scores_3d"]).cpu()).reshape(-1)
        for lab, box, score in zip(labels, boxes, scores):
            lab = int(lab)
            pred.setdefault(lab, {}).setdefault(sid, []).append((box, score))
            gt.setdefault(lab, {}).setdefault(sid, [])          # a predicted class is evaluated even where it has no GT
        if ann["gt_num"] != 0:
            g = np.asarray(ann["gt_boxes_upright_depth"], dtype=np.float32)
            if g.shape[-1] == 6:
                g = np.concatenate((g, np.zeros((g.shape[0], 1), dtype=np.float32)), axis=-1)
            elif g.shape[-1] != 7:
                raise NotImplementedError
            for lab, box in zip(np.asarray(ann["class"]).reshape(-1), g):
                gt.setdefault(int(lab), {}).setdefault(sid, []).append(box)
    rec, prec, ap = eval_map_recall(pred, gt, metric)
    ret = {}
    header, columns = ["classes"], [[label2cat[c] for c in ap[0]] + ["Overall"]]
    for i, thr in enumerate(metric):
        header += ["AP_%.2f" % thr, "AR_%.2f" % thr]
        for c in ap[i]:
            ret["%s_AP_%.2f" % (label2cat[c], thr)] = float(np.ravel(ap[i][c])[0])
        ret["mAP_%.2f" % thr] = float(np.mean(list(ap[i].values())))
        columns.append(["%.4f" % float(np.ravel(v)[0]) for v in ap[i].values()] + ["%.4f" % ret["mAP_%.2f" % thr]])
        last = []
        for c in rec[i]:
            ret["%s_rec_%.2f" % (label2cat[c], thr)] = float(rec[i][c][-1])
            last.append(rec[i][c][-1])
        ret["mAR_%.2f" % thr] = float(np.mean(last))
        columns.append(["%.4f" % float(v) for v in last] + ["%.4f" % ret["mAR_%.2f" % thr]])
    text = "\n" + _table(header, columns)
    (logger.info if hasattr(logger, "info") else print)(text)
    return ret
