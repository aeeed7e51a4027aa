"""Generate golden input/output vectors by IMPORTING the reference's pure-torch modules here
(this container only; /root/reference never travels).  Run:  python tests/golden/make_fixtures.py

Importable reference pieces: cagroup_utils (coder, rotation), cagroup3d_assigner,
loss_utils, iou3d_loss (axis-aligned), rotated_iou torch half, cagroup_proposal_target_layer,
common_utils.rotate_points_along_z, CAGroup3DHead._bbox_pred_to_bbox, and -- via oracle/_ref --
the compiled reference iou3d_cpu.cpp.  The package __init__ files are NOT executed (they pull in
MinkowskiEngine / spconv / CUDA extensions); bare module objects with the right __path__ stand in for the
packages, and absent third-party modules are stubbed.  Native ops the reference modules call
(sort_vertices, boxes_iou3d_gpu) are bound to THIS repo's CPU oracle so that only the reference's torch
logic around them is exercised.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

np.int = int          # removed in numpy >= 1.24; the reference still uses them
np.long = np.int64


def _pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m
    return m


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def setup_reference_imports(oracle):
    from cagroup3d_amd import _lib
    from cagroup3d_amd.ops import iou3d_nms_utils as my_iou, rotated_iou as my_rot

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                self[k] = EasyDict(v) if isinstance(v, dict) else v
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    _stub("easydict", EasyDict=EasyDict)
    _stub("turtle", forward=None)
    _stub("SharedArray")
    for p in ("pcdet", "pcdet/models", "pcdet/models/model_utils", "pcdet/models/dense_heads",
              "pcdet/models/dense_heads/target_assigner", "pcdet/models/roi_heads",
              "pcdet/models/roi_heads/target_assigner", "pcdet/utils", "pcdet/ops", "pcdet/ops/rotated_iou",
              "pcdet/ops/rotated_iou/cuda_op", "pcdet/ops/iou3d_nms", "pcdet/ops/knn"):
        _pkg(p.replace("/", "."), os.path.join(REF, p))

    def sort_vertices_forward(vertices, mask, num_valid):
        with _lib.use_library(oracle):
            return my_rot.sort_v(vertices, mask, num_valid)
    _stub("sort_vertices", sort_vertices_forward=sort_vertices_forward)

    def boxes_iou3d_cpu_stub(a, b):
        with _lib.use_library(oracle):
            return my_iou.boxes_iou3d_gpu(a, b)
    _stub("pcdet.ops.iou3d_nms.iou3d_nms_utils", boxes_iou3d_gpu=boxes_iou3d_cpu_stub, nms_gpu=None, nms_normal_gpu=None)
    _stub("pcdet.ops.knn", knn=None)
    _pkg("pcdet.ops.roiaware_pool3d", os.path.join(REF, "pcdet/ops/roiaware_pool3d"))
    _stub("pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils")        # LiDAR RoI pooling: not on this path
    # ME stub: enough for `class CAGroup3DHead(nn.Module)` to be defined at import
    me = _stub("MinkowskiEngine")
    for n in ("MinkowskiConvolution", "MinkowskiBatchNorm", "MinkowskiELU", "MinkowskiReLU",
              "MinkowskiGenerativeConvolutionTranspose", "SparseTensor", "SparseTensorQuantizationMode"):
        setattr(me, n, object)


def main():
    from cagroup3d_amd import _lib
    from util import rand_boxes
    oracle = _lib.bind(os.path.join(ROOT, "oracle", "liboracle.so"))
    setup_reference_imports(oracle)
    out = {}
    g = torch.Generator().manual_seed(1234)

    # ---- cagroup_utils: rotation, coder
    from pcdet.models.model_utils import cagroup_utils as ref_cu
    pts = torch.randn(5, 7, 3, generator=g)
    ang = torch.randn(5, generator=g)
    for ax in (0, 1, 2):
        out["rot3d_axis%d" % ax] = ref_cu.rotation_3d_in_axis(pts, ang, axis=ax).numpy()
    out["rot3d_points"], out["rot3d_angles"] = pts.numpy(), ang.numpy()
    boxes6, anchors6 = rand_boxes(40, 3)[:, :6], rand_boxes(40, 4)[:, :6]
    c6 = ref_cu.CAGroupResidualCoder(code_size=6)
    enc6 = c6.encode_torch(boxes6.clone(), anchors6.clone())
    out.update(coder6_boxes=boxes6.numpy(), coder6_anchors=anchors6.numpy(), coder6_enc=enc6.numpy(),
               coder6_dec=c6.decode_torch(enc6, anchors6.clone()).numpy())
    boxes7, anchors7 = rand_boxes(40, 5), rand_boxes(40, 6)
    c7 = ref_cu.CAGroupResidualCoder(code_size=7, encode_angle_by_sincos=True)
    enc7 = c7.encode_torch(boxes7.clone(), anchors7.clone())
    out.update(coder7_boxes=boxes7.numpy(), coder7_anchors=anchors7.numpy(), coder7_enc=enc7.numpy(),
               coder7_dec=c7.decode_torch(enc7, anchors7.clone()).numpy())
    out["bias_init_001"] = np.float64(ref_cu.bias_init_with_prob(0.01))

    # ---- assigner
    from pcdet.models.dense_heads.target_assigner import cagroup3d_assigner as ref_as
    gt = rand_boxes(9, 7, yaw=True, extent=2.0)
    gt[:, 3:6] += 0.6
    gt_labels = torch.tensor([0, 1, 1, 2, 0, 3, 2, 1, 0])
    pts_list = []
    for c in range(4):
        base = gt[gt_labels == c][:, :3]
        p = base[torch.randint(0, len(base), (300,), generator=g)] + torch.randn(300, 3, generator=g) * 0.5
        pts_list.append(p)
    cfg = sys.modules["easydict"].EasyDict(LIMIT=27, TOPK=18, N_SCALES=4)
    assigner = ref_as.CAGroup3DAssigner(cfg)
    ctr, boxes, labels = assigner.assign(pts_list, gt, gt_labels)
    sem_pts = torch.cat(pts_list)
    sem_labels, ins_labels = ref_as.CAGroup3DAssigner.assign_semantic(sem_pts, gt, gt_labels, 4)
    out.update(assign_gt=gt.numpy(), assign_gt_labels=gt_labels.numpy(),
               assign_points=np.stack([p.numpy() for p in pts_list]), assign_centerness=ctr.numpy(),
               assign_boxes=boxes.numpy(), assign_labels=labels.numpy(), assign_sem_labels=sem_labels.numpy(),
               assign_ins_labels=ins_labels.numpy(),
               assign_inside=ref_as.find_points_in_boxes(sem_pts, gt).numpy())

    # ---- losses
    from pcdet.utils import loss_utils as ref_lu
    from pcdet.ops.rotated_iou import oriented_iou_loss as ref_oi
    sys.modules["pcdet.ops.rotated_iou"].cal_iou_3d = ref_oi.cal_iou_3d     # what the skipped __init__ exports
    from pcdet.utils import iou3d_loss as ref_il
    pred = torch.randn(50, 6, generator=g)
    tgt = torch.randint(-1, 6, (50,), generator=g)
    out.update(focal_pred=pred.numpy(), focal_target=tgt.numpy(),
               focal_loss=ref_lu.FocalLoss(gamma=2.0, alpha=0.25)(pred, tgt.clone(), avg_factor=7.0).numpy())
    cp, ct = torch.randn(30, 1, generator=g), torch.rand(30, 1, generator=g)
    out.update(bce_pred=cp.numpy(), bce_target=ct.numpy(),
               bce_loss=ref_lu.CrossEntropy(use_sigmoid=True)(cp, ct, avg_factor=5.0).numpy())
    sp, st, sw = torch.randn(40, 3, generator=g) * 0.1, torch.randn(40, 3, generator=g) * 0.1, torch.rand(40, 3, generator=g)
    out.update(sl1_pred=sp.numpy(), sl1_target=st.numpy(), sl1_weight=sw.numpy(),
               sl1_loss=ref_lu.SmoothL1Loss(beta=0.04, reduction="sum")(sp, st, weight=sw).numpy())
    wp, wt = torch.randn(1, 20, 6, generator=g), torch.randn(1, 20, 6, generator=g)
    out.update(wsl1_pred=wp.numpy(), wsl1_target=wt.numpy(),
               wsl1_loss=ref_lu.WeightedSmoothL1Loss.smooth_l1_loss(wp - wt, 1.0 / 9.0).numpy())   # the module itself needs .cuda()
    b1 = torch.FloatTensor([[0, 0, 0, 10, 10, 10], [10, 10, 10, 20, 20, 20], [32, 32, 32, 38, 40, 42]])
    b2 = torch.FloatTensor([[0, 0, 0, 10, 20, 20], [0, 10, 10, 10, 19, 20], [10, 10, 10, 20, 20, 20]])
    out.update(aa_b1=b1.numpy(), aa_b2=b2.numpy(), aa_iou=ref_lu.axis_aligned_bbox_overlaps_3d(b1, b2).numpy(),
               aa_giou_aligned=ref_lu.axis_aligned_bbox_overlaps_3d(b1, b2, mode="giou", is_aligned=True).numpy())
    pa, pb = rand_boxes(30, 8, yaw=False, extent=1.0)[:, :6], rand_boxes(30, 9, yaw=False, extent=1.0)[:, :6]
    wts = torch.rand(30, generator=g)
    out.update(aaloss_pred=pa.numpy(), aaloss_target=pb.numpy(), aaloss_weight=wts.numpy(),
               aaloss=ref_il.IoU3DLoss(with_yaw=False, loss_weight=1.0)(pa, pb, weight=wts, avg_factor=3.0).numpy())

    # ---- rotated IoU (torch half of the reference + oracle sort_vertices)
    ra, rb = rand_boxes(200, 10, extent=1.5).view(1, -1, 7), rand_boxes(200, 11, extent=1.5).view(1, -1, 7)
    rb[0, :40] = ra[0, :40]
    rb[0, :40, :2] += 0.1
    out.update(riou_a=ra.numpy(), riou_b=rb.numpy(), riou_3d=ref_oi.cal_iou_3d(ra, rb).numpy(),
               riou_corners=ref_oi.box2corners_th(ra[..., [0, 1, 3, 4, 6]]).numpy())
    out["riou_loss"] = ref_il.IoU3DLoss(with_yaw=True)(ra[0], rb[0], weight=torch.ones(200), avg_factor=10.0).numpy()

    # ---- compiled reference BEV IoU (oracle/_ref)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import iou3d_cpu_ref
    a7, b7 = rand_boxes(120, 21), rand_boxes(90, 22)
    b7[:10] = a7[:10]
    b7[10:20, :6] = a7[10:20, :6]
    iou = torch.zeros(120, 90)
    iou3d_cpu_ref.boxes_iou_bev_cpu(a7.contiguous(), b7.contiguous(), iou)
    out.update(bev_a=a7.numpy(), bev_b=b7.numpy(), bev_iou_ref=iou.numpy())

    # ---- common_utils.rotate_points_along_z
    from pcdet.utils import common_utils as ref_cm
    rp, ra_ = torch.randn(6, 5, 4, generator=g), torch.randn(6, generator=g)
    out.update(rotz_points=rp.numpy(), rotz_angle=ra_.numpy(), rotz_out=ref_cm.rotate_points_along_z(rp, ra_).numpy())

    # ---- proposal target layer (reference sampling logic, two host RNG streams)
    from pcdet.models.roi_heads.target_assigner import cagroup_proposal_target_layer as ref_pt
    layer = ref_pt.ProposalTargetLayer(roi_per_image=32, fg_ratio=0.9, reg_fg_thresh=0.3)
    gtb = [rand_boxes(6, 30 + i, yaw=False, extent=2.0) for i in range(2)]
    gtl = [torch.randint(0, 3, (6,), generator=g) for _ in range(2)]
    rois = torch.zeros(2, 50, 7)
    rlab = torch.zeros(2, 50, dtype=torch.long)
    for i in range(2):
        src = torch.randint(0, 6, (50,), generator=g)
        rois[i] = gtb[i][src] + torch.randn(50, 7, generator=g) * torch.tensor([0.15, 0.15, 0.15, 0.1, 0.1, 0.1, 0.0])
        rois[i, :, 3:6] = rois[i, :, 3:6].clamp(min=0.1)
        rlab[i] = gtl[i][src]
    rscore = torch.rand(2, 50, generator=g)
    bd = dict(batch_size=2, rois=rois.clone(), roi_scores=rscore.clone(), roi_labels=rlab.clone(),
              gt_bboxes_3d=[x.clone() for x in gtb], gt_labels_3d=[x.clone() for x in gtl])
    np.random.seed(7)
    torch.manual_seed(7)
    res = layer(bd)
    out.update(ptl_rois=rois.numpy(), ptl_scores=rscore.numpy(), ptl_labels=rlab.numpy(),
               ptl_gt=np.stack([x.numpy() for x in gtb]), ptl_gt_labels=np.stack([x.numpy() for x in gtl]))
    for k in ("rois", "gt_of_rois", "gt_label_of_rois", "gt_iou_of_rois", "roi_scores", "roi_labels", "reg_valid_mask",
              "rcnn_cls_labels"):
        out["ptl_out_" + k] = res[k].numpy()

    # ---- CAGroup3DHead._bbox_pred_to_bbox (fcaf3d yaw parametrisation and plain 6-dof)
    from pcdet.models.dense_heads import cagroup_head as ref_head
    fake = types.SimpleNamespace(yaw_parametrization="fcaf3d")
    p3 = torch.randn(25, 3, generator=g)
    bp8 = torch.cat([torch.rand(25, 6, generator=g) + 0.1, torch.randn(25, 2, generator=g)], 1)
    out.update(b2b_points=p3.numpy(), b2b_pred8=bp8.numpy(),
               b2b_box8=ref_head.CAGroup3DHead._bbox_pred_to_bbox(fake, p3, bp8).numpy(),
               b2b_box6=ref_head.CAGroup3DHead._bbox_pred_to_bbox(fake, p3, bp8[:, :6]).numpy())

    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%d arrays, %.1f KB" % (len(out), os.path.getsize(path) / 1e3))


if __name__ == "__main__":
    main()
