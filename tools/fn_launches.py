"""Device launches and host time per function of the head / RoI head / loss (dev tool, GPU only)."""
import os, sys, time, functools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from cagroup3d_amd import build_model, me
from cagroup3d_amd.pcdet.models.dense_heads import cagroup_head as H
from cagroup3d_amd.pcdet.models.roi_heads import cagroup_roi_head as R
from cagroup3d_amd.pcdet.models.dense_heads.target_assigner import cagroup3d_assigner as A
from cagroup3d_amd.pcdet.models.roi_heads.target_assigner import cagroup_proposal_target_layer as T
import bench


def wrap(obj, name):
    fn = getattr(obj, name)
    @functools.wraps(fn)
    def w(*a, **k):
        with record_function("FN_" + name):
            return fn(*a, **k)
    setattr(obj, name, w)


for name in ("_forced_selection", "_class_branches_batched", "get_bboxes_batched", "_loss_batched", "_vote_targets_masks",
             "_bbox_pred_to_bbox", "_finish_nms"):
    wrap(H.CAGroup3DHead, name)
for name in ("roi_grid_pool", "assign_targets", "_refine", "get_box_reg_layer_loss", "reoder_rois_for_refining"):
    wrap(R.CAGroup3DRoIHead, name)
wrap(R.SimplePoolingLayer, "forward")
for name in ("assign_all_classes",):
    wrap(A.CAGroup3DAssigner, name)
for name in ("sample_rois_for_rcnn", "subsample_rois"):
    wrap(T.ProposalTargetLayer, name)
wrap(H, "find_points_in_boxes")

me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(3):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    ret, tb, disp = model(bench.fresh(batch))
    torch.cuda.synchronize()
evs = prof.events()
fns = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("FN_")]
launch = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU and
          e.name in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync", "hipMemcpyWithStream",
                     "hipModuleLaunchKernel", "hipExtLaunchKernel")]
print("total launches (runtime calls) in forward:", len(launch))
for f in sorted(fns, key=lambda e: e.time_range.start):
    n = sum(1 for l in launch if f.time_range.start <= l.time_range.start <= f.time_range.end)
    print("%-34s host %7.2f ms  launches %5d" % (f.name, (f.time_range.end - f.time_range.start) / 1e3, n))
