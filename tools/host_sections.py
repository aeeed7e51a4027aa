"""Host time of a training step by section (wall-clock around the sub-modules' forward calls, backward and the optimizer)
at batch 1, where the GPU is never the bottleneck: where the ~28 ms of Python / launch time per step go.  dev tool; GPU box."""
import collections
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402
from cagroup3d_amd.optim import ClippedAdamW  # noqa: E402

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
from cagroup3d_amd.hostpin import pin_host_threads  # noqa: E402
pin_host_threads(0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = ClippedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
batch = build_model.synthetic_batch("S50k", int(os.environ.get("BATCH", "1")), device=dev)
acc = collections.OrderedDict()


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, timed)


wrap(model, "voxelization", "voxelization (lookup of the prepared maps)")
wrap(model.backbone_3d, "forward", "backbone forward")
wrap(model.dense_head, "forward", "head forward (shared part + class branches)")
wrap(model.dense_head, "get_bboxes_batched", "  of which proposals: decode + NMS") if hasattr(model.dense_head, "get_bboxes_batched") else None
wrap(model.dense_head, "_class_branches_batched", "  of which class branches")
wrap(model.roi_head, "forward", "RoI head forward (sampling, pooling, FC)")
wrap(model, "get_training_loss", "losses (assignment, stage-1 terms, RoI terms)")
for _ in range(5):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
acc.clear()
N = 20
tb = tf = to = 0.0
orig_backward = torch.Tensor.backward


def timed_backward(self, *a, **k):
    global tb
    t0 = time.perf_counter()
    orig_backward(self, *a, **k)
    tb += time.perf_counter() - t0


torch.Tensor.backward = timed_backward
wrap(opt, "clip_and_step", "clip + AdamW")
t0 = time.perf_counter()
for _ in range(N):
    bench.train_step(model, opt, batch, 10.0)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
bench.finish_prefetch(model)
print("batch %d: %.2f ms/step issued, %.2f ms/step synchronised" % (batch["batch_size"], t_issue / N * 1e3, t_all / N * 1e3))
for k, v in acc.items():
    print("  %-55s %6.2f ms" % (k, v / N * 1e3))
print("  %-55s %6.2f ms" % ("backward (autograd engine + Python backward functions)", tb / N * 1e3))
