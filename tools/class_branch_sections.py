"""Host time and device tail of every section of CAGroup3DHead._class_branches_batched in a training step (a device sync at
every section boundary: host = issue time from an empty queue, tail = what the device still had to do).  dev tool; GPU box."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench
from cagroup3d_amd import build_model, me
from cagroup3d_amd.pcdet.models.dense_heads import cagroup_head as H
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
from cagroup3d_amd.optim import ClippedAdamW
opt = ClippedAdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
batch = build_model.synthetic_batch("S50k", int(os.environ.get("CG3D_PROFILE_BATCH", "4")), device=dev)
for _ in range(6):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
host = collections.OrderedDict(); tail = collections.OrderedDict()
N = 10
for _ in range(N):
    H.TICKS = []
    bench.train_step(model, opt, batch, 10.0)
    t = H.TICKS
    for i in range(1, len(t)):
        host[t[i][0]] = host.get(t[i][0], 0.0) + (t[i][1] - t[i - 1][2])
        tail[t[i][0]] = tail.get(t[i][0], 0.0) + (t[i][2] - t[i][1])
H.TICKS = None
print("section                host ms   GPU tail ms")
for k in host:
    print(f"{k:22s} {1e3 * host[k] / N:7.3f}   {1e3 * tail[k] / N:7.3f}")
print(f"{'total':22s} {1e3 * sum(host.values()) / N:7.3f}   {1e3 * sum(tail.values()) / N:7.3f}")

bench.finish_prefetch(model)
