"""Inference (model.eval()) throughput on synthetic S50k scenes: proposals -> RoI head -> per-class NMS (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
model, cfg = bench.make_model("scannet", True, "cuda")
model.eval()
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
with torch.no_grad():
    for _ in range(3):
        preds, _ = model(bench.fresh(batch))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        preds, _ = model(bench.fresh(batch))
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("eval: %.1f ms per 4-scene batch = %.1f scenes/s; boxes per scene %s" % (dt * 1e3, 4 / dt, [int(p["pred_boxes"].shape[0]) for p in preds]))
