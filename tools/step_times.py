"""Per-step wall times of the bench loop (host clock around train_step; the step's own blocking reads keep the host within
one step of the device): is the run-to-run scatter of bench.py made of outlier steps or of a shifted level?  dev tool; GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402
from cagroup3d_amd.optim import ClippedAdamW  # noqa: E402

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
try:
    from cagroup3d_amd.hostpin import pin_host_threads
    if os.environ.get("CG3D_PIN", "1") != "0":
        pin_host_threads(0)
except Exception as e:  # noqa: BLE001
    print("no pinning:", e)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = ClippedAdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
model.split_late_parameters(opt)            # as bench.py does
batch = build_model.synthetic_batch("S50k", 4, device=dev)
if os.environ.get("CG3D_SWITCH_US"):
    sys.setswitchinterval(float(os.environ["CG3D_SWITCH_US"]) * 1e-6)
if os.environ.get("CG3D_AUTOGRAD_ST", "1") == "1":
    torch.autograd.set_multithreading_enabled(False)        # backward nodes on the calling thread
main_prio = int(os.environ.get("CG3D_MAIN_PRIORITY", "0"))
if main_prio:
    # the whole step on a high-priority stream (the dry run's stream stays at normal priority)
    torch.cuda.synchronize()
    _s = torch.cuda.Stream(priority=main_prio)
    torch.cuda.set_stream(_s)
for _ in range(6):
    bench.train_step(model, opt, batch, 10.0)
import gc
gc.collect()
gc.freeze()
if os.environ.get("CG3D_GC_MANUAL") == "1":
    gc.disable()
    _ts = bench.train_step

    def _step(*a, **k):
        gc.collect(1)                 # the young generations, at the start of the step: the issuing thread has slack there
        return _ts(*a, **k)
    bench.train_step = _step
elif os.environ.get("CG3D_GC_OFF") == "1":
    gc.disable()
elif os.environ.get("CG3D_GC_THRESHOLD"):
    gc.set_threshold(int(os.environ["CG3D_GC_THRESHOLD"]), 50, 50)
torch.cuda.synchronize()
N = int(os.environ.get("STEPS", "80"))
ts = []
t_prev = time.perf_counter()
for _ in range(N):
    bench.train_step(model, opt, batch, 10.0)
    t = time.perf_counter()
    ts.append(1e3 * (t - t_prev))
    t_prev = t
torch.cuda.synchronize()
bench.finish_prefetch(model)
a = np.array(ts)
print("mean %.2f  median %.2f  p10 %.2f  p90 %.2f  max %.2f  | first 10: %s" %
      (a.mean(), np.median(a), np.percentile(a, 10), np.percentile(a, 90), a.max(), np.round(a[:10], 1)))
print("by block of 10:", np.round(a[: N // 10 * 10].reshape(-1, 10).mean(1), 2))
