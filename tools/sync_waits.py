"""Where the issuing thread waits for the device in a training step: every blocking read (.tolist / .cpu / .item / .numpy on a
device tensor) on the main thread, by call site, with the time spent inside it.  A wait of tens of microseconds means the
device had nothing left to do (the host is the bound there); a long one means the host was ahead.  dev tool; GPU box."""
import collections
import os
import sys
import threading
import time
import traceback

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402
from cagroup3d_amd.optim import ClippedAdamW  # noqa: E402

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = ClippedAdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
model.split_late_parameters(opt)            # as bench.py does
batch = build_model.synthetic_batch("S50k", int(os.environ.get("BATCH", "4")), device=dev)
for _ in range(6):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()

acc = collections.OrderedDict()
main = threading.main_thread()
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if fr.filename.startswith(ROOT) and "tools/" not in fr.filename:
            return "%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.name)
    return "?"


def wrap(name):
    fn = getattr(torch.Tensor, name)

    def timed(self, *a, **k):
        if not self.is_cuda or threading.current_thread() is not main:
            return fn(self, *a, **k)
        t0 = time.perf_counter()
        r = fn(self, *a, **k)
        dt = time.perf_counter() - t0
        s = site()
        e = acc.setdefault(s, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += dt
        e[2] += t0 - STEP0[0]
        return r
    setattr(torch.Tensor, name, timed)


STEP0 = [0.0]
for n in ("tolist", "cpu", "item"):
    wrap(n)
N = 20
t_all = time.perf_counter()
for _ in range(N):
    STEP0[0] = time.perf_counter()
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
step = (time.perf_counter() - t_all) / N
print("step %.2f ms; blocking reads on the issuing thread: %d per step, %.2f ms per step inside them" %
      (1e3 * step, sum(e[0] for e in acc.values()) / N, 1e3 * sum(e[1] for e in acc.values()) / N))
print("  at ms   wait ms  site            (mean position of the calls in the step, total wait per step)")
for s, (n, t, at) in acc.items():
    print("  %6.2f  %7.3f  x%-2d %s" % (1e3 * at / n, 1e3 * t / N, n // N, s))
bench.finish_prefetch(model)
