"""CPU time of the two host threads of a training step: the issuing thread (train_step) and the worker thread (the next batch's
coordinate dry run + program compile), against the step's wall time.  Both hold the interpreter lock while they run Python, so
(main + worker) CPU time close to the wall time means the step is bound by the interpreter, not by the device.  dev tool; GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, engine, me  # noqa: E402
from cagroup3d_amd.optim import ClippedAdamW  # noqa: E402

me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = ClippedAdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
batch = build_model.synthetic_batch("S50k", int(os.environ.get("BATCH", "4")), device=dev)
jobs = []
orig = model.prefetch_coordinates


def timed(b):
    c0, w0 = time.thread_time(), time.perf_counter()
    r = orig(b)
    jobs.append((time.thread_time() - c0, time.perf_counter() - w0))
    return r


model.prefetch_coordinates = timed
comp = []
cb = engine.compile_backbone


def timed_compile(*a, **k):
    c0 = time.thread_time()
    r = cb(*a, **k)
    comp.append(time.thread_time() - c0)
    return r


engine.compile_backbone = timed_compile
for _ in range(8):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
del jobs[:], comp[:]
N = 30
cpu, wall = [], []
for _ in range(N):
    c0, w0 = time.thread_time(), time.perf_counter()
    bench.train_step(model, opt, batch, 10.0)
    cpu.append(time.thread_time() - c0)
    wall.append(time.perf_counter() - w0)
torch.cuda.synchronize()
bench.finish_prefetch(model)
ms = lambda x: 1e3 * float(np.median(x))
print("lanes %s | step wall %.2f ms | issuing thread CPU %.2f ms | worker thread: job wall %.2f ms, CPU %.2f ms (of it program compile %.2f ms) | CPU sum %.2f ms" % (
    engine.LANES, ms(wall), ms(cpu), ms([j[1] for j in jobs]), ms([j[0] for j in jobs]), ms(comp), ms(cpu) + ms([j[0] for j in jobs])))
