"""cProfile of the coordinate dry run of a batch (CAGroup3D.prefetch_coordinates: every map, plan and table of the next step +
the backbone's launch program) run INLINE on an idle device: what the worker thread costs the interpreter per step.
dev tool; GPU box."""
import cProfile
import os
import pstats
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
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = ClippedAdamW(model.parameters(), lr=cfg.OPTIMIZATION.LR, weight_decay=cfg.OPTIMIZATION.WEIGHT_DECAY)
batch = build_model.synthetic_batch("S50k", int(os.environ.get("BATCH", "4")), device=dev)
for _ in range(6):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
bench.finish_prefetch(model)
me.prepare_weights(True)
N = 10
for _ in range(3):
    model.prefetch_coordinates(bench.fresh(batch))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(N):
    model.prefetch_coordinates(bench.fresh(batch))
    torch.cuda.synchronize()
print("dry run inline on an idle device: %.2f ms per batch (wall)" % (1e3 * (time.perf_counter() - t) / N))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    model.prefetch_coordinates(bench.fresh(batch))
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
print("per batch: %.2f ms under the profiler" % (1e3 * st.total_tt / N))
st.sort_stats("cumulative").print_stats(45)
st.sort_stats("tottime").print_stats(35)
