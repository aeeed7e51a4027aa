"""cProfile of the issuing thread between the first blocking read of the step (class_rows) and the start of backward: the
stretch where the device waits for the host.  dev tool; GPU box."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402
from cagroup3d_amd.ops import head_stage  # noqa: E402
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
pr = cProfile.Profile()
orig_rows, orig_bwd = head_stage.class_rows, torch.Tensor.backward


def rows(*a, **k):
    r = orig_rows(*a, **k)
    pr.enable()
    return r


def bwd(self, *a, **k):
    pr.disable()
    return orig_bwd(self, *a, **k)


head_stage.class_rows = rows
torch.Tensor.backward = bwd
N = 10
for _ in range(N):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
bench.finish_prefetch(model)
st = pstats.Stats(pr)
print("per step: %.2f ms under the profiler" % (1e3 * st.total_tt / N))
st.sort_stats("cumulative").print_stats(70)
st.sort_stats("tottime").print_stats(40)
