"""cProfile of ALL the Python a training step runs, on one thread: backward on the calling thread
(torch.autograd.set_multithreading_enabled(False)) and the coordinate dry run inline (CG3D_PREFETCH_THREAD=0), batch 1 (the
GPU never blocks the host).  Sorted by own time: where the interpreter spends the host's ~28 ms.  dev tool; GPU box."""
import cProfile
import os
import pstats
import sys

os.environ["CG3D_PREFETCH_THREAD"] = "0"
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
opt = ClippedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
batch = build_model.synthetic_batch("S50k", 1, device=dev)
torch.autograd.set_multithreading_enabled(False)
for _ in range(5):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
N = 10
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    bench.train_step(model, opt, batch, 10.0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", "70")))
