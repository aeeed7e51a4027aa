"""Host side of the bench step: is the step bound by the Python / launch path or by the GPU?  Prints (a) the time the
host needs to ISSUE a step (no sync inside), (b) the synchronised step time, (c) a cProfile of the issue path
(dev tool, GPU box)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
from cagroup3d_amd import build_model, me  # noqa: E402

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
batch = build_model.synthetic_batch("S50k", int(os.environ.get("CG3D_PROFILE_BATCH", "4")), device=dev)
for _ in range(5):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    bench.train_step(model, opt, batch, 10.0)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3 * (t1 - t0) / N:.2f} ms/step, with final sync {1e3 * (t2 - t0) / N:.2f} ms/step (GPU backlog at the end {1e3 * (t2 - t1):.1f} ms)")
# each step synchronised: host + GPU serialised only where the step itself syncs
ts = []
for _ in range(N):
    torch.cuda.synchronize()
    a = time.perf_counter()
    bench.train_step(model, opt, batch, 10.0)
    b = time.perf_counter()
    torch.cuda.synchronize()
    c = time.perf_counter()
    ts.append((b - a, c - a))
print("per step from idle: issue %.2f ms, done %.2f ms" % (1e3 * sum(t[0] for t in ts) / N, 1e3 * sum(t[1] for t in ts) / N))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    bench.train_step(model, opt, batch, 10.0)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
