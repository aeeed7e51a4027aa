"""Host wall-clock per phase of the training step WITHOUT extra synchronisation (dev tool): where the host spends its
time, including the time it sits in the step's own host reads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench

me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
params = [p for p in model.parameters()]
use_prefetch = os.environ.get("CG3D_PREFETCH", "1") != "0"
acc = {}
prepared = None


def tick(name, t0):
    t1 = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t1 - t0) * 1e3
    return t1


N = 12
for it in range(N + 3):
    if it == 3:
        acc.clear()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
    t = time.perf_counter()
    b = bench.fresh(batch)
    opt.zero_grad(set_to_none=True)
    model.module_list[1].semantic_threshold = 0.15
    b["points"][:, -3:] = b["points"][:, -3:] / 255.
    b["sp_tensor"] = model.voxelization(b["points"], prepared)
    prepared = None
    t = tick("voxelize", t)
    b.update(model.module_list[0](b)); t = tick("backbone", t)
    b.update(model.module_list[1](b)); t = tick("head", t)
    b.update(model.module_list[2](b)); t = tick("roi", t)
    loss, tb, disp = model.get_training_loss(b); t = tick("loss", t)
    loss.backward(); t = tick("backward", t)
    torch.nn.utils.clip_grad_norm_(params, 10); opt.step(); t = tick("clip+opt", t)
    if use_prefetch:
        prepared = model.prefetch_coordinates(batch); t = tick("prefetch", t)
torch.cuda.synchronize()
total = (time.perf_counter() - t_start) * 1e3 / N
print("prefetch=%s  step %.1f ms;  host ms/step: " % (use_prefetch, total) + "  ".join("%s %.1f" % (k, v / N) for k, v in acc.items()))
