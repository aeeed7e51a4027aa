"""aten-op level CPU/GPU profile of one training step per phase (dev tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
from cagroup3d_amd import build_model, me
import bench

me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(3):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=False) as prof:
    b = bench.fresh(batch)
    opt.zero_grad(set_to_none=True)
    model.module_list[1].semantic_threshold = 0.15
    b["points"][:, -3:] = b["points"][:, -3:] / 255.
    with record_function("PH_voxelize"):
        b["sp_tensor"] = model.voxelization(b["points"])
    with record_function("PH_backbone"):
        b.update(model.module_list[0](b))
    with record_function("PH_head"):
        b.update(model.module_list[1](b))
    with record_function("PH_roi"):
        b.update(model.module_list[2](b))
    with record_function("PH_loss"):
        loss, tb, disp = model.get_training_loss(b)
    with record_function("PH_backward"):
        loss.backward()
    with record_function("PH_clip"):
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
    with record_function("PH_opt"):
        opt.step()
    torch.cuda.synchronize()
ka = prof.key_averages()
print("%-40s %8s %10s" % ("name", "count", "cpu_ms"))
for e in sorted(ka, key=lambda e: -e.self_cpu_time_total)[:45]:
    print("%-40s %8d %10.2f" % (e.key[:40], e.count, e.self_cpu_time_total / 1e3))
print("PHASES")
for e in ka:
    if e.key.startswith("PH_"):
        print("%-20s cpu_total %8.2f ms" % (e.key, e.cpu_time_total / 1e3))
