"""aten-op level CPU profile of one training step, per phase (dev tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cagroup3d_amd import build_model, me
import bench

me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(3):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()


def phase(name, fn):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        out = fn()
        torch.cuda.synchronize()
    nk = sum(1 for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA)
    ka = [e for e in prof.key_averages() if e.device_type == torch.autograd.DeviceType.CPU]
    tot = sum(e.self_cpu_time_total for e in ka) / 1e3
    n = sum(e.count for e in ka)
    print("== %-10s host %7.2f ms, %5d ops, %5d device activities (kernels + copies)" % (name, tot, n, nk))
    for e in sorted(ka, key=lambda e: -e.self_cpu_time_total)[:14]:
        print("     %-42s %6d %8.2f ms" % (e.key[:42], e.count, e.self_cpu_time_total / 1e3))
    return out


b = bench.fresh(batch)
opt.zero_grad(set_to_none=True)
model.module_list[1].semantic_threshold = 0.15
b["points"][:, -3:] = b["points"][:, -3:] / 255.
b["sp_tensor"] = model.voxelization(b["points"])
phase("backbone", lambda: b.update(model.module_list[0](b)))
phase("head", lambda: b.update(model.module_list[1](b)))
phase("roi", lambda: b.update(model.module_list[2](b)))
loss, tb, disp = phase("loss", lambda: model.get_training_loss(b))
phase("backward", lambda: loss.backward())
phase("clip+opt", lambda: (torch.nn.utils.clip_grad_norm_(model.parameters(), 10), opt.step()))
torch.cuda.synchronize()
