"""Where does the GPU sit idle inside one training step?  Kernel timeline -> gaps -> the CPU-side op that was
running while the GPU waited (dev tool, GPU only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity, record_function
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


def step():
    b = bench.fresh(batch)
    opt.zero_grad(set_to_none=True)
    model.module_list[1].semantic_threshold = 0.15
    b["points"][:, -3:] = b["points"][:, -3:] / 255.
    with record_function("PH_backbone"):
        b["sp_tensor"] = model.voxelization(b["points"])
        b.update(model.module_list[0](b))
    with record_function("PH_head"):
        b.update(model.module_list[1](b))
    with record_function("PH_roi"):
        b.update(model.module_list[2](b))
    with record_function("PH_loss"):
        loss, tb, disp = model.get_training_loss(b)
    with record_function("PH_backward"):
        loss.backward()
    with record_function("PH_opt"):
        torch.nn.utils.clip_grad_norm_(bench._PARAMS.get(id(model)) or list(model.parameters()), 10)
        opt.step()


step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    step()
    torch.cuda.synchronize()
evs = prof.events()
kern = sorted([(e.time_range.start, e.time_range.end, e.name) for e in evs
               if e.device_type == torch.autograd.DeviceType.CUDA], key=lambda t: t[0])
phases = sorted([(e.time_range.start, e.time_range.end, e.name) for e in evs
                 if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith("PH_")], key=lambda t: t[0])
# second step only
t0 = [p for p in phases if p[2] == "PH_backbone"][1][0]
kern = [k for k in kern if k[0] >= t0]
phases = [p for p in phases if p[0] >= t0]
busy = sum(k[1] - k[0] for k in kern)
span = kern[-1][1] - kern[0][0]
print("step span %.1f ms, GPU busy %.1f ms, idle %.1f ms, %d kernels" % (span / 1e3, busy / 1e3, (span - busy) / 1e3, len(kern)))


def phase_of(t):
    # the CPU phase running at host time t (GPU kernels lag the host, so this is 'what the host was doing')
    for s, e, n in phases:
        if s <= t <= e:
            return n
    return "between"


idle = {}
cnt = {}
end = kern[0][1]
for s, e, n in kern[1:]:
    if s > end:
        ph = phase_of(s)
        idle[ph] = idle.get(ph, 0) + (s - end)
        cnt[ph] = cnt.get(ph, 0) + 1
    end = max(end, e)
for ph in idle:
    print("  GPU idle while host in %-12s %7.2f ms over %4d gaps" % (ph, idle[ph] / 1e3, cnt[ph]))
for s, e, n in phases:
    kb = sum(min(k[1], e) - max(k[0], s) for k in kern if k[1] > s and k[0] < e)
    print("  %-12s host %7.2f ms   GPU busy inside that window %7.2f ms" % (n, (e - s) / 1e3, kb / 1e3))
