"""Which ops do the device fill kernels of a step come from?  torch profiler, fills grouped by the enclosing
aten op / autograd node (dev tool, GPU only)."""
import os, sys, collections
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
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
names = tuple(os.environ.get("OPS", "aten::fill_,aten::zero_").split(","))
ev = [e for e in prof.events() if e.name in names]
cnt = collections.Counter()
for e in ev:
    p = e.cpu_parent
    chain = []
    while p is not None and len(chain) < 3:
        chain.append(p.name)
        p = p.cpu_parent
    st = [s for s in (e.stack or []) if "cagroup3d_amd" in s or "bench.py" in s]
    cnt[(" <- ".join(chain), st[0].split("repo/")[-1] if st else "")] += 1
print(names, "ops in one step:", len(ev))
for k, v in cnt.most_common(30):
    print("%4d  %-70s %s" % (v, k[0][:70], k[1][:90]))
