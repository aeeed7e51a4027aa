"""Host-side wall time of every line-level call in _loss_batched / _face_distances (dev tool): finds host stalls that
leave the GPU idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
from cagroup3d_amd.pcdet.models.dense_heads.target_assigner import cagroup3d_assigner as A
import bench

me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(4):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()

log = []
orig_stack, orig_cat = torch.stack, torch.cat


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        dt = (time.perf_counter() - t0) * 1e3
        if dt > 0.3:
            shp = tuple(r.shape) if torch.is_tensor(r) else None
            log.append((dt, name, shp))
        return r
    return w


torch.stack = timed("stack", orig_stack)
torch.cat = timed("cat", orig_cat)
import gc
gc_t = []
def cb(phase, info):
    if phase == "start":
        cb.t0 = time.perf_counter()
    else:
        gc_t.append(((time.perf_counter() - cb.t0) * 1e3, info.get("generation")))
gc.callbacks.append(cb)
for i in range(3):
    log.clear(); gc_t.clear()
    t0 = time.perf_counter()
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
    print("step %.1f ms; slow stack/cat calls:" % ((time.perf_counter() - t0) * 1e3), [(round(d, 2), n, s) for d, n, s in log],
          " gc:", [(round(d, 2), g) for d, g in gc_t if d > 0.2])
