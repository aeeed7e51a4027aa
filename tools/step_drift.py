"""Does a long unsynchronised run drift?  Memory / allocator / gc statistics every 10 steps (dev tool, GPU only)."""
import os, sys, gc, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench
me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(5):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
if os.environ.get("PREALLOC"):
    x = torch.empty(int(os.environ["PREALLOC"]) << 30, dtype=torch.uint8, device="cuda"); del x
if os.environ.get("CYCLES") == "1":
    import collections
    gc.collect()
    gc.set_debug(gc.DEBUG_SAVEALL)
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
    n = gc.collect()
    cnt = collections.Counter(type(o).__name__ for o in gc.garbage)
    print("step garbage: %d unreachable objects; top types: %s" % (n, cnt.most_common(25)))
    tens = [o for o in gc.garbage if torch.is_tensor(o)]
    print("step garbage tensors: %d, %.2f GB" % (len(tens), sum(t.numel() * t.element_size() for t in tens if t.is_cuda) / 2**30))
    import types
    fr = [o for o in gc.garbage if isinstance(o, (types.FunctionType, types.MethodType))]
    print("step garbage functions:", collections.Counter(getattr(f, "__qualname__", str(f)) for f in fr).most_common(20))
    cls = collections.Counter(type(o).__module__ + "." + type(o).__qualname__ for o in gc.garbage if not isinstance(o, (dict, list, tuple, types.FunctionType, types.CellType)))
    print("step garbage classes:", cls.most_common(30))
    sys.exit(0)
if os.environ.get("GC_OFF") == "1":
    gc.collect(); gc.freeze(); gc.disable()
evs = []
t0 = time.perf_counter()
for i in range(120):
    h0 = time.perf_counter()
    bench.train_step(model, opt, batch, 10)
    h1 = time.perf_counter()
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append((e, (h1 - h0) * 1e3))
    if i % 10 == 9:
        st = torch.cuda.memory_stats()
        print("step %3d host-elapsed %.0f ms  reserved %.2f GB allocated %.2f GB  device mallocs %d  retries %d  gc %s" % (
            i, (time.perf_counter() - t0) * 1e3, st["reserved_bytes.all.current"] / 2**30, st["allocated_bytes.all.current"] / 2**30,
            st["segment.all.allocated"], st["num_alloc_retries"], gc.get_count()), flush=True)
torch.cuda.synchronize()
d = [evs[i][0].elapsed_time(evs[i + 1][0]) for i in range(119)]
for k in range(0, 119, 10):
    print("steps %3d-%3d: gpu interval mean %.1f ms | host time per step mean %.1f ms" % (k, k + 9, sum(d[k:k + 10]) / len(d[k:k + 10]),
          sum(h for _, h in evs[k:k + 10]) / len(evs[k:k + 10])))
