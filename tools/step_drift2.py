"""Long-run drift: is it the GPU (kernels slower) or the host (gaps)?  Conv kernel time per step from HIP events + a
host-only microbenchmark per step block (dev tool, GPU only)."""
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
def host_probe():
    t0 = time.perf_counter()
    s = 0
    for i in range(200000):
        s += i * i
    return (time.perf_counter() - t0) * 1e3
me.KernelProfile.reset(); me.KernelProfile.enabled = True
marks, evs, probes = [], [], []
for i in range(100):
    bench.train_step(model, opt, batch, 10)
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    marks.append(len(me.KernelProfile.records))
    if i % 10 == 9:
        probes.append(host_probe())
torch.cuda.synchronize()
recs = me.KernelProfile.records
prev = 0
per = []
for m in marks:
    per.append(sum(r[0].elapsed_time(r[1]) for r in recs[prev:m])); prev = m
d = [evs[i].elapsed_time(evs[i + 1]) for i in range(99)]
for k in range(0, 100, 10):
    print("steps %3d-%3d: step interval %.1f ms | conv fwd/dgrad kernel time %.2f ms/step | host probe %.1f ms" % (
        k, k + 9, sum(d[k:k + 10]) / len(d[k:k + 10]), sum(per[k:k + 10]) / 10, probes[k // 10]))
try:
    import subprocess
    print(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout[-1500:])
except Exception as ex:
    print("rocm-smi failed", ex)
