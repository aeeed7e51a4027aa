"""Long-run drift vs GPU clocks / power sampled from sysfs during the run (dev tool, GPU only)."""
import os, sys, time, glob, threading
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
if os.environ.get("SLEEP"):
    time.sleep(float(os.environ["SLEEP"]))
sclk = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
pw = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input")
samples, stop = [], False
def poll():
    while not stop:
        try:
            cur = [l for l in open(sclk[0]).read().splitlines() if "*" in l]
            p = int(open(pw[0]).read()) / 1e6 if pw else -1
            samples.append((time.perf_counter(), cur[0] if cur else "?", p))
        except Exception as e:
            samples.append((time.perf_counter(), repr(e), -1))
        time.sleep(0.25)
th = threading.Thread(target=poll, daemon=True); th.start()
t0 = time.perf_counter()
marks = []
for i in range(100):
    bench.train_step(model, opt, batch, 10)
    marks.append(time.perf_counter())
torch.cuda.synchronize()
stop = True
print("sysfs:", sclk[:1], pw[:1])
for k in range(0, 100, 10):
    a, b = (marks[k - 1] if k else t0), marks[k + 9]
    ss = [s for s in samples if a <= s[0] <= b]
    print("steps %3d-%3d: %.1f ms/step (host clock) | sclk %s | power %s W" % (k, k + 9, (b - a) * 100, sorted(set(s[1] for s in ss)), ["%.0f" % s[2] for s in ss][:4]))
