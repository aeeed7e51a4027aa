"""Thread scaling of the cpu_baseline leg (the CPU oracle's full S50k training step) on the GPU box's host cores:
justifies the thread count bench.py uses for `cpu_baseline` (profiles/rNN_cpu_thread_scaling.txt).

    python tools/cpu_thread_scaling.py [8 16 32 64 128 ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
threads = [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]
print("host: %d logical CPUs; sample: 1 scene of S50k per step, 2 warm-up + median of up to 5 timed steps (budget 60 s per point)" % os.cpu_count())
print("%8s %12s %10s" % ("threads", "scenes/s", "s/step"))
for t in threads:
    if t > (os.cpu_count() or 1):
        continue
    env = dict(os.environ, OMP_NUM_THREADS=str(t), CG3D_CPU_BUDGET_S="60")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-sample", "S50k:1"], env=env,
                         capture_output=True, text=True, timeout=1800)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not lines:
        print("%8d failed: %s" % (t, out.stderr[-200:]))
        continue
    r = json.loads(lines[-1])
    print("%8d %12.4f %10.2f   (%s)" % (t, r["value"], 1.0 / r["value"], r["sample"].split(";")[1].strip()), flush=True)
