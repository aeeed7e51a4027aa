"""gpu: bench.py in its multi-process form -- torch.distributed.run, two ranks, every rank on cuda:0 over gloo (RCCL
refuses two ranks on one device; the driver's 2/4/8-GPU runs are the only place the RCCL build of this path executes).
Exercises init, the parameter broadcast, the in-step normaliser all-reduce, the gradient exchange from inside backward,
the barrier-bracketed timing and the max-over-ranks reduction end to end."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_one_device():
    env = dict(os.environ, CG3D_SINGLE_DEVICE="1", CG3D_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--config", "S5k"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["steps"] == 2 and b["scaling"] == "weak" and b["value"] > 0
    assert b["config"]["parallelism"] == "dp2" and "cpu_baseline" not in b      # the CPU leg runs at N=1 only
    assert abs(b["value"] - 2 * 4 * 2 / (b["ms_per_step"] * 2e-3)) < 1e-6 * b["value"]   # whole-job scenes / time
