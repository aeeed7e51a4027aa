"""GPU time and launches per PHASE of the training step (dev tool, GPU box):

    rocprofv3 --kernel-trace --output-format csv -d /tmp/ph -o ph -- python tools/phase_kernels.py run
    python tools/phase_kernels.py report /tmp/ph/.../ph_kernel_trace.csv

`run` executes a few steps with a distinctive marker launch (a fill of 7001 + i elements) between the phases of the LAST
step (and a device sync, so the worker thread's kernels do not leak across); `report` cuts the kernel trace at the markers."""
import csv
import os
import re
import sys

PHASES = ["voxelise", "backbone fwd", "head fwd", "roi head fwd", "losses", "backward", "optimizer"]

if sys.argv[1] == "run":
    os.environ.setdefault("CG3D_PREFETCH_THREAD", "0")
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    import torch
    import bench
    from cagroup3d_amd import build_model, me
    me.PRECISION = 1
    me.HEAD_PRECISION = me.heads_from_env()
    dev = torch.device("cuda", 0)
    model, cfg = bench.make_model("scannet", True, dev)
    model.train()
    from cagroup3d_amd.optim import ClippedAdamW
    opt = ClippedAdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    batch = build_model.synthetic_batch("S50k", 4, device=dev)
    for _ in range(4):
        bench.train_step(model, opt, batch, 10.0)
    torch.cuda.synchronize()

    def mark(i):
        torch.cuda.synchronize()
        torch.empty(64, device=dev).normal_()          # a kernel nothing else in the step launches
        torch.cuda.synchronize()
    # one step, phase by phase (the detector's own forward, unrolled)
    b = bench.fresh(batch)
    b["cur_epoch"] = 0
    opt.zero_grad(set_to_none=True)
    me._ROWS16.clear(); me._STATS.clear(); me.prepare_weights(True)
    mark(0)
    b["points"][:, -3:] = b["points"][:, -3:] / 255.
    b["sp_tensor"] = model.voxelization(b["points"], None)
    mark(1)
    b.update(model.module_list[0](b)); mark(2)
    b.update(model.module_list[1](b)); mark(3)
    b.update(model.module_list[2](b)); mark(4)
    loss, tb, _ = model.get_training_loss(b); mark(5)
    me.finish_weights()
    loss.backward(); mark(6)
    opt.clip_and_step(10.0); mark(7)
    print("loss", float(loss))
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "distribution" in r["Kernel_Name"] or "normal" in r["Kernel_Name"].lower()]
    marks = marks[-8:]
    assert len(marks) == 8, len(marks)
    print("%-14s %9s %9s   top kernels" % ("phase", "ms", "launches"))
    for p in range(7):
        seg = rows[marks[p] + 1:marks[p + 1]]
        tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg) / 1e6
        by = {}
        for r in seg:
            k = re.sub(r"\(.*", "", r["Kernel_Name"])
            k = re.sub(r"^void ", "", k)[:60]
            d = by.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        aten = sum(v[0] for k, v in by.items() if k.startswith("at::native") or "rocclr" in k or "rocprim" in k)
        aten_ms = sum(v[1] for k, v in by.items() if k.startswith("at::native") or "rocclr" in k or "rocprim" in k)
        print("%-14s %9.2f %9d   (torch / rocclr / rocprim: %d launches, %.2f ms)" % (PHASES[p], tot, len(seg), aten, aten_ms))
        for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:6]:
            print("      %7.3f ms %4d  %s" % (v[1], v[0], k))
