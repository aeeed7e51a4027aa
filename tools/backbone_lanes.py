"""Device time of the backbone's launch program (forward table, backward table) on one queue and on its lanes.

    python tools/backbone_lanes.py [--config S50k] [--batch 4] [--iters 10]

The same compiled tables are issued with every lane on the one stream (CG3D_LANES_RUN=0) and with a stream per lane; then
again with the weight gradients on a lane of their own (engine.WGRAD_LANE = 2).  A pass is ONE foreign call, so the host is
not in the way: the numbers are what the device needs for the pass (HIP events on the main stream around the call; the table
ends with lane 0 waiting for the others)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from cagroup3d_amd import build_model, engine, me


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="S50k")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dataset", default="scannet")
    ap.add_argument("--wgrad-lanes", default="0,2,1", help="engine.WGRAD_LANE values to go through")
    ap.add_argument("--dappm-lanes", default="", help="engine.DAPPM_LANES, e.g. 3 or 3,2")
    ap.add_argument("--marks", action="store_true",
                    help="a k_clip_coef launch before and after each pass: cuts for tools/lane_timeline.py in a rocprofv3 kernel trace")
    args = ap.parse_args()
    dev = "cuda"
    engine.AUTOTUNE = False                 # (this tool chooses the queues itself)
    engine.DAPPM_LANES = [int(x) for x in args.dappm_lanes.split(",") if x]
    me.PRECISION, me.BF16_ROWS = 1, True
    model, _ = build_model.build_cagroup3d(args.dataset, seed=0)
    model = model.to(dev).train()
    net = model.backbone_3d
    batch = build_model.synthetic_batch(args.config, args.batch, device=dev)

    from cagroup3d_amd import _lib
    from ctypes import c_float, c_int64
    lib = _lib.get()
    scratch = torch.zeros(4, dtype=torch.float64, device=dev)
    nc = torch.zeros(2, dtype=torch.float32, device=dev)

    def mark():
        if args.marks:
            lib.call("cg3d_grad_norm_clip", None, None, c_int64(0), None, c_float(1.0), _lib.ptr(scratch), _lib.ptr(nc), _lib.ptr(nc[1:]), lib.stream())

    def step(timed):
        for p in net.parameters():
            p.grad = None
        me._ROWS16.clear(); me._ROWS48.clear(); me._STATS.clear()       # (what CAGroup3D.forward does at the start of a step:
        me.zero_arena().reset()                                         #  the tables hold views of the previous pass's arena)
        me.WANT_BN_STATS = True
        pts = batch["points"].clone()
        pts[:, -3:] = pts[:, -3:] / 255.
        sp = model.voxelization(pts)
        me.prepare_weights(True)
        try:
            comp = engine.compile_backbone(net, sp)          # (maps, plans, tables: outside the timed call)
        except engine.NotReady:
            comp = None                                      # first step: the weights enter the arena
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        mark()
        out = net({"sp_tensor": sp, "batch_size": batch["batch_size"], "engine_program": comp})["sp_tensor"]
        mark()
        e[1].record()
        me.finish_weights()
        up = torch.ones_like(out.F)
        loss = (out.F * up).sum()
        torch.cuda.synchronize()
        e[2].record()
        mark()
        loss.backward()
        mark()
        e[3].record()
        torch.cuda.synchronize()
        return e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])

    for wl in [int(x) for x in args.wgrad_lanes.split(",")]:
        engine.WGRAD_LANE = wl
        for lanes_run in (False, True, False, True):
            engine.LANES_RUN = lanes_run
            for _ in range(3):
                step(False)
            p0 = engine.STATS["program_passes"]
            t = [step(True) for _ in range(args.iters)]
            assert engine.STATS["program_passes"] == p0 + args.iters, "the program path did not run"
            f = sorted(x[0] for x in t)[len(t) // 2]
            b = sorted(x[1] for x in t)[len(t) // 2]
            print("DAPPM lanes %s, weight gradients on lane %s | %s: forward %.3f ms, backward %.3f ms, sum %.3f ms (median of %d)" % (
                engine.DAPPM_LANES or "-", wl if wl else "of their layer", "a stream per lane" if lanes_run else "one stream      ", f, b, f + b, args.iters), flush=True)


if __name__ == "__main__":
    main()
