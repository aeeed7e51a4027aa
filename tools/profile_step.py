"""Wall-clock breakdown of one training step (synchronised between stages; dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model
import bench


def main():
    forced = "--natural" not in sys.argv
    model, cfg = bench.make_model("scannet", forced, "cuda")
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
    batch = build_model.synthetic_batch("S50k", 4, device="cuda")
    for _ in range(2):
        bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()

    def tick(label, t0):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print("%-28s %8.1f ms" % (label, (t1 - t0) * 1e3))
        return t1
    from cagroup3d_amd import me
    me.KernelProfile.reset(); me.KernelProfile.enabled = True; me.KernelProfile.wgrad = True
    bench.train_step(model, opt, batch, 10)
    torch.cuda.synchronize()
    me.KernelProfile.enabled = False
    agg = {}
    for r in me.KernelProfile.records:
        k = r[4]
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += r[0].elapsed_time(r[1]); a[2] += r[2]
    print("conv launches by shape (kind, K, cin, cout, pairs, n_out, nseg): calls, ms, TF/s")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
        print("  %-50s %3d  %7.3f ms  %6.1f TF" % (str(k), a[0], a[1], a[2] / a[1] / 1e9))
    print("  total conv ms:", sum(a[1] for a in agg.values()))
    for rep in range(1):
        print("---- step", rep)
        b = bench.fresh(batch)
        opt.zero_grad(set_to_none=True)
        t = time.perf_counter(); t_start = t
        model.module_list[1].semantic_threshold = 0.15
        b["points"][:, -3:] = b["points"][:, -3:] / 255.
        b["sp_tensor"] = model.voxelization(b["points"]); t = tick("voxelization", t)
        b.update(model.module_list[0](b)); t = tick("backbone fwd", t)
        b.update(model.module_list[1](b)); t = tick("dense head fwd (+NMS)", t)
        b.update(model.module_list[2](b)); t = tick("roi head fwd", t)
        loss, tb, disp = model.get_training_loss(b); t = tick("losses", t)
        loss.backward(); t = tick("backward", t)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10); t = tick("clip", t)
        opt.step(); t = tick("adamw", t)
        print("%-28s %8.1f ms" % ("TOTAL", (t - t_start) * 1e3))
        print("rois per scene:", [len(p[0]) for p in b["pred_bbox_list"]], "class voxels:", sum(len(x) for x in b["one_stage_results"][0][3][0]) if False else "")


if __name__ == "__main__":
    main()
