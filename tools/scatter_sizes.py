"""Sizes and index multiplicities of the scatter-add backward launches of one training step (dev tool, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cagroup3d_amd import me, build_model
me.PRECISION = 1
dev = torch.device("cuda", 0)
model, cfg = bench.make_model("scannet", True, dev)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
batch = build_model.synthetic_batch("S50k", 4, device=dev)
for _ in range(3):
    bench.train_step(model, opt, batch, 10.0)
orig = me.GatherRowsFunction.backward
def patched(ctx, dout):
    (idx,) = ctx.saved_tensors
    cnt = torch.bincount(idx.long(), minlength=ctx.n_src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(ctx, dout)
    e1.record(); torch.cuda.synchronize()
    print("scatter_add_rows: n=%d c=%d n_src=%d rows hit=%d max multiplicity=%d mean=%.1f  %.1f us" % (
        idx.shape[0], dout.shape[1], ctx.n_src, int((cnt > 0).sum()), int(cnt.max()), idx.shape[0] / max(int((cnt > 0).sum()), 1), e0.elapsed_time(e1) * 1e3))
    return out
me.GatherRowsFunction.backward = staticmethod(patched)
bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
