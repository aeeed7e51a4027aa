"""Per-shape table of the timed sparse-conv launches of one bench step (dev tool): which (K, cin, cout, rows) shapes
the conv time sits in and how far each is from its own roofline bound.
usage (GPU box): python tools/conv_shapes.py [--wgrad]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cagroup3d_amd import me, build_model

me.PRECISION = 1

me.HEAD_PRECISION = me.heads_from_env()
dev = torch.device("cuda", 0)
DATASET, CFG, BATCH = os.environ.get("DATASET", "scannet"), os.environ.get("CFG", "S50k"), int(os.environ.get("BATCH", "4"))
print("# %s, %d x %s scenes per step, heads: %s, backbone rows: %s" % (DATASET, BATCH, CFG, os.environ.get("CG3D_HEADS", "split"),
                                                                   "bf16" if os.environ.get("CG3D_ACT_BF16", "1") != "0" else "fp32"))
model, cfg = bench.make_model(DATASET, True, dev, build_model.VOXEL_SIZE_OF_CONFIG.get(CFG))
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
batch = build_model.synthetic_batch(CFG, BATCH, device=dev)
for _ in range(4):
    bench.train_step(model, opt, batch, 10.0)
me.KernelProfile.reset()
me.KernelProfile.enabled = True
me.KernelProfile.wgrad = "--wgrad" in sys.argv
STEPS = 5
for _ in range(STEPS):
    bench.train_step(model, opt, batch, 10.0)
torch.cuda.synchronize()
me.KernelProfile.enabled = False
agg = collections.OrderedDict()
for ev0, ev1, flops, nbytes, meta, _pair_bytes in me.KernelProfile.records:       # nbytes: SURVEY 8(d) (every tensor once)
    d = agg.setdefault(meta[:4] + (meta[5],), [0, 0.0, 0.0, 0.0, 0])
    d[0] += 1; d[1] += ev0.elapsed_time(ev1); d[2] += flops; d[3] += nbytes; d[4] += meta[4]
print("%-14s %3s %5s %5s %8s %9s %5s %8s %8s %7s %7s %6s" % ("kind", "K", "cin", "cout", "rows", "pairs", "n/st", "ms/step", "avg us", "GB/s", "TF/s", "bound%"))
tot = collections.Counter()
for key, (n, ms, fl, by, pairs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kind, K, cin, cout, rows = key
    s = ms * 1e-3
    if kind.endswith("x3"):
        fl = 3.0 * fl                 # split operands: three bf16 products per fp32-accurate product (bench.py prices them the same way)
    bound = max(fl / (157.3e12 if kind in ("pairs", "wgrad") else 2.5e15), by / 8e12)
    tot[kind] += ms / STEPS
    print("%-14s %3d %5d %5d %8d %9d %5.1f %8.3f %8.1f %7.0f %7.1f %6.1f" % (
        kind, K, cin, cout, rows, pairs // n, n / STEPS, ms / STEPS, ms / n * 1e3, by / s / 1e9, fl / s / 1e12, 100 * bound / s))
print(dict(tot))
