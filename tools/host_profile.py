"""cProfile of the host side of the training step (dev tool, GPU only): where the launch-bound time goes."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench

me.PRECISION = 1
model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(3):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
st.print_callers("item")
st.print_callers("'cpu'")
