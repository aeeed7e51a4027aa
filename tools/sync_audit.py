"""List the host<->device synchronisation points of one training step (dev tool, GPU only)."""
import collections, os, sys, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import build_model, me
import bench

model, cfg = bench.make_model("scannet", True, "cuda")
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)
batch = build_model.synthetic_batch("S50k", 4, device="cuda")
for _ in range(2):
    bench.train_step(model, opt, batch, 10)
torch.cuda.synchronize()
counts = collections.Counter()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message):
        return
    for fr in reversed(traceback.extract_stack()):
        if fr.filename.startswith(ROOT) and "sync_audit" not in fr.filename:
            counts["%s:%d %s" % (os.path.relpath(fr.filename, ROOT), fr.lineno, fr.line.strip()[:90])] += 1
            return
    counts["<other> %s" % str(message)[:80]] += 1

warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
bench.train_step(model, opt, batch, 10)
torch.cuda.set_sync_debug_mode(0)
print("sync points in one step: %d" % sum(counts.values()))
for k, v in counts.most_common(60):
    print("%4d  %s" % (v, k))
