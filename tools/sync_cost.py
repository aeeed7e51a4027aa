import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dist.init_process_group("nccl", rank=0, world_size=1)
n = 200_000_000
flat = torch.zeros(n, device="cuda")
grads = [torch.randn(s, device="cuda") for s in [27 * 512 * 512] * 16 + [729 * 64 * 64] * 18 + [64 * 64 * 27] * 200]
views, off = [], 0
for g in grads:
    views.append(flat[off:off + g.numel()]); off += g.numel()
def t(fn, name, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); print("%-40s %.3f ms" % (name, (time.perf_counter() - t0) / it * 1e3))
t(lambda: torch._foreach_copy_(views, grads), "foreach_copy %d tensors %.0f MB" % (len(grads), off * 4 / 1e6))
t(lambda: dist.all_reduce(flat, op=dist.ReduceOp.AVG), "all_reduce AVG 506 MB (1 rank)")
t(lambda: dist.all_reduce(flat[:off]), "all_reduce SUM %.0f MB (1 rank)" % (off * 4 / 1e6))
t(lambda: dist.all_reduce(flat[:1000]), "all_reduce 4 KB (1 rank)")
