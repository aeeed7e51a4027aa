"""Does a host read on a side stream wait for a busy main stream?  (dev tool)"""
import time, torch
a = torch.randn(8192, 8192, device="cuda")
import os
side = torch.cuda.Stream(priority=int(os.environ.get("PRIO", "0")))
x = torch.arange(10, device="cuda")
torch.cuda.synchronize()


def busy():
    for _ in range(40):
        torch.mm(a, a)


for name, fn in (("item", lambda t: t.sum().item()), ("cpu", lambda t: t.cpu()),
                 ("pinned+stream sync", None), ("nonzero", lambda t: torch.nonzero(t))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    busy()
    t1 = time.perf_counter()
    with torch.cuda.stream(side):
        y = x + 1
        if fn is None:
            buf = torch.empty(y.shape, dtype=y.dtype, pin_memory=True)
            buf.copy_(y, non_blocking=True)
            side.synchronize()
        else:
            fn(y)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("%-20s enqueue main %.1f ms | side host read took %.2f ms | main finished %.1f ms later" % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
