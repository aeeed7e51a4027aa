"""Timing of cg3d_scatter_add_rows / cg3d_interp_bwd under different index multiplicities (dev tool, GPU only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ctypes import c_int32, c_int64
from cagroup3d_amd import _lib
from microbench_conv import timeit
lib = _lib.get()
ptr = _lib.ptr
for n, c, nu, name in ((175616, 128, 175616, "identity"), (175616, 128, 60000, "random, 3x multiplicity"),
                       (175616, 128, 10000, "random, 17x multiplicity"), (400000, 64, 100000, "head-like, 4x"),
                       (400000, 64, 400000, "head-like, permutation")):
    dout = torch.randn(n, c, device="cuda")
    if nu == n and name == "identity":
        idx = torch.arange(n, device="cuda", dtype=torch.int32)
    elif nu == n:
        idx = torch.randperm(n, device="cuda").int()
    else:
        idx = torch.randint(0, nu, (n,), device="cuda", dtype=torch.int32)
    df = torch.zeros(nu, c, device="cuda")
    t = timeit(lambda: lib.call("cg3d_scatter_add_rows", ptr(dout), ptr(idx), ptr(df), c_int64(n), c_int32(c), lib.stream()), 10, 2)
    sidx, order = torch.sort(idx.long())
    t2 = timeit(lambda: torch.sort(idx.long()), 5, 1)
    print("scatter_add_rows n=%d c=%d unique<=%d (%s): %.1f us  -> %.0f GB/s payload   [torch.sort of idx: %.1f us]" % (
        n, c, nu, name, t * 1e3, n * c * 4 / t / 1e6, t2 * 1e3))
