"""The four BatchNorm launches (statistics, apply, backward reduce, backward apply) per backbone shape: time and the
HBM rate over the bytes each MUST move (dev tool, GPU box)."""
import os
import sys
from ctypes import c_float, c_int32, c_int64

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cagroup3d_amd import _lib, me  # noqa: E402
from cagroup3d_amd._lib import ptr  # noqa: E402


def timeit(fn, n=30, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    lib = _lib.get()
    print("%8s %5s | %22s | %22s | %22s | %22s" % ("rows", "C", "stats us (GB/s)", "apply us (GB/s)", "bwd reduce us (GB/s)", "bwd apply us (GB/s)"))
    for rows, C in ((155773, 64), (82107, 128), (23015, 256), (5330, 512), (1229, 1024), (28000, 128)):
        for res, act in ((False, 1), (True, 1)):
            x = torch.randn(rows, C, device="cuda")
            dy = torch.randn(rows, C, device="cuda")
            r = torch.randn(rows, C, device="cuda") if res else None
            gamma, beta = torch.ones(1, C, device="cuda"), torch.zeros(1, C, device="cuda")
            chunks, nchunk, gco, group_n, achunks, nachunk, _ = me._bn_chunks((0, rows), x.device, C)
            ws = torch.empty(max(nchunk, 1) * 2 * C, device="cuda")
            mean, var = torch.empty(1, C, device="cuda"), torch.empty(1, C, device="cuda")
            y, y16 = torch.empty_like(x), torch.empty(x.shape, dtype=torch.int16, device="cuda")
            dx, dx16 = torch.empty_like(x), torch.empty(x.shape, dtype=torch.int16, device="cuda")
            dres = torch.empty_like(x) if res else None
            dbeta, dgamma = torch.empty(1, C, device="cuda"), torch.empty(1, C, device="cuda")
            S = lib.stream
            f_stats = lambda: lib.call("cg3d_bn_sums", ptr(x), ptr(chunks), c_int64(nchunk), c_int32(1), c_int32(C), ptr(ws), S())
            f_apply = lambda: lib.call("cg3d_bn_apply", ptr(x), ptr(r), ptr(achunks), c_int64(nachunk), c_int32(C), ptr(mean), ptr(var),
                                       c_float(1e-5), ptr(gamma), ptr(beta), c_int32(act), ptr(y), ptr(y16), S())
            f_red = lambda: lib.call("cg3d_bn_bwd_sums", ptr(dy), ptr(x), ptr(y), ptr(chunks), c_int64(nchunk), c_int32(1), c_int32(C),
                                     ptr(mean), ptr(var), c_float(1e-5), c_int32(act), ptr(ws), S())
            f_bapp = lambda: lib.call("cg3d_bn_bwd_apply", ptr(dy), ptr(x), ptr(y), ptr(achunks), c_int64(nachunk), c_int32(C), ptr(mean),
                                      ptr(var), c_float(1e-5), ptr(gamma), ptr(dbeta), ptr(dgamma), ptr(group_n), c_int32(act),
                                      c_int32(1), ptr(dx), ptr(dx16), ptr(dres), S())
            t = [timeit(f) for f in (f_stats, f_apply, f_red, f_bapp)]
            e = rows * C * 4.0
            by = [e, e * (2.5 + (1 if res else 0)), e * 3, e * (4.5 + (1 if res else 0))]
            print("%8d %5d%s | " % (rows, C, "+r" if res else "  ") + " | ".join("%8.1f (%6.0f)" % (tt, b / tt / 1e3) for tt, b in zip(t, by)), flush=True)


if __name__ == "__main__":
    main()
