"""The BatchNorm launches of a training step as the model issues them (statistics from the slot table: cg3d_bn_apply_sums,
cg3d_bn_bwd_sums, cg3d_bn_bwd_apply_sums) per backbone shape, with fp32 and with bf16 row storage (CG3D_BN_STORE_BF16):
time per launch and the rate over the bytes each must move (dev tool, GPU box)."""
import os
import sys
from ctypes import c_float, c_int32, c_int64

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cagroup3d_amd import _lib, me  # noqa: E402
from cagroup3d_amd._lib import ptr  # noqa: E402
from mb_bn import timeit  # noqa: E402


def main():
    lib = _lib.get()
    S = lib.stream
    print("%8s %5s %5s | %20s | %20s | %20s" % ("rows", "C", "store", "apply us (GB/s)", "bwd sums us (GB/s)", "bwd apply us (GB/s)"))
    for rows, C in ((155773, 64), (82107, 128), (23015, 256), (5330, 512), (1229, 1024)):
        for res in (False, True):
            for s16 in (False, True):
                dt = torch.int16 if s16 else torch.float32
                w = 2.0 if s16 else 4.0
                mk = lambda: (torch.randn(rows, C, device="cuda").to(torch.bfloat16).view(torch.int16) if s16 else torch.randn(rows, C, device="cuda"))
                x, dy, r = mk(), mk(), (mk() if res else None)
                xf = torch.randn(rows, C, device="cuda")
                gamma, beta = torch.ones(1, C, device="cuda"), torch.zeros(1, C, device="cuda")
                red, nred, gco, group_n, app, napp, _ = me._bn_chunks((0, rows), x.device, C)
                sums = torch.zeros(me.BN_SLOTS * 2 * C, device="cuda")
                lib.call("cg3d_bn_sums", ptr(xf), ptr(red), c_int64(nred), c_int32(1), c_int32(C), ptr(sums), S())
                dsums = torch.zeros(me.BN_SLOTS * 2 * C, device="cuda")
                mean, var = torch.empty(1, C, device="cuda"), torch.empty(1, C, device="cuda")
                y = torch.empty(rows, C, dtype=dt, device="cuda")
                y16 = None if s16 else torch.empty(rows, C, dtype=torch.int16, device="cuda")
                dx = torch.empty(rows, C, dtype=dt, device="cuda")
                dx16 = None if s16 else torch.empty(rows, C, dtype=torch.int16, device="cuda")
                dres = torch.empty(rows, C, dtype=dt, device="cuda") if res else None
                dbeta, dgamma = torch.empty(1, C, device="cuda"), torch.empty(1, C, device="cuda")
                act = 1 | (0x100 if s16 else 0)
                f_apply = lambda: lib.call("cg3d_bn_apply_sums", ptr(x), ptr(r), ptr(app), c_int64(napp), c_int32(1), c_int32(C), ptr(sums),
                                           ptr(group_n), c_float(1e-5), ptr(gamma), ptr(beta), c_int32(act), ptr(y), ptr(y16), ptr(mean), ptr(var),
                                           ptr(None), ptr(None), ptr(None), c_float(0.1), S())
                f_red = lambda: lib.call("cg3d_bn_bwd_sums", ptr(dy), ptr(x), ptr(y), ptr(red), c_int64(nred), c_int32(1), c_int32(C), ptr(mean),
                                         ptr(var), c_float(1e-5), c_int32(act), ptr(dsums), S())
                f_bapp = lambda: lib.call("cg3d_bn_bwd_apply_sums", ptr(dy), ptr(x), ptr(y), ptr(app), c_int64(napp), c_int32(1), c_int32(C),
                                          ptr(mean), ptr(var), c_float(1e-5), ptr(gamma), ptr(dsums), ptr(group_n), c_int32(act), c_int32(1),
                                          ptr(dx), ptr(dx16), ptr(dres), ptr(dbeta), ptr(dgamma), S())
                t = [timeit(f) for f in (f_apply, f_red, f_bapp)]
                e = rows * C
                by = [e * (w * (2 + (1 if res else 0)) + (0 if s16 else 2)), e * w * 3, e * (w * (4 + (1 if res else 0)) + (0 if s16 else 2))]
                print("%8d %5d%s %5s | " % (rows, C, "+r" if res else "  ", "bf16" if s16 else "fp32") +
                      " | ".join("%8.1f (%6.0f)" % (tt, b / tt / 1e3) for tt, b in zip(t, by)), flush=True)


if __name__ == "__main__":
    main()
