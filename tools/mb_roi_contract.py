"""The per-RoI 7^3 -> centre contraction alone (R x 343 x 128 -> 128): me.roi_contract (bf16 gather + cg3d_linear_fwd with the
contraction split over the chip) against the fp32 library form, forward and forward + backward.  dev tool; GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from cagroup3d_amd import me  # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


R, G, C, C2, N = int(os.environ.get("R", 512)), 343, 128, 128, 60000
dev = torch.device("cuda", 0)
feats = torch.randn(N, C, device=dev).requires_grad_(True)
idx = torch.randint(0, N, (R * G,), device=dev)
w = (torch.randn(G, C, C2, device=dev) * 0.01).requires_grad_(True)
dy = torch.randn(R, C2, device=dev)


def fwd():
    with torch.no_grad():
        return me.roi_contract(feats, idx, w)


def both():
    feats.grad = w.grad = None
    me.roi_contract(feats, idx, w).backward(dy)


if os.environ.get("UNITS"):          # one configuration (under rocprofv3 --kernel-trace --stats: per-kernel times)
    me.PRECISION, me.ROI_CONTRACT_UNITS = 1, int(os.environ["UNITS"])
    me.ROI_CONTRACT_PARTIALS = os.environ.get("PARTIALS", "1") != "0"
    print("fwd %7.1f us   fwd+bwd %7.1f us" % (timed(fwd), timed(both)))
    sys.exit(0)
me.PRECISION = 0
print("fp32 library form:            fwd %7.1f us   fwd+bwd %7.1f us" % (timed(fwd), timed(both)))
me.PRECISION = 1
me.HEAD_PRECISION = me.heads_from_env()
for partials in (True, False):
    for units in (128, 256, 512, 1024, 2048):
        me.ROI_CONTRACT_PARTIALS, me.ROI_CONTRACT_UNITS = partials, units
        print("bf16, units %4d, %-8s      fwd %7.1f us   fwd+bwd %7.1f us" % (units, "partials" if partials else "atomics", timed(fwd), timed(both)))
