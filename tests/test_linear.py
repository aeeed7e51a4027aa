"""cg3d_linear_fwd: the streaming bf16 MFMA product behind the kernel-size-1 convolutions (csrc/linear.hip).

CPU (-m "not gpu"): the oracle's restatement against a torch product of the bf16-rounded operands (pins the fragment order
of the weights as the kernel reads them).  -m gpu: the HIP kernel against the oracle -- ragged row counts, both channel
widths of a workgroup, the split contraction, the BatchNorm statistics epilogue -- and LinearFunction end to end."""
import ctypes

import pytest
import torch

from cagroup3d_amd import _lib, me
from cagroup3d_amd._lib import ptr

RTOL, ATOL = 1e-4, 1e-5          # same operands, fp32 accumulation: only the summation order differs


def _bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _run(lib, x, w, bias, ksplit=1, stats=False, transposed_product=False, partials=False):
    """Y = X @ W through cg3d_linear_fwd on `lib` (transposed_product: dX = X @ W^T on the plain fragment copy)."""
    dev = x.device
    n, cin = x.shape
    K, wi, wo = 1, w.shape[0], w.shape[1]
    with _lib.use_library(lib):
        x16 = me._to_bf16(x.contiguous())
        wt = torch.empty((K, wo, wi), dtype=torch.int16, device=dev)
        wp = torch.empty((K, wi, wo), dtype=torch.int16, device=dev)
        w3 = w.contiguous().view(1, wi, wo)
        lib.call("cg3d_spconv_prep_weights_frag", ptr(w3), ptr(None), ptr(wt), ptr(wp), ctypes.c_int32(1), ctypes.c_int64(K),
                 ctypes.c_int32(wi), ctypes.c_int32(wo), lib.stream())
        cout = wi if transposed_product else wo
        y = torch.full((n, cout), float("nan"), dtype=torch.float32, device=dev)
        st = torch.zeros((me.BN_SLOTS, 2, cout), dtype=torch.float32, device=dev) if stats else None
        part = torch.full((ksplit, n, cout), float("nan"), dtype=torch.float32, device=dev) if partials else None
        lib.call("cg3d_linear_fwd", ptr(x16), ptr(wp if transposed_product else wt), ptr(bias), ptr(y), ctypes.c_int64(n),
                 ctypes.c_int32(cin), ctypes.c_int32(cout), ctypes.c_int32(ksplit), ptr(st), ptr(part), lib.stream())
    return y, st


def _case(n, cin, cout, seed, dev="cpu"):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, generator=g)
    w = torch.randn(cin, cout, generator=g) / cin ** 0.5
    b = torch.randn(cout, generator=g)
    return x.to(dev), w.to(dev), b.to(dev)


@pytest.mark.parametrize("n,cin,cout", [(1, 64, 64), (130, 64, 128), (257, 192, 64), (300, 128, 256)])
def test_oracle_linear_is_the_product_of_the_bf16_rounded_operands(oracle, n, cin, cout):
    x, w, b = _case(n, cin, cout, 0)
    y, st = _run(oracle, x, w, b, stats=True)
    ref = _bf16(x).double() @ _bf16(w).double() + b.double()
    torch.testing.assert_close(y.double(), ref, rtol=RTOL, atol=ATOL)
    torch.testing.assert_close(st.sum(0)[0].double(), ref.sum(0), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(st.sum(0)[1].double(), (ref * ref).sum(0), rtol=1e-4, atol=1e-3)
    # the data-gradient form: the plain fragment copy, roles of the channel counts swapped
    dy = torch.randn(n, cout, generator=torch.Generator().manual_seed(1))
    dx, _ = _run(oracle, dy, w, None, transposed_product=True)
    torch.testing.assert_close(dx.double(), _bf16(dy).double() @ _bf16(w).double().t(), rtol=RTOL, atol=ATOL)


def test_oracle_linear_rejects_what_the_kernel_rejects(oracle):
    x, w, b = _case(16, 64, 64, 0)
    for bad in ((48, 64, 1), (64, 96, 1), (64, 64, 2)):
        with pytest.raises(_lib.CG3DError):
            oracle.call("cg3d_linear_fwd", ptr(torch.zeros(16, bad[0], dtype=torch.int16)), ptr(torch.zeros(64 * 64, dtype=torch.int16)),
                        ptr(None), ptr(torch.zeros(16, bad[1])), ctypes.c_int64(16), ctypes.c_int32(bad[0]), ctypes.c_int32(bad[1]),
                        ctypes.c_int32(bad[2]), ptr(None), ptr(None), oracle.stream())


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout,ksplit", [(1, 64, 64, 1), (127, 64, 128, 1), (1000, 128, 64, 1), (4099, 256, 256, 1),
                                               (20000, 64, 64, 1), (20000, 512, 128, 1), (20000, 128, 192, 1), (0, 64, 64, 1),
                                               (300, 1024, 128, 4), (64, 43904, 128, 49)])
def test_hip_linear_matches_oracle(oracle, hip, n, cin, cout, ksplit):
    x, w, b = _case(n, cin, cout, 2)
    want, want_st = _run(oracle, x, w, b, ksplit=ksplit, stats=ksplit == 1)
    got, got_st = _run(hip, x.cuda(), w.cuda(), b.cuda(), ksplit=ksplit, stats=ksplit == 1)
    tol = dict(rtol=RTOL, atol=ATOL) if cin <= 1024 else dict(rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(got.cpu(), want, **tol)
    if ksplit > 1:                                  # the same split through the scratch of stored partial products
        got_p, _ = _run(hip, x.cuda(), w.cuda(), b.cuda(), ksplit=ksplit, partials=True)
        torch.testing.assert_close(got_p.cpu(), want, **tol)
    if ksplit == 1:
        scale = max(1.0, float(n)) ** 0.5
        torch.testing.assert_close(got_st.sum(0).cpu(), want_st.sum(0), rtol=1e-4, atol=1e-3 * scale)
    if n:
        dy = torch.randn(n, cout, generator=torch.Generator().manual_seed(3))
        if cin <= 1024:
            want_dx, _ = _run(oracle, dy, w, None, transposed_product=True)
            got_dx, _ = _run(hip, dy.cuda(), w.cuda(), None, transposed_product=True)
            torch.testing.assert_close(got_dx.cpu(), want_dx, rtol=RTOL, atol=ATOL)


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout", [(5000, 64, 128), (20000, 128, 64), (33, 1024, 128), (284, 1024, 128), (1, 64, 64), (700, 256, 512)])
def test_linear_function_own_kernel_against_the_library_path(hip, monkeypatch, n, cin, cout):
    """me.linear forward / dX / dW / db with the streaming kernel == the library path on bf16-rounded operands."""
    x, w, b = _case(n, cin, cout, 4, "cuda")
    x, w, gy = _bf16(x), _bf16(w), _bf16(torch.randn(n, cout, device="cuda"))      # operands both paths represent exactly
    out = {}
    with _lib.use_library(hip):
        monkeypatch.setattr(me, "PRECISION", 1)
        for own in (True, False):
            monkeypatch.setattr(me, "LINEAR_KERNEL", own)
            xs, ws, bs = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = me.linear(xs, ws, bs)
            y.backward(gy)
            out[own] = (y.detach(), xs.grad, ws.grad, bs.grad)
    torch.testing.assert_close(out[True][0], out[False][0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out[True][1], out[False][1], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out[True][2], out[False][2], rtol=1e-3, atol=1e-2)      # n-row contraction, different split
    torch.testing.assert_close(out[True][3], gy.sum(0), rtol=1e-4, atol=1e-3)


# ------------------------------------------------------------------ the per-RoI 7^3 -> centre contraction (me.roi_contract)
def _roi_case(n_src, R, G, C, C2, seed):
    g = torch.Generator().manual_seed(seed)
    feats = _bf16(torch.randn(n_src, C, generator=g))
    idx = torch.randint(0, n_src, (R * G,), generator=g)
    idx[:G] = 7                                              # a degenerate RoI: every grid point on one voxel row
    w = _bf16(torch.randn(G, C, C2, generator=g) * 0.02)
    dy = _bf16(torch.randn(R, C2, generator=g))
    return feats, idx, w, dy


def _roi_run(feats, idx, w, dy, fused):
    prec, me.PRECISION = me.PRECISION, 1 if fused else 0
    try:
        f, k = feats.clone().requires_grad_(True), w.clone().requires_grad_(True)
        assert me.RoiContractFunction.available(f, k) == fused
        y = me.roi_contract(f, idx, k)
        (y * dy).sum().backward()
        return y.detach(), f.grad, k.grad
    finally:
        me.PRECISION = prec


@pytest.mark.parametrize("n_src,R,G,C,C2", [(500, 5, 27, 64, 64), (800, 130, 8, 128, 64), (300, 3, 343, 128, 128)])
def test_oracle_roi_contraction_is_gather_times_kernel(oracle, n_src, R, G, C, C2):
    """bf16-exact operands: the fused form (bf16 gather, split contraction, pair-list weight gradient) == the fp32 form."""
    case = _roi_case(n_src, R, G, C, C2, 5)
    with _lib.use_library(oracle):
        got, want = _roi_run(*case, True), _roi_run(*case, False)
    for name, a, b in zip(("pooled", "d features", "d kernel"), got, want):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max())), msg=lambda m: name + ": " + m)


@pytest.mark.gpu
@pytest.mark.parametrize("n_src,R,G,C,C2", [(500, 5, 27, 64, 64), (3000, 130, 8, 128, 64), (20000, 512, 343, 128, 128), (2000, 1, 343, 128, 128)])
def test_hip_roi_contraction_matches_oracle(oracle, hip, n_src, R, G, C, C2):
    case = _roi_case(n_src, R, G, C, C2, 6)
    with _lib.use_library(oracle):
        want = _roi_run(*case, True)
    with _lib.use_library(hip):
        got = _roi_run(*[t.cuda() for t in case], True)
    for name, a, b in zip(("pooled", "d features", "d kernel"), got, want):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-4, atol=1e-4 * max(1.0, float(b.abs().max())), msg=lambda m: name + ": " + m)


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout,bias", [(512, 128, 256, False), (512, 256, 256, True), (77, 64, 128, True), (3000, 256, 64, False)])
def test_linear_with_the_weight_stored_like_nn_linear(oracle, hip, n, cin, cout, bias):
    """me.linear_t (the RoI head's FC layers): y = x @ w^T + b, dx, dw, db on the own kernel == torch's F.linear on operands
    bf16 represents exactly, and == the oracle running the same path."""
    g = torch.Generator().manual_seed(n + cin)
    x, w, dy = _bf16(torch.randn(n, cin, generator=g)), _bf16(torch.randn(cout, cin, generator=g) / cin ** 0.5), _bf16(torch.randn(n, cout, generator=g))
    b = torch.randn(cout, generator=g) if bias else None

    def run(lib, dev, own):
        prec, me.PRECISION = me.PRECISION, 1 if own else 0
        try:
            with _lib.use_library(lib):
                xs, ws = x.to(dev).clone().requires_grad_(True), w.to(dev).clone().requires_grad_(True)
                bs = b.to(dev).clone().requires_grad_(True) if bias else None
                y = me.linear_t(xs, ws, bs)
                y.backward(dy.to(dev))
                return [y.detach().cpu(), xs.grad.cpu(), ws.grad.cpu()] + ([bs.grad.cpu()] if bias else [])
        finally:
            me.PRECISION = prec
    got, ref, want = run(hip, "cuda", True), run(hip, "cuda", False), run(oracle, "cpu", True)
    for name, a, r, o in zip(("y", "dx", "dw", "db"), got, ref, want):
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(a, r, rtol=1e-4, atol=1e-4 * scale, msg=lambda m: name + " vs F.linear: " + m)
        torch.testing.assert_close(a, o, rtol=1e-4, atol=1e-4 * scale, msg=lambda m: name + " vs oracle: " + m)
