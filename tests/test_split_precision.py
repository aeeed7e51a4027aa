"""The split precision ("bf16x3", me.PREC_SPLIT): fp32-accurate products on the bf16 matrix pipe for the two heads.

BASELINE.json configs[1] runs the BACKBONE in bf16; the reference's heads are fp32 (cagroup_head.py:227-282,
cagroup_roi_head.py:69-91).  With x = hi + lo (hi = bf16(x), lo = bf16(x - hi)) a product is
xhi whi + xlo whi + xhi wlo + O(2^-16): operand rows [hi | lo | hi], weights [Whi ; Whi ; Wlo], the bf16 kernels unchanged on
a three times longer contraction (include/cagroup3d_hip.h, cg3d_to_bf16_split).

CPU (-m "not gpu"): the oracle's split operands are what the definition says (bit level), the split convolution / linear /
grouped convolution agree with the oracle's fp32 arithmetic to 1e-4, the class-branch launch program equals the per-layer
path under it.  -m gpu: the device's split operands are bit-identical to the oracle's; the head's layer shapes on the
benchmark's own maps -- forward, data gradient, weight gradient -- against the FP32 oracle at rtol 1e-4."""
import ctypes
import os

import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, build_model, engine, me
from cagroup3d_amd._lib import ptr

RTOL, ATOL = 1e-4, 1e-5


def _bits(x):
    """bf16 bit patterns (int16 view) -> fp32 values."""
    return (x.to(torch.int32) << 16).view(torch.float32)


def _split_ref(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo


def _rows(lib, x):
    with _lib.use_library(lib):
        return me._to_split(x.contiguous())


def _weights(lib, w3, frag):
    K, cin, cout = w3.shape
    wt = torch.empty((K, cout, 3 * cin), dtype=torch.int16, device=w3.device)
    wp = torch.empty((K, cin, 3 * cout), dtype=torch.int16, device=w3.device)
    lib.call("cg3d_spconv_prep_weights_split", ptr(w3), ptr(None), ptr(wt), ptr(wp), ctypes.c_int32(1), ctypes.c_int64(K),
             ctypes.c_int32(cin), ctypes.c_int32(cout), ctypes.c_int32(1 if frag else 0), lib.stream())
    return wt, wp


def _frag_index(n, k, kdim):
    return ((((n >> 5) * (kdim >> 4) + (k >> 4)) * 64) + ((k >> 3) & 1) * 32 + (n & 31)) * 8 + (k & 7)


# ------------------------------------------------------------------------------------------------ CPU: the operands
def test_oracle_split_rows_are_hi_lo_hi(oracle):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 24, generator=g) * torch.logspace(-6, 6, 24).view(1, -1)
    x[0, 0], x[1, 1], x[2, 2] = 0.0, float("inf"), -1e-40          # zero, infinity (lo must be 0, not NaN), a denormal
    r = _bits(_rows(oracle, x)).view(37, 3, 24)
    hi, lo = _split_ref(x)
    lo[1, 1] = 0.0
    assert torch.equal(r[:, 0], hi) and torch.equal(r[:, 2], hi) and torch.equal(r[:, 1], lo)
    fin = torch.isfinite(x)
    rel = ((r[:, 0] + r[:, 1] - x).abs()[fin] / x.abs().clamp(min=1e-30)[fin]).max()
    assert rel < 2.0 ** -16, rel                 # 16 significant bits survive the split


@pytest.mark.parametrize("frag", [False, True])
def test_oracle_split_weights_are_the_three_part_operands(oracle, frag):
    g = torch.Generator().manual_seed(1)
    K, cin, cout = 3, 64, 32
    w = torch.randn(K, cin, cout, generator=g)
    wt, wp = _weights(oracle, w, frag)
    hi, lo = _split_ref(w)
    parts = (hi, hi, lo)
    wt, wp = _bits(wt).view(K, -1), _bits(wp).view(K, -1)
    for k in range(K):
        for ci in (0, 7, 8, 33, 63):
            for co in (0, 5, 31):
                for p in range(3):
                    it = _frag_index(co, p * cin + ci, 3 * cin) if frag else co * 3 * cin + p * cin + ci
                    ip = _frag_index(ci, p * cout + co, 3 * cout) if frag else ci * 3 * cout + p * cout + co
                    assert wt[k, it] == parts[p][k, ci, co] and wp[k, ip] == parts[p][k, ci, co]


def test_oracle_table_form_writes_the_same_split_operands(oracle):
    g = torch.Generator().manual_seed(2)
    with _lib.use_library(oracle):
        for frag in (False, True):
            K, cin, cout = 2, 128, 64
            w = torch.randn(K, cin, cout, generator=g)
            want_t, want_p = _weights(oracle, w, frag)
            got_t, got_p = torch.zeros_like(want_t), torch.zeros_like(want_p)
            rows = []
            per = cin * cout
            for k in range(K):
                for t in range((cin // 64) * (cout // 64)):
                    rows.append([w.data_ptr() + k * per * 4, got_t.data_ptr() + k * 3 * per * 2, got_p.data_ptr() + k * 3 * per * 2,
                                 cin, cout, t | ((3 << 29) if frag else 0) | (1 << 28)])
            tab = torch.tensor(rows, dtype=torch.int64)
            oracle.call("cg3d_spconv_prep_weights_bf16_table", ptr(tab), ctypes.c_int64(len(rows)), oracle.stream())
            assert torch.equal(got_t, want_t) and torch.equal(got_p, want_p)


# ------------------------------------------------------------------------------------------------ CPU: the layers
def _conv_case(dev, prec, seed=0, n=3000, cin=64, cout=128, ks=3):
    g = torch.Generator().manual_seed(seed)
    coords = torch.cat([torch.zeros(n, 1), torch.randint(0, 14, (n, 3), generator=g).float()], 1).to(dev)
    with me.precision_scope(prec):
        sp = me.SparseTensor(features=torch.zeros(n, 1, device=dev), coordinates=coords)
        km = sp.coordinate_manager.kernel_map(sp.coordinate_map_key, sp.coordinate_map_key, ks, 1, False)
        x = torch.randn(km.n_in, cin, generator=g).to(dev).requires_grad_(True)
        w = (torch.randn(ks ** 3, cin, cout, generator=g) / (cin * 27) ** 0.5).to(dev).requires_grad_(True)
        dy = torch.randn(km.n_out, cout, generator=g).to(dev)
        y = me.SparseConvFunction.apply(x, w, None, km)
        (y * dy).sum().backward()
    return y.detach().cpu(), x.grad.cpu(), w.grad.cpu()


def _close(got, want, names, rtol=RTOL, atol=ATOL):
    for nm, a, r in zip(names, got, want):
        scale = max(float(r.abs().max()), 1.0)
        torch.testing.assert_close(a, r, rtol=rtol, atol=atol * scale, msg=lambda m: nm + ": " + m)


def test_oracle_split_convolution_matches_fp32(oracle):
    with _lib.use_library(oracle):
        want = _conv_case("cpu", 0)
        got = _conv_case("cpu", 3)
        plain = _conv_case("cpu", 1)
    _close(got, want, ("y", "dx", "dw"))
    # (and the plain bf16 operands do NOT meet that bound: the test would notice a split that silently fell back)
    assert float((plain[0] - want[0]).abs().max()) > 20 * float((got[0] - want[0]).abs().max())


def test_oracle_split_linear_matches_fp32(oracle):
    g = torch.Generator().manual_seed(5)
    x0, w0, b0, dy = torch.randn(700, 128, generator=g), torch.randn(128, 64, generator=g) / 11.3, torch.randn(64, generator=g), torch.randn(700, 64, generator=g)

    def run(prec):
        with me.precision_scope(prec):
            x, w, b = (t.clone().requires_grad_(True) for t in (x0, w0, b0))
            y = me.linear(x, w, b)
            y.backward(dy)
            return y.detach(), x.grad, w.grad, b.grad
    with _lib.use_library(oracle):
        _close(run(3), run(0), ("y", "dx", "dw", "db"))


def test_class_branch_program_equals_per_layer_path_in_the_split_precision(oracle):
    """The launch program of the class branches emits the same calls as the per-layer path under the split precision:
    three-part operands, contractions of 3 x 64 / 3 x 128 channels, three-pass weight gradients."""
    from test_engine import _class_branch_inputs, _class_branch_step, _l2
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 3
        os.environ["CG3D_ENGINE_ANY"] = "1"
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            head = model.dense_head.train()
            fine, coarse, feat, up = _class_branch_inputs(head, "cpu", base=30)
            state = {k: v.clone() for k, v in head.state_dict().items()}

            def run(use_program):
                head.load_state_dict(state)
                return _class_branch_step(head, fine, coarse, feat, up, use_program, 2)
            ref = run(False)
            before = engine.CLASS_STATS["program_passes"]
            got = run(True)
            assert engine.CLASS_STATS["program_passes"] == before + 1
            me.PRECISION = 0
            exact = run(False)
        finally:
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)
            me._WeightPlan.reset()
    assert _l2(ref[0], got[0]) <= 1e-5 and _l2(ref[1], got[1]) <= 1e-4, (_l2(ref[0], got[0]), _l2(ref[1], got[1]))
    for k in ref[2]:
        if float(ref[2][k].norm()) > 1e-3:
            assert _l2(ref[2][k], got[2][k]) <= 1e-4, (k, _l2(ref[2][k], got[2][k]))
    # and the split precision follows the fp32 arithmetic through four convolution + BatchNorm + ELU stages
    assert _l2(exact[0], got[0]) <= 1e-4 and _l2(exact[1], got[1]) <= 1e-3, (_l2(exact[0], got[0]), _l2(exact[1], got[1]))


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_device_split_operands_are_bit_identical_to_the_oracle(oracle, hip):
    g = torch.Generator().manual_seed(7)
    x = torch.randn(5003, 64, generator=g) * torch.logspace(-4, 4, 64).view(1, -1)
    assert torch.equal(_rows(hip, x.cuda()).cpu(), _rows(oracle, x))
    for frag in (False, True):
        for (K, cin, cout) in ((5, 64, 64), (2, 128, 256), (3, 32, 32)):
            w = torch.randn(K, cin, cout, generator=g)
            a = _weights(oracle, w, frag)
            b = _weights(hip, w.cuda(), frag)
            assert torch.equal(a[0], b[0].cpu()) and torch.equal(a[1], b[1].cpu()), (frag, K, cin, cout)
            # the table form (one launch per step for every recorded weight): full 64 x 64 tiles take the 16-byte path
            per = cin * cout
            wd = w.cuda()
            got_t, got_p = torch.zeros_like(b[0]), torch.zeros_like(b[1])
            rows = [[wd.data_ptr() + k * per * 4, got_t.data_ptr() + k * 3 * per * 2, got_p.data_ptr() + k * 3 * per * 2, cin, cout,
                     t | ((3 << 29) if frag else 0) | (1 << 28)]
                    for k in range(K) for t in range(-(-cin // 64) * -(-cout // 64))]
            tab = torch.tensor(rows, dtype=torch.int64).cuda()
            hip.call("cg3d_spconv_prep_weights_bf16_table", ptr(tab), ctypes.c_int64(len(rows)), hip.stream())
            assert torch.equal(got_t.cpu(), a[0]) and torch.equal(got_p.cpu(), a[1]), ("table", frag, K, cin, cout)


@pytest.mark.gpu
@pytest.mark.parametrize("ks", [9, 5])
def test_benchmark_class_map_convolution_in_the_split_precision_matches_the_fp32_oracle(oracle, hip, monkeypatch, ks):
    """The grouped 9^3 / 5^3 class-branch convolutions (cagroup_head.py:254-278) on a benchmark-shaped class map: the device in
    the split precision (k_spconv_tile2 on a 192-channel contraction, three-pass bf16 weight gradient) against the oracle's
    FP32 arithmetic -- forward, data gradient, weight gradient at rtol 1e-4."""
    from test_full_size_parity import _class_map_case, bench_coords
    coords = bench_coords()
    G, B = 18, 4
    vs = [0.08 + 0.02 * (c % 5) for c in range(G)] if ks == 9 else [0.30 + 0.05 * (c % 5) for c in range(G)]
    tile_calls, wgrad = [], []
    real_tile = me._conv_tile
    monkeypatch.setattr(me, "PRECISION", 3)
    with _lib.use_library(hip):
        monkeypatch.setattr(me, "_conv_tile", lambda *a, **k: (tile_calls.append((a[2].K, a[0].shape[1])), real_tile(*a, **k))[1])
        real_call = hip.call
        monkeypatch.setattr(hip, "call", lambda n, *a: (wgrad.append(int(a[-2].value)) if n == "cg3d_spconv_pairs_wgrad" else None, real_call(n, *a))[1])
        out = _class_map_case(coords.cuda(), B, G, ks, seed=ks, vs=vs)
        torch.cuda.synchronize()
        monkeypatch.setattr(me, "_conv_tile", real_tile)
        monkeypatch.setattr(hip, "call", real_call)
    monkeypatch.setattr(me, "PRECISION", 0)
    with _lib.use_library(oracle):
        ref = _class_map_case(coords, B, G, ks, seed=ks, vs=vs)
    assert ref[3] == out[3]
    assert tile_calls == [(ks ** 3, 192), (ks ** 3, 192)], tile_calls          # split rows of 64 channels on the tile kernel
    assert wgrad == [3], wgrad
    _close([o.cpu() for o in out[:3]], ref[:3], ("y", "dx", "dw"))


HEAD_LAYERS = [
    ("feature_offset 64->64 k3 @2", 2, 3, 1, False, 64, 64),
    ("RoI grid conv 128->128 k3 @4 (tile)", 4, 3, 1, False, 128, 128),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,in_stride,ks,cstride,transpose,cin,cout", HEAD_LAYERS, ids=[l[0] for l in HEAD_LAYERS])
def test_benchmark_head_layer_in_the_split_precision_matches_the_fp32_oracle(oracle, hip, monkeypatch, name, in_stride, ks, cstride,
                                                                            transpose, cin, cout):
    from test_full_size_parity import _layer_case, bench_coords
    coords = bench_coords()
    monkeypatch.setattr(me, "PRECISION", 3)
    with _lib.use_library(hip):
        out = _layer_case(coords.cuda(), in_stride, ks, cstride, transpose, cin, cout, seed=cin + cout + ks)
        torch.cuda.synchronize()
    monkeypatch.setattr(me, "PRECISION", 0)
    with _lib.use_library(oracle):
        ref = _layer_case(coords, in_stride, ks, cstride, transpose, cin, cout, seed=cin + cout + ks)
    assert ref[3] == out[3]
    _close([o.cpu() for o in out[:3]], ref[:3], ("y", "dx", "dw"))


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout", [(155773, 64, 64), (53718, 128, 64), (512, 43904, 128)])
def test_benchmark_linear_layer_in_the_split_precision_matches_fp32(hip, n, cin, cout):
    """me.linear under the split precision (cg3d_linear_fwd on 3 cin channels, three-pass weight gradient): the vote-offset
    block's 1x1x1 convolutions, the class branches' fuse layer shape, and the RoI head's 7^3 contraction as the flattened
    product [R, 343 * 128] x [343 * 128, 128] -- against the fp64 product of the same fp32 operands."""
    g = torch.Generator().manual_seed(n + cin)
    x = torch.randn(n, cin, generator=g).cuda()
    w = (torch.randn(cin, cout, generator=g) / cin ** 0.5).cuda()
    dy = torch.randn(n, cout, generator=g).cuda()
    calls = []
    orig = hip.call
    hip.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    try:
        with _lib.use_library(hip), me.precision_scope(3):
            xs, ws = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            y = me.linear(xs, ws)
            y.backward(dy)
    finally:
        hip.call = orig
    assert calls.count("cg3d_linear_fwd") >= 2 and "cg3d_to_bf16_split" in calls, calls
    want = (x.double() @ w.double(), dy.double() @ w.double().t(), x.double().t() @ dy.double())
    for nm, a, r in zip(("y", "dx", "dw"), (y.detach(), xs.grad, ws.grad), want):
        scale = max(float(r.abs().max()), 1.0)
        torch.testing.assert_close(a.double(), r, rtol=RTOL, atol=ATOL * scale * (4 if nm == "dw" else 1), msg=lambda m: nm + ": " + m)


@pytest.mark.gpu
def test_class_branch_program_in_the_split_precision_on_the_device(hip):
    """engine.run_class_branches under the split precision against the per-layer path under it, and against the per-layer
    path in fp32 (what the split stands in for)."""
    from test_engine import _class_branch_inputs, _class_branch_step, _l2
    prec, me.PRECISION = me.PRECISION, 3
    try:
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        head = model.dense_head.cuda().train()
        fine, coarse, feat, up = _class_branch_inputs(head, "cuda")
        state = {k: v.clone() for k, v in head.state_dict().items()}

        def run(use_program):
            head.load_state_dict(state)
            return _class_branch_step(head, fine, coarse, feat, up, use_program, 2)
        for e in (False, False, True, True):
            run(e)
        ref, ref2 = run(False), run(False)
        before = engine.CLASS_STATS["program_passes"]
        got = run(True)
        assert engine.CLASS_STATS["program_passes"] == before + 1
        me.PRECISION = 0
        exact = run(False)
    finally:
        me.PRECISION = prec
        me._WeightPlan.reset()
    assert _l2(ref[0], got[0]) <= 3 * _l2(ref[0], ref2[0]) + 1e-4, (_l2(ref[0], got[0]), _l2(ref[0], ref2[0]))
    assert _l2(ref[1], got[1]) <= 3 * _l2(ref[1], ref2[1]) + 1e-3
    # split vs fp32 through the four stages: within the run-to-run noise of the fp32 atomics + the split's own 1e-5
    assert _l2(exact[0], got[0]) <= 2e-4, _l2(exact[0], got[0])
    assert _l2(exact[1], got[1]) <= 2e-3, _l2(exact[1], got[1])
    bad = {k: _l2(exact[2][k], got[2][k]) for k in exact[2] if float(exact[2][k].norm()) > 1e-3 and _l2(exact[2][k], got[2][k]) > 5e-3}
    assert not bad, bad
