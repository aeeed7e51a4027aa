"""Activations stored as bf16 in the backbone's launch program (engine.ACT_BF16; include/cagroup3d_hip.h: CG3D_BN_STORE_BF16,
CG3D_TILE_OUT_BF16, CG3D_LINEAR_OUT_BF16, cg3d_from_bf16) -- BASELINE.json configs[1] "bf16 backbone".

The storage flag changes WHERE a value is rounded, not how it is computed: operands are widened to fp32, statistics and
per-channel constants stay fp32 / fp64, results are rounded to nearest even on the store.  So:
  CPU: the oracle's bf16-storage BatchNorm calls equal its fp32 calls on the widened inputs, rounded (bit for bit);
  gpu: the device's calls against the oracle's (a bf16 ulp where the fp32 statistics differ in their last bits), the tile /
       linear kernels' bf16 rows equal their own fp32 rows rounded (bit for bit: the sums are formed identically), and the
       whole backbone pass with bf16 storage against the same pass with fp32 storage."""
import ctypes
import os

import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, build_model, engine, me
from cagroup3d_amd._lib import ptr

S16 = 0x100


def _b16(x):
    """fp32 -> int16 view of the bf16 rows (round to nearest even)."""
    return x.to(torch.bfloat16).view(torch.int16)


def _f32(x16):
    return x16.view(torch.bfloat16).float()


def _bn_case(lib, dev, n, c, G, act, s16, seed=0):
    """One grouped BatchNorm layer forward + backward through the C-ABI; bf16 storage or fp32 storage of the SAME values."""
    g = torch.Generator().manual_seed(seed)
    bounds = tuple(int(v) for v in np.linspace(0, n, G + 1).astype(np.int64))
    x = _f32(_b16(torch.randn(n, c, generator=g) * 2 + 0.5))
    res = _f32(_b16(torch.randn(n, c, generator=g)))
    dy = _f32(_b16(torch.randn(n, c, generator=g)))
    gamma, beta = torch.rand(G, c, generator=g) + 0.5, torch.randn(G, c, generator=g)
    with _lib.use_library(lib):
        red, nred, _, group_n, app, napp, _ = me._bn_chunks(bounds, torch.device(dev), c)
        xs, rs, dys, ga, be = (t.to(dev) for t in (x, res, dy, gamma, beta))
        sums = torch.zeros(me.BN_SLOTS * 2 * G * c, device=dev)
        lib.call("cg3d_bn_sums", ptr(xs), ptr(red), ctypes.c_int64(nred), ctypes.c_int32(G), ctypes.c_int32(c), ptr(sums), lib.stream())
        mean, var = torch.zeros(G, c, device=dev), torch.zeros(G, c, device=dev)
        flag = S16 if s16 else 0
        xin, rin, dyin = (_b16(t) if s16 else t for t in (xs, rs, dys))
        y = torch.empty((n, c), dtype=torch.int16 if s16 else torch.float32, device=dev)
        lib.call("cg3d_bn_apply_sums", ptr(xin), ptr(rin), ptr(app), ctypes.c_int64(napp), ctypes.c_int32(G), ctypes.c_int32(c), ptr(sums),
                 ptr(group_n), ctypes.c_float(1e-5), ptr(ga), ptr(be), ctypes.c_int32(act | flag), ptr(y), ptr(None), ptr(mean), ptr(var),
                 ptr(None), ptr(None), ptr(None), ctypes.c_float(0.1), lib.stream())
        # the backward reads the STORED output: bf16 storage hands it the rounded rows, so give the fp32 run the same ones
        yb = y if s16 else _f32(_b16(y))
        dsums = torch.zeros(me.BN_SLOTS * 2 * G * c, device=dev)
        lib.call("cg3d_bn_bwd_sums", ptr(dyin), ptr(xin), ptr(yb), ptr(red), ctypes.c_int64(nred), ctypes.c_int32(G), ctypes.c_int32(c),
                 ptr(mean), ptr(var), ctypes.c_float(1e-5), ctypes.c_int32(act | flag), ptr(dsums), lib.stream())
        dx = torch.empty_like(y)
        dr = torch.empty_like(y)
        dbeta, dgamma = torch.zeros(G, c, device=dev), torch.zeros(G, c, device=dev)
        lib.call("cg3d_bn_bwd_apply_sums", ptr(dyin), ptr(xin), ptr(yb), ptr(app), ctypes.c_int64(napp), ctypes.c_int32(G), ctypes.c_int32(c),
                 ptr(mean), ptr(var), ctypes.c_float(1e-5), ptr(ga), ptr(dsums), ptr(group_n), ctypes.c_int32(act | flag), ctypes.c_int32(1),
                 ptr(dx), ptr(None), ptr(dr), ptr(dbeta), ptr(dgamma), lib.stream())
    out = [t.cpu() for t in (y, dx, dr)]
    if s16:
        out = [_f32(t) for t in out]
    return out + [mean.cpu(), var.cpu(), dbeta.cpu(), dgamma.cpu()]


@pytest.mark.parametrize("n,c,G,act", [(300, 64, 1, 1), (257, 128, 3, 2), (64, 1024, 1, 0)])
def test_oracle_bf16_storage_is_the_fp32_call_rounded(oracle, n, c, G, act):
    got = _bn_case(oracle, "cpu", n, c, G, act, True)
    want = _bn_case(oracle, "cpu", n, c, G, act, False)
    for nm, a, r in zip(("y", "dx", "dres"), got[:3], want[:3]):
        assert torch.equal(a, _f32(_b16(r))), nm
    for nm, a, r in zip(("mean", "var", "dbeta", "dgamma"), got[3:], want[3:]):
        assert torch.equal(a, r), nm


def test_oracle_from_bf16_widens(oracle):
    x = torch.randn(16, 24)
    out = torch.empty(16, 24)
    x16 = _b16(x)
    oracle.call("cg3d_from_bf16", ptr(x16), ptr(out), ctypes.c_int64(x.numel()), oracle.stream())
    assert torch.equal(out, _f32(x16))


def _ulp_close(a, r, ulps=1.0):
    """|a - r| <= `ulps` bf16 units in the last place of the larger magnitude (+ a floor for values near zero)."""
    tol = ulps * torch.maximum(a.abs(), r.abs()) * 2.0 ** -7 + 1e-6
    bad = (a - r).abs() > tol
    return int(bad.sum()), float(((a - r).abs() / tol).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,G,act", [(155773, 64, 1, 1), (23015, 256, 1, 1), (53718, 64, 18, 2), (1229, 1024, 1, 0)])
def test_device_bf16_storage_batchnorm_matches_the_oracle(oracle, hip, n, c, G, act):
    got = _bn_case(hip, "cuda", n, c, G, act, True)
    want = _bn_case(oracle, "cpu", n, c, G, act, True)
    for nm, a, r in zip(("mean", "var", "dbeta", "dgamma"), got[3:], want[3:]):
        scale = max(float(r.abs().max()), 1.0)
        # dbeta / dgamma sum dz = dy * act'(y) over the rows: where the fp32 statistics of the two sides differ in their last
        # bits an output within rounding of zero is 0 on one side and a tiny positive number on the other, and that row's dy
        # (|dy| ~ 1, |xhat| up to ~4) enters one sum and not the other -- a handful of rows per layer
        atol = 1e-5 * scale if not nm.startswith("d") else max(3e-4 * scale, 8.0 if act else 3e-4 * scale)
        torch.testing.assert_close(a, r, rtol=1e-4, atol=atol, msg=lambda m: nm + ": " + m)
    for nm, a, r in zip(("y", "dx", "dres"), got[:3], want[:3]):
        nbad, worst = _ulp_close(a, r, 1.0 if nm != "dx" else 2.0)
        # (dx subtracts two nearly equal terms where dz is small: a last-bit difference of the fp32 statistics moves more bits there)
        assert nbad <= (1e-4 if nm != "dx" else 2e-3) * a.numel(), (nm, nbad, worst)


@pytest.mark.gpu
def test_device_from_bf16_and_tile_and_linear_bf16_rows(hip):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4096, 64, generator=g).cuda()
    out = torch.empty_like(x)
    x16 = _b16(x)
    hip.call("cg3d_from_bf16", ptr(x16), ptr(out), ctypes.c_int64(x.numel()), hip.stream())
    assert torch.equal(out, _f32(x16))
    # linear: Y = X W + b as fp32 rows and as bf16 rows
    n, cin, cout = 5000, 128, 256
    xs, w, b = torch.randn(n, cin, generator=g).cuda(), (torch.randn(cin, cout, generator=g) / cin ** 0.5).cuda(), torch.randn(cout, generator=g).cuda()
    with _lib.use_library(hip), me.precision_scope(1):
        x16 = me._to_bf16(xs)
        wt, _ = me._prep_frag(w.view(1, cin, cout), True, False)
        y32 = torch.empty(n, cout, device="cuda")
        y16 = torch.empty(n, cout, dtype=torch.int16, device="cuda")
        st32, st16 = torch.zeros(me.BN_SLOTS * 2 * cout, device="cuda"), torch.zeros(me.BN_SLOTS * 2 * cout, device="cuda")
        for y, st, flag in ((y32, st32, 0), (y16, st16, 0x10000)):
            hip.call("cg3d_linear_fwd", ptr(x16), ptr(wt), ptr(b), ptr(y), ctypes.c_int64(n), ctypes.c_int32(cin), ctypes.c_int32(cout),
                     ctypes.c_int32(1 | flag), ptr(st), ptr(None), hip.stream())
    assert torch.equal(_f32(y16), _f32(_b16(y32)))
    torch.testing.assert_close(st16.view(me.BN_SLOTS, -1).sum(0), st32.view(me.BN_SLOTS, -1).sum(0), rtol=1e-5, atol=1e-2)
    # tile convolution on a random map
    coords = torch.cat([torch.zeros(20000, 1), torch.randint(0, 40, (20000, 3), generator=g).float()], 1).cuda()
    with _lib.use_library(hip), me.precision_scope(1):
        sp = me.SparseTensor(features=torch.zeros(20000, 1, device="cuda"), coordinates=coords)
        km = sp.coordinate_manager.kernel_map(sp.coordinate_map_key, sp.coordinate_map_key, 3, 1, False)
        plan = km.tile_plan(False)
        xf = torch.randn(km.n_in, 64, generator=g).cuda()
        w3 = (torch.randn(27, 64, 128, generator=g) / 40).cuda()
        x16 = me._to_bf16(xf)
        wt, _ = me._prep_frag(w3, True, False)
        outs = []
        for flag, dt in ((0, torch.float32), (2, torch.int16)):
            y = torch.empty(km.n_out, 128, dtype=dt, device="cuda")
            st = torch.zeros(me.BN_SLOTS * 2 * 128, device="cuda")
            hip.call("cg3d_spconv_tile_fwd", ptr(x16), ptr(wt), ptr(plan.slots), ptr(plan.live), ptr(plan.pass_tab), ptr(plan.npass),
                     ptr(plan.ulist), ctypes.c_int32(plan.maxpass), ctypes.c_int32(plan.ucap), ptr(plan.tiles), ctypes.c_int64(plan.ntile),
                     ptr(plan.order), ptr(None), ptr(y), ctypes.c_int64(km.n_in), ctypes.c_int64(km.n_out), ctypes.c_int32(27), ctypes.c_int32(64),
                     ctypes.c_int32(128), ctypes.c_int32(1), ctypes.c_int32(flag), ptr(st), hip.stream())
            outs.append((y, st))
    assert float(outs[0][0].abs().max()) > 0
    assert torch.equal(_f32(outs[1][0]), _f32(_b16(outs[0][0])))
    torch.testing.assert_close(outs[1][1].view(me.BN_SLOTS, -1).sum(0), outs[0][1].view(me.BN_SLOTS, -1).sum(0), rtol=1e-5, atol=1e-2)


def _l2(a, b):
    return float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("cfgname,n", [("S5k", 2), ("S50k", 1)])
def test_backbone_pass_with_bf16_storage_follows_the_fp32_storage_pass(hip, cfgname, n):
    """The whole BiResNet program, forward + backward, with activations / gradients stored as bf16 against the same program
    with fp32 storage (bf16 operand copies): same maps, same kernels, same weights.  What differs is one rounding per
    stored tensor (2^-9 relative), which this untrained BatchNorm-heavy net amplifies like it amplifies the run-to-run noise of
    the atomics -- the yardstick is that noise (two fp32-storage runs against each other)."""
    from test_engine import _backbone_step
    prec, me.PRECISION = me.PRECISION, 1
    keep = engine.ACT_BF16
    try:
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        model = model.cuda()
        batch = build_model.synthetic_batch(cfgname, n, device="cuda")
        state = {k: v.clone() for k, v in model.state_dict().items()}

        def run(act16):
            engine.ACT_BF16 = act16
            model.load_state_dict(state)
            return _backbone_step(model, batch, True, "cuda")
        for a in (False, False, True, True):
            run(a)
        before = engine.STATS["program_passes"]
        ref, ref2 = run(False), run(False)
        got = run(True)
        assert engine.STATS["program_passes"] == before + 3
    finally:
        me.PRECISION = prec
        engine.ACT_BF16 = keep
    assert torch.equal(ref[0], got[0])
    noise, err = _l2(ref[1], ref2[1]), _l2(ref[1], got[1])
    print("output: bf16 storage vs fp32 storage %.3e (fp32 storage run to run %.3e)" % (err, noise))
    assert err <= 3 * noise + 2e-2, (err, noise)
    worst = {}
    for k in ref[2]:
        if float(ref[2][k].norm()) > 1e-3:
            nz, e = _l2(ref[2][k], ref2[2][k]), _l2(ref[2][k], got[2][k])
            worst[k] = (e, nz)
    bad = {k: v for k, v in worst.items() if v[0] > 3 * v[1] + 0.25}
    top = sorted(worst.items(), key=lambda kv: -kv[1][0])[:5]
    print("parameter gradients, largest deviations (bf16 storage, fp32 run to run):", top)
    assert not bad, bad
    for k in ref[3]:
        a, b, c = ref[3][k].float(), got[3][k].float(), ref2[3][k].float()
        if float(a.norm()) > 1e-2 * a.numel() ** 0.5:
            assert _l2(a, b) <= 3 * _l2(a, c) + 3e-2, k
        else:       # a running mean of (nearly) centred activations: compare on the scale of a unit-variance channel
            assert float((a - b).abs().max()) <= 3 * float((a - c).abs().max()) + 1e-2, k
