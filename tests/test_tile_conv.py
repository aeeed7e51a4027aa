"""Tile plans (cg3d_tile_plan_build) and the LDS-staged sparse convolution built on them (cg3d_spconv_tile_fwd).

A plan's slot numbering is the implementation's choice, so a plan is checked by DECODING it back into the kernel map
it encodes (bit-exact) and through its invariants; the convolution is compared with the oracle's dense-map bf16
emulation (same RNE operand rounding, fp32 accumulate: only the summation order differs).
CPU (-m "not gpu"): the oracle's plan builder / plan-driven convolution.  -m gpu: the HIP kernels against them."""
import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, me
from util import rand_coords, surface_coords

RTOL, ATOL = 1e-4, 1e-5


def _kernel_map(coords, ks=3, stride=1):
    x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1, device=coords.device))
    mgr = x.coordinate_manager
    out_key = mgr.stride(x.coordinate_map_key, stride) if stride > 1 else x.coordinate_map_key
    return mgr.kernel_map(x.coordinate_map_key, out_key, ks, 1, False)


def decode_plan(plan):
    """TilePlan -> nbr int32 [K, n_out] (numpy), checking the plan's invariants on the way."""
    K, n_out = plan.K, plan.n_out
    slots = plan.slots.cpu().numpy().view(np.uint16)
    live = plan.live.cpu().numpy()
    ptab, npass, ulist, cursor = plan.pass_tab.cpu().numpy(), plan.npass.cpu().numpy(), plan.ulist.cpu().numpy(), plan.cursor.cpu().numpy()
    tiles = plan.tiles.cpu().numpy() if plan.tiles is not None else None
    order = plan.order.cpu().numpy() if getattr(plan, "order", None) is not None else None
    if order is not None:
        assert sorted(order.tolist()) == list(range(n_out)), "order must be a permutation of the output rows"
    assert cursor[1] == 0, "plan overflowed"
    nbr = np.full((K, n_out), -1, np.int32)
    used = 0
    for t in range(plan.ntile):
        row0, rows = (tiles[t, 1], tiles[t, 2]) if tiles is not None else (t * 128, min(128, n_out - t * 128))
        k_next = 0
        for p in range(npass[t]):
            k0, k1, uoff, ucnt = ptab[t, p]
            assert k0 == k_next and k1 > k0 or (k0 == k1 == K == 0), "passes must partition the offsets in order"
            assert 0 <= ucnt <= plan.ucap
            k_next = k1
            used += ucnt
            ul = ulist[uoff:uoff + ucnt]
            assert len(np.unique(ul)) == ucnt, "a pass stages every input row once"
            s = slots[t, k0:k1, :]
            assert s.max(initial=0) <= ucnt
            assert (s[:, rows:] == 0).all()
            dec = np.where(s > 0, ul[np.maximum(s.astype(np.int64) - 1, 0)] if ucnt else -1, -1).astype(np.int32)
            out_rows = np.arange(row0, row0 + rows) if order is None else order[row0:row0 + rows]      # position -> output row
            nbr[k0:k1, out_rows] = dec[:, :rows]
            lv = np.stack([(s[:, m * 32:(m + 1) * 32] > 0).any(1) for m in range(4)], 1)
            lv_plan = (live[t, k0:k1, None] >> np.arange(4)[None]) & 1
            assert (lv == lv_plan.astype(bool)).all(), "liveness bits"
        assert k_next == K
    assert used == cursor[0]
    return nbr


def _plan_roundtrip(nbr, P, ucap, tiles=None, sort_rows=False):
    plan = me.build_tile_plan(nbr, P, tiles=tiles, ucap=ucap, sort_rows=sort_rows)
    dec = decode_plan(plan)
    return plan, dec


@pytest.mark.parametrize("n,ks,stride,ucap", [(3000, 3, 1, 511), (3000, 3, 1, 128), (2500, 3, 2, 200), (900, 5, 1, 300), (130, 3, 1, 511),
                                             (1, 3, 1, 511)])
def test_oracle_plan_decodes_to_kernel_map(oracle, n, ks, stride, ucap):
    with _lib.use_library(oracle):
        km = _kernel_map(surface_coords(n, batch=2, extent=max(6, int(n ** 0.5) // 3), seed=n + ks), ks, stride)
        P = int((km.nbr >= 0).sum())
        plan, dec = _plan_roundtrip(km.nbr.contiguous(), P, ucap)
        assert np.array_equal(dec, km.nbr.numpy())
        if ucap == 128 and n >= 1000:
            assert int(plan.npass.max()) > 1, "a 128-row LDS must force several passes on a 3^3 map"


def test_oracle_plan_empty_map_and_grouped_tiles(oracle):
    with _lib.use_library(oracle):
        nbr = torch.full((27, 0), -1, dtype=torch.int32)
        plan = me.build_tile_plan(nbr, 0)
        assert plan.ntile == 0 and decode_plan(plan).shape == (27, 0)
        km = _kernel_map(surface_coords(2000, batch=2, extent=14, seed=4))
        n = km.n_out
        bounds = (0, n // 3, n // 3 + 5, n)                     # three row groups, one of them tiny
        tiles = km.tiles(bounds)
        plan, dec = _plan_roundtrip(km.nbr.contiguous(), int((km.nbr >= 0).sum()), 511, tiles)
        assert np.array_equal(dec, km.nbr.numpy())


def _row_order_reference(nbr, window=1024):
    """numpy restatement of cg3d_tile_row_order: inside every window rows sorted by (set of live offsets, row)."""
    nbr = np.asarray(nbr)
    K, n = nbr.shape
    sig = ((nbr >= 0).astype(np.uint64) << np.arange(K, dtype=np.uint64)[:, None]).sum(0)
    order = np.empty(n, np.int32)
    for w0 in range(0, n, window):
        idx = np.arange(w0, min(n, w0 + window))
        order[idx] = idx[np.lexsort((idx, sig[idx]))]
    return order


@pytest.mark.parametrize("n,ks,stride,transposed", [(5000, 3, 2, True), (3000, 2, 2, True), (2600, 3, 1, False), (1, 3, 2, True)])
def test_oracle_sorted_rows_plan_is_the_same_map_with_fewer_live_offsets(oracle, n, ks, stride, transposed):
    """cg3d_tile_row_order + a plan on the permuted rows: decodes to the same kernel map; on the transposed map of a strided
    convolution the tiles have far fewer live (tile, offset) entries than in arrival order."""
    with _lib.use_library(oracle):
        km = _kernel_map(surface_coords(n, batch=2, extent=max(6, int(n ** 0.5) // 3), seed=n + ks), ks, stride)
        nbr = (km.nbrT if transposed else km.nbr).contiguous()
        P = int((nbr >= 0).sum())
        plain, dec0 = _plan_roundtrip(nbr, P, 511)
        plan, dec = _plan_roundtrip(nbr, P, 511, sort_rows=True)
        assert plan.order is not None and np.array_equal(plan.order.numpy(), _row_order_reference(nbr.numpy()))
        assert np.array_equal(dec, nbr.numpy()) and np.array_equal(dec0, nbr.numpy())
        live_sorted, live_plain = int((plan.live.numpy() != 0).sum()), int((plain.live.numpy() != 0).sum())
        if stride == 2 and n >= 2000:
            assert live_sorted < 0.6 * live_plain, (live_sorted, live_plain)
        # the policy: windows of 1024 rows on sparse non-symmetric maps, arrival order everywhere else
        pol = km.tile_plan(transposed)
        if stride == 2 and P < 0.2 * nbr.shape[0] * nbr.shape[1]:
            assert pol.order is not None and np.array_equal(pol.order.numpy(), _row_order_reference(nbr.numpy(), 1024))
        else:
            assert pol.order is None
        assert np.array_equal(decode_plan(pol), nbr.numpy())
        # window 128: every tile keeps its rows (a permutation inside the tile)
        p128, dec128 = _plan_roundtrip(nbr, P, 511, sort_rows=128)
        assert np.array_equal(p128.order.numpy(), _row_order_reference(nbr.numpy(), 128)) and np.array_equal(dec128, nbr.numpy())
        assert (p128.order.numpy() // 128 == np.arange(p128.n_out) // 128).all()


def _tile_conv(coords, x, w, bias, ks, stride, ucap, ksplit, transposed):
    """conv through a plan; transposed: the data-gradient problem (dY rows -> dX rows, W^T)."""
    km = _kernel_map(coords, ks, stride)
    nbr = (km.nbrT if transposed else km.nbr).contiguous()
    n_in = km.n_out if transposed else km.n_in
    P = int((nbr >= 0).sum())
    plan = me.build_tile_plan(nbr, P, ucap=ucap)
    wt, wp = me._prep_frag(w, want_t=not transposed, want_p=transposed)
    cin, cout = (w.shape[2], w.shape[1]) if transposed else (w.shape[1], w.shape[2])
    x16 = me._to_bf16(x[:n_in].contiguous())
    y = me._conv_tile(x16, wp if transposed else wt, plan, bias, cin, cout, n_in, P, ksplit)
    # the same operator on the dense map (bf16 rows, classic weight layout)
    wb = me._prep_bf16_both(w)[1 if transposed else 0]
    yref = me._conv_implicit_bf16(x16, wb, nbr, bias, nbr.shape[1], cin, cout, P)
    return y, yref


@pytest.mark.parametrize("cin,cout,n,ks,stride", [(64, 64, 2500, 3, 1), (64, 128, 2000, 3, 2), (128, 128, 1500, 3, 1), (64, 192, 1200, 3, 1)])
def test_oracle_plan_conv_equals_dense_map_conv(oracle, cin, cout, n, ks, stride):
    """CPU: the oracle's plan-driven convolution reads neighbours only through the plan == its dense-map convolution."""
    torch.manual_seed(cin + cout)
    coords = surface_coords(n, batch=2, extent=max(8, int(n ** 0.5) // 3), seed=n)
    x = torch.randn(coords.shape[0], cin)
    w = torch.randn(ks ** 3, cin, cout) / (cin * 27) ** 0.5
    with _lib.use_library(oracle):
        for transposed in (False, True):
            xx = torch.randn(coords.shape[0], cout) if transposed else x
            y, yref = _tile_conv(coords, xx, w, None if transposed else torch.randn(cout), ks, stride, 200, 1, transposed)
            torch.testing.assert_close(y, yref, rtol=RTOL, atol=ATOL * max(float(yref.abs().max()), 1.0))


def test_map_onto_itself_transposes_by_flipping_the_offsets(oracle):
    """nbrT[k] == nbr[K-1-k] for a centred odd kernel on one coordinate map: the manager answers `nbrT` with the flipped
    map (no second hash lookup); checked against the transposed map found by lookup."""
    with _lib.use_library(oracle):
        for ks in (3, 5):
            coords = surface_coords(1500, batch=2, extent=10, seed=ks)
            x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1))
            mgr = x.coordinate_manager
            km = mgr.kernel_map(x.coordinate_map_key, x.coordinate_map_key, ks, 1, False)
            assert km.symmetric
            src = mgr.get(x.coordinate_map_key)
            offs = me._offsets(ks, 1, coords.device)
            by_lookup = me.CoordinateManager._lookup_map(src.coords, src, (-offs).contiguous())      # i - off = o
            assert torch.equal(km.nbrT, by_lookup)
        km2 = _kernel_map(surface_coords(1500, batch=2, extent=10, seed=1), 3, 2)
        assert not km2.symmetric


def _transposed_by_lookup(coords, ks, stride):
    x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1, device=coords.device))
    mgr = x.coordinate_manager
    out_key = mgr.stride(x.coordinate_map_key, stride)
    km = mgr.kernel_map(x.coordinate_map_key, out_key, ks, 1, False)
    src, dst = mgr.get(x.coordinate_map_key), mgr.get(out_key)
    offs = me._offsets(ks, src.tensor_stride, coords.device)
    by_lookup = me.CoordinateManager._lookup_map(src.coords, dst, (-offs).contiguous())          # i - off = o
    return km, by_lookup


def test_oracle_transposed_map_by_scatter_equals_lookup(oracle):
    """cg3d_kernel_map_transpose (nbrT scattered from nbr) == the transposed map found through the hash table, strided maps"""
    with _lib.use_library(oracle):
        for ks, stride, n in ((3, 2, 1500), (2, 2, 900), (5, 2, 700)):
            km, by_lookup = _transposed_by_lookup(surface_coords(n, batch=2, extent=9, seed=ks + n), ks, stride)
            assert not km.symmetric and torch.equal(km.nbrT, by_lookup)


@pytest.mark.parametrize("ks,cin,cout", [(3, 64, 64), (5, 64, 128)])
def test_oracle_data_gradient_on_the_forward_plan_with_reversed_weights(oracle, ks, cin, cout):
    """wrev: conv(dY, W^T) through the plan of the transposed map == the same call on the FORWARD plan with weight slot
    K-1-k for offset k (grouped: reversed within every group)."""
    torch.manual_seed(ks)
    with _lib.use_library(oracle):
        y_t, y_rev = _wrev_case(surface_coords(1200, batch=2, extent=9, seed=ks), ks, cin, cout)
    torch.testing.assert_close(y_rev, y_t, rtol=RTOL, atol=ATOL * max(float(y_t.abs().max()), 1.0))


def _wrev_case(coords, ks, cin, cout, G=2):
    km = _kernel_map(coords, ks)
    n = km.n_out
    bounds = (0, n // 3, n)
    K = ks ** 3
    w = (torch.randn(G * K, cin, cout) / (cin * 27) ** 0.5).to(coords.device)
    dy = torch.randn(n, cout).to(coords.device)
    P = int((km.nbr >= 0).sum())
    tiles = km.tiles(bounds)
    _, wp = me._prep_frag(w, False, True)
    dy16 = me._to_bf16(dy)
    plan_t = me.build_tile_plan(km.nbrT.contiguous(), P, tiles=tiles, ucap=300)
    plan_f = me.build_tile_plan(km.nbr.contiguous(), P, tiles=tiles, ucap=300)
    y_t = me._conv_tile(dy16, wp, plan_t, None, cout, cin, n, P)
    y_rev = me._conv_tile(dy16, wp, plan_f, None, cout, cin, n, P, wrev=True)
    return y_t.cpu(), y_rev.cpu()


def test_fragment_order_is_a_permutation_of_the_classic_layout(oracle):
    with _lib.use_library(oracle):
        w = torch.randn(5, 64, 96)
        wt, wp = me._prep_frag(w, True, False)
        classic_t, classic_p = me._prep_bf16_both(w)
        assert sorted(wt.view(5, -1)[2].tolist()) == sorted(classic_t.view(5, -1)[2].tolist())
        # element (co, ci) sits at [co/32][ci/16][(ci/8 & 1)*32 + co%32][ci%8]
        f = wt.view(5, 96 // 32, 64 // 16, 64, 8)
        for co, ci in ((0, 0), (33, 17), (95, 63), (40, 8)):
            assert f[3, co // 32, ci // 16, ((ci // 8) & 1) * 32 + co % 32, ci % 8] == classic_t[3, co, ci]
        w2 = torch.randn(2, 64, 32)
        _, wp2 = me._prep_frag(w2, False, True)
        cp = me._prep_bf16_both(w2)[1]
        g = wp2.view(2, 64 // 32, 32 // 16, 64, 8)
        for ci, co in ((0, 0), (33, 17), (63, 31)):
            assert g[1, ci // 32, co // 16, ((co // 8) & 1) * 32 + ci % 32, co % 8] == cp[1, ci, co]


def _conv_then_bn(coords, cin, cout, dev, fused):
    """tile conv -> training-mode BatchNorm through the public functions; returns y, (mean, var), running stats."""
    km = _kernel_map(coords, 3, 1)
    P = int((km.nbr >= 0).sum())
    g = torch.Generator().manual_seed(cin + cout)
    x16 = me._to_bf16(torch.randn(km.n_in, cin, generator=g).to(dev))
    w = (torch.randn(27, cin, cout, generator=g) / (cin * 27) ** 0.5).to(dev)
    plan = me.build_tile_plan(km.nbr, P)
    wf, _ = me._prep_frag(w, True, False)
    old = me.FUSED_BN_STATS
    me.FUSED_BN_STATS = fused
    me._STATS.clear()
    try:
        y = me._conv_tile(x16, wf, plan, None, cin, cout, km.n_in, P, want_stats=True)
        assert (y.data_ptr() in me._STATS) == bool(fused)
        bn = torch.nn.BatchNorm1d(cout).to(dev).train()
        out = me.fused_bn_act(y, [bn], None, me.ACT_RELU)
        assert y.data_ptr() not in me._STATS, "the BatchNorm consumes the partials"
    finally:
        me.FUSED_BN_STATS = old
        me._STATS.clear()
    return y, out, bn.running_mean.clone(), bn.running_var.clone()


def test_oracle_bn_statistics_from_the_conv_epilogue(oracle):
    """cg3d_spconv_tile_fwd(stats=...) fills the same statistics table as cg3d_bn_sums over the stored rows."""
    with _lib.use_library(oracle):
        coords = surface_coords(1500, batch=2, extent=10, seed=5)
        y1, o1, rm1, rv1 = _conv_then_bn(coords, 64, 128, "cpu", True)
        y0, o0, rm0, rv0 = _conv_then_bn(coords, 64, 128, "cpu", False)
    torch.testing.assert_close(y1, y0)
    torch.testing.assert_close(rm1, rm0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rv1, rv0, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(o1, o0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm1, 0.1 * y1.mean(0), rtol=1e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("ks,stride,n", [(3, 2, 20000), (2, 2, 9000), (5, 2, 3000), (3, 2, 1)])
def test_hip_transposed_map_by_scatter_equals_lookup(hip, ks, stride, n):
    with _lib.use_library(hip):
        km, by_lookup = _transposed_by_lookup(surface_coords(n, batch=3, extent=max(6, int(n ** 0.5) // 3), seed=ks + n).cuda(), ks, stride)
        assert torch.equal(km.nbrT, by_lookup)


@pytest.mark.gpu
@pytest.mark.parametrize("ks,dil,n", [(3, 1, 20000), (5, 1, 6000), (3, 2, 8000), (9, 1, 1500), (3, 1, 1)])
def test_hip_self_map_by_half_the_lookups_is_bit_identical(hip, ks, dil, n):
    """cg3d_kernel_map_self (K/2 hash lookups + mirrored writes) == cg3d_kernel_map on the map of a coordinate map onto itself"""
    with _lib.use_library(hip):
        coords = surface_coords(n, batch=3, extent=max(6, int(n ** 0.5) // 3), seed=n + ks).cuda()
        x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1, device="cuda"))
        src = x.coordinate_manager.get(x.coordinate_map_key)
        offs = me._offsets(ks, dil, coords.device)
        full = me.CoordinateManager._lookup_map(src.coords, src, offs, False)
        half = me.CoordinateManager._lookup_map(src.coords, src, offs, True)
        assert torch.equal(full, half)
        assert n < 100 or int((full >= 0).sum()) > full.shape[1]          # (more than the centre column)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,n", [(64, 64, 9000), (128, 128, 45000), (128, 256, 30000), (256, 512, 3000), (64, 192, 12000)])
def test_hip_bn_statistics_from_the_conv_epilogue(hip, cin, cout, n):
    """The loader waves of the tile kernel sum every output channel while they store a tile (one to three tiles per
    workgroup, 1-4 channel blocks): the BatchNorm that follows gets the same mean / variance / running statistics / output
    as from its own statistics pass."""
    with _lib.use_library(hip):
        coords = surface_coords(n, batch=4, extent=max(8, int(n ** 0.5) // 3), seed=n).cuda()
        y1, o1, rm1, rv1 = _conv_then_bn(coords, cin, cout, "cuda", True)
        y0, o0, rm0, rv0 = _conv_then_bn(coords, cin, cout, "cuda", False)
    torch.testing.assert_close(y1, y0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(rm1, rm0, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(rv1, rv0, rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(o1, o0, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(rm1, 0.1 * y1.mean(0), rtol=1e-3, atol=1e-5)

@pytest.mark.gpu
@pytest.mark.parametrize("n,ks,stride,ucap", [(9000, 3, 1, 511), (9000, 3, 1, 128), (6000, 3, 2, 255), (1500, 5, 1, 300), (700, 9, 1, 511),
                                             (130, 3, 1, 511), (1, 3, 1, 511)])
def test_hip_plan_decodes_to_kernel_map(hip, n, ks, stride, ucap):
    with _lib.use_library(hip):
        km = _kernel_map(surface_coords(n, batch=2, extent=max(6, int(n ** 0.5) // 3), seed=n + ks).cuda(), ks, stride)
        P = int((km.nbr >= 0).sum())
        for nbr in (km.nbr, km.nbrT):
            plan, dec = _plan_roundtrip(nbr.contiguous(), P, ucap)
            assert np.array_equal(dec, nbr.cpu().numpy())


@pytest.mark.gpu
def test_hip_plan_grouped_tiles_and_empty(hip):
    with _lib.use_library(hip):
        plan = me.build_tile_plan(torch.full((27, 0), -1, dtype=torch.int32, device="cuda"), 0)
        assert plan.ntile == 0
        km = _kernel_map(surface_coords(5000, batch=2, extent=20, seed=4).cuda())
        n = km.n_out
        bounds = (0, n // 3, n // 3 + 5, n)
        plan, dec = _plan_roundtrip(km.nbr.contiguous(), int((km.nbr >= 0).sum()), 511, km.tiles(bounds))
        assert np.array_equal(dec, km.nbr.cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,n,ks,stride,ucap,ksplit", [
    (64, 64, 9000, 3, 1, 511, 1), (64, 128, 6000, 3, 2, 511, 1), (128, 128, 4000, 3, 1, 511, 1), (128, 128, 4000, 3, 1, 128, 1),
    (256, 256, 1500, 3, 1, 511, 1), (256, 256, 1500, 3, 1, 511, 3), (128, 256, 3000, 3, 2, 300, 2), (512, 512, 600, 3, 1, 511, 4),
    (64, 64, 700, 9, 1, 511, 1), (64, 128, 1500, 5, 1, 200, 1), (128, 64, 3000, 3, 1, 511, 1), (64, 64, 100, 3, 1, 511, 1),
    (64, 192, 5000, 3, 1, 511, 1), (192, 64, 5000, 3, 1, 511, 1), (128, 320, 2000, 3, 1, 511, 2),
])
def test_hip_tile_conv_matches_oracle(oracle, hip, cin, cout, n, ks, stride, ucap, ksplit):
    """forward and data-gradient problem through the HIP plan + LDS-staged kernel == the oracle's dense-map bf16
    emulation (and == the HIP dense-map kernel it replaces)."""
    torch.manual_seed(cin * 7 + cout + ks)
    coords = surface_coords(n, batch=2, extent=max(8, int(n ** 0.5) // 3), seed=n + ks)
    w = torch.randn(ks ** 3, cin, cout) / (cin * min(ks, 3) ** 3) ** 0.5
    for transposed in (False, True):
        x = torch.randn(coords.shape[0], cout if transposed else cin)
        bias = None if transposed else torch.randn(cout)
        with _lib.use_library(oracle):
            _, ref = _tile_conv(coords, x, w, bias, ks, stride, ucap, 1, transposed)
        with _lib.use_library(hip):
            y, ydense = _tile_conv(coords.cuda(), x.cuda(), w.cuda(), bias.cuda() if bias is not None else None, ks, stride, ucap,
                                   ksplit, transposed)
        scale = max(float(ref.abs().max()), 1.0)
        torch.testing.assert_close(y.cpu(), ref, rtol=RTOL, atol=ATOL * scale)
        torch.testing.assert_close(ydense.cpu(), ref, rtol=RTOL, atol=ATOL * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("n,ks,stride,cin,cout", [(30000, 3, 2, 128, 64), (20000, 2, 2, 64, 128), (9000, 3, 2, 256, 128), (700, 3, 2, 64, 64)])
def test_hip_sorted_rows_plan_and_convolution(oracle, hip, n, ks, stride, cin, cout):
    """Device row order == the oracle's (bit-exact); the plan on the permuted rows decodes to the map; the convolution
    through it (rows stored at order[position]) == the oracle's dense-map bf16 emulation == the unpermuted plan."""
    coords = surface_coords(n, batch=3, extent=max(6, int(n ** 0.5) // 3), seed=n + ks)
    torch.manual_seed(n)
    w = torch.randn(ks ** 3, cin, cout) / (cin * 8) ** 0.5          # the data-gradient problem: dY [n_out, cout] -> dX [n_in, cin]
    with _lib.use_library(oracle):
        km0 = _kernel_map(coords, ks, stride)
        ref_order = me.build_tile_plan(km0.nbrT.contiguous(), int((km0.nbrT >= 0).sum()), sort_rows=True).order
        dy = torch.randn(km0.n_out, cout)
        ref = me._conv_implicit_bf16(me._to_bf16(dy), me._prep_bf16_both(w)[1], km0.nbrT.contiguous(), None, km0.n_in, cout, cin,
                                     int((km0.nbrT >= 0).sum()))
    with _lib.use_library(hip):
        km = _kernel_map(coords.cuda(), ks, stride)
        nbrT = km.nbrT.contiguous()
        P = int((nbrT >= 0).sum())
        plan, dec = _plan_roundtrip(nbrT, P, 511, sort_rows=True)
        assert torch.equal(plan.order.cpu(), ref_order)
        assert np.array_equal(dec, nbrT.cpu().numpy())
        plain = me.build_tile_plan(nbrT, P)
        assert int((plan.live != 0).sum()) < int((plain.live != 0).sum())
        _, wp = me._prep_frag(w.cuda(), False, True)
        dy16 = me._to_bf16(dy.cuda())
        y = me._conv_tile(dy16, wp, plan, None, cout, cin, km.n_out, P)
        y0 = me._conv_tile(dy16, wp, plain, None, cout, cin, km.n_out, P)
        assert km.tile_plan(True).order is not None, "the autograd path must pick the sorted plan for this map"
    scale = max(float(ref.abs().max()), 1.0)
    torch.testing.assert_close(y.cpu(), ref, rtol=RTOL, atol=ATOL * scale)
    torch.testing.assert_close(y0.cpu(), ref, rtol=RTOL, atol=ATOL * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("ks,cin,cout", [(3, 64, 64), (3, 128, 128), (5, 64, 64), (9, 64, 64)])
def test_hip_data_gradient_on_the_forward_plan_with_reversed_weights(oracle, hip, ks, cin, cout):
    """the device kernel's wrev path (and, for K > 32 at 64 channels, its rows-staged-once-per-pass path) == the oracle"""
    coords = surface_coords(1200 if ks == 9 else 4000, batch=2, extent=9 if ks == 9 else 16, seed=ks)
    torch.manual_seed(ks)
    with _lib.use_library(oracle):
        ref_t, ref_rev = _wrev_case(coords, ks, cin, cout)
    torch.manual_seed(ks)
    with _lib.use_library(hip):
        y_t, y_rev = _wrev_case(coords.cuda(), ks, cin, cout)
    scale = max(float(ref_t.abs().max()), 1.0)
    torch.testing.assert_close(y_t, ref_t, rtol=RTOL, atol=ATOL * scale)
    torch.testing.assert_close(y_rev, ref_t, rtol=RTOL, atol=ATOL * scale)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,n", [(64, 64, 45000), (128, 128, 45000), (256, 256, 22000), (64, 128, 70000)])
def test_hip_tile_conv_with_several_units_per_workgroup(hip, cin, cout, n):
    """More (tile, channel block) units than CUs: every persistent workgroup walks 2-3 units -- the loader drains the
    previous unit's output tile from the LDS buffer it is about to refill.  Against the dense-map kernel on the device
    (the oracle comparison of both kernels is above, at sizes the CPU finishes in seconds)."""
    torch.manual_seed(n)
    with _lib.use_library(hip):
        coords = surface_coords(n, batch=4, extent=max(8, int(n ** 0.5) // 3), seed=n).cuda()
        km = _kernel_map(coords, 3, 1)
        assert -(-km.n_out // 128) * max(cout // 128, 1) > 300, "needs more units than the chip has CUs"
        P = int((km.nbr >= 0).sum())
        x16 = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
        w = torch.randn(27, cin, cout, device="cuda") / (cin * 27) ** 0.5
        bias = torch.randn(cout, device="cuda")
        plan = me.build_tile_plan(km.nbr, P)
        wf, _ = me._prep_frag(w, True, False)
        y = me._conv_tile(x16, wf, plan, bias, cin, cout, km.n_in, P)
        yref = me._conv_implicit_bf16(x16, me._prep_bf16_t(w), km.nbr, bias, km.n_out, cin, cout, P)
        torch.testing.assert_close(y, yref, rtol=RTOL, atol=ATOL * max(float(yref.abs().max()), 1.0))


@pytest.mark.gpu
def test_hip_tile_conv_grouped_rows_use_their_group_weights(oracle, hip):
    torch.manual_seed(3)
    coords = surface_coords(5000, batch=2, extent=20, seed=11)
    cin = cout = 64
    G = 3
    w = torch.randn(G * 27, cin, cout) / (cin * 27) ** 0.5
    x = torch.randn(coords.shape[0], cin)

    def run(c, xx, ww):
        km = _kernel_map(c)
        n = km.n_out
        bounds = (0, n // 3, n // 3 + 70, n)
        tiles = km.tiles(bounds)
        P = int((km.nbr >= 0).sum())
        plan = me.build_tile_plan(km.nbr.contiguous(), P, tiles=tiles)
        wt, _ = me._prep_frag(ww, True, False)
        x16 = me._to_bf16(xx[:km.n_in].contiguous())
        y = me._conv_tile(x16, wt, plan, None, cin, cout, km.n_in, P)
        yref = me._conv_implicit_bf16(x16, me._prep_bf16_t(ww), km.nbr, None, km.n_out, cin, cout, P, tiles)
        return y, yref
    with _lib.use_library(oracle):
        _, ref = run(coords, x, w)
    with _lib.use_library(hip):
        y, _ = run(coords.cuda(), x.cuda(), w.cuda())
    torch.testing.assert_close(y.cpu(), ref, rtol=RTOL, atol=ATOL * max(float(ref.abs().max()), 1.0))


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,n", [(128, 128, 6000), (256, 128, 3000), (128, 256, 3000)])
def test_autograd_conv_takes_the_tile_path_and_matches_oracle(oracle, hip, monkeypatch, cin, cout, n):
    """SparseConvFunction in the bf16 mode on a same-map 3^3 convolution: forward and data gradient through the tile kernel
    (fragment-ordered weights from _prep_frag), weight gradient unchanged == the oracle's bf16 emulation."""
    from test_hip_parity import _conv_case, both, close
    torch.manual_seed(cin + cout)
    coords = surface_coords(n, batch=2, extent=max(8, int(n ** 0.5) // 3), seed=n)
    feats = torch.randn(coords.shape[0], cin)
    w = torch.randn(27, cin, cout) / (cin * 27) ** 0.5
    bias = torch.randn(cout)
    dy = torch.randn(coords.shape[0], cout)
    monkeypatch.setattr(me, "TILE_MIN_ROWS", 0)
    monkeypatch.setattr(me, "PRECISION", 1)
    calls = []
    real = me._conv_tile
    monkeypatch.setattr(me, "_conv_tile", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    ref, out = both(oracle, hip, _conv_case, coords, feats, w, bias, dy, 3, 1)
    assert len(calls) == 2, "forward and data gradient must run the tile kernel on the device"
    for r, o in zip(ref, out):
        close(r, o, float(r.abs().max()))
