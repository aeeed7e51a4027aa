"""-m gpu: the kernels that carry the benchmark, against the CPU oracle ON THE BENCHMARK'S OWN MAPS.

`bench.py` times BASELINE.json configs[1]: 4 synthetic S50k scenes per GPU.  At that size a persistent workgroup of
`k_spconv_tile` walks 5-6 units (1 200-1 500 tiles on 256 CUs) and `k_spconv_pairs_wgrad_rows16` runs thousands of
segments -- regimes the small-map parity cases never enter.  Every case below voxelises the benchmark batch itself,
strides it the way BiResNet does (biresnet.py:358-406: tensor strides 2, 4, 8, 16) and runs the layer's forward, data
gradient and weight gradient through the public autograd op in the bench precision (bf16 operands gathered from bf16 row
copies, fp32 accumulate) on the device and on the oracle's bit-level emulation of the same rounding, so only the fp32
summation order differs: rtol 1e-4, atol 1e-5 x the magnitude of the sums.
The oracle finishes one such layer in seconds on the host cores."""
import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, build_model, me

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5
_BATCH = {}


def bench_coords(cfg="S50k", scenes=4, voxel=0.02):
    """Voxel coordinates of a benchmark batch (default: S50k x 4, 0.02 m), as CAGroup3D.voxelization makes them."""
    k = (cfg, scenes, voxel)
    if k not in _BATCH:
        pts = build_model.synthetic_batch(cfg, scenes, device="cpu")["points"]
        c = pts[:, :4].clone()
        c[:, 1:] = torch.floor(c[:, 1:] / voxel)
        _BATCH[k] = c.int().contiguous()
    return _BATCH[k]


def _layer_case(coords, in_stride, ks, conv_stride, transpose, cin, cout, seed, calls=None):
    """One BiResNet-shaped layer on the map of tensor stride `in_stride`: y, dx, dw of SparseConvFunction."""
    dev = coords.device
    x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1, device=dev))
    mgr = x.coordinate_manager
    keys = {1: x.coordinate_map_key}
    s = 1
    while s < max(in_stride, in_stride * conv_stride):
        keys[s * 2] = mgr.stride(keys[s], 2)
        s *= 2
    in_key = keys[in_stride]
    if transpose:
        out_key = keys[in_stride // conv_stride]
    else:
        out_key = keys[in_stride * conv_stride]
    km = mgr.kernel_map(in_key, out_key, ks, 1, transpose)
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(km.n_in, cin, generator=g).to(dev)
    w = (torch.randn(km.K, cin, cout, generator=g) / (cin * min(km.K, 27)) ** 0.5).to(dev)
    dy = torch.randn(km.n_out, cout, generator=g).to(dev)
    xf = feats.requires_grad_(True)
    wp = w.requires_grad_(True)
    y = me.SparseConvFunction.apply(xf, wp, None, km)
    (y * dy).sum().backward()
    return y.detach(), xf.grad, wp.grad, (km.n_in, km.n_out, int((km.nbr >= 0).sum()))


# (name, tensor stride of the input map, kernel, conv stride, transposed, cin, cout, expected rows in)
LAYERS = [
    ("layer1 64->64 @2", 2, 3, 1, False, 64, 64),
    ("layer3_ 128->128 @4", 4, 3, 1, False, 128, 128),
    ("layer3 256->256 @8", 8, 3, 1, False, 256, 256),
    ("layer4 512->512 @16", 16, 3, 1, False, 512, 512),
    ("down3 128->256 @4->8", 4, 3, 2, False, 128, 256),
    ("layer2 first conv 64->128 @2->4", 2, 3, 2, False, 64, 128),
    ("out convT k2 256->256 @4->2", 4, 2, 2, True, 256, 256),
]


@pytest.mark.parametrize("name,in_stride,ks,cstride,transpose,cin,cout", LAYERS, ids=[l[0] for l in LAYERS])
def test_benchmark_layer_matches_oracle(oracle, hip, monkeypatch, name, in_stride, ks, cstride, transpose, cin, cout):
    _check_layer(oracle, hip, monkeypatch, bench_coords(), name, in_stride, ks, cstride, transpose, cin, cout, True)


# BASELINE.json configs[3] (SUN RGB-D-shaped: 8 single-view 100 k-point scenes, 3 votes: the 64 -> 192 `feature_offset`
# convolution, cagroup_head.py:170-172) and configs[4] (4 x 200 k points at 0.01 m: 2 500+ tiles per launch, five rounds of
# tile-kernel units and more): the layers that carry those runs, on THEIR maps.
OTHER = [
    ("S100k-yaw x 8: feature_offset 64->192 @2", "S100k-yaw", 8, 0.02, 2, 3, 1, False, 64, 192, True),
    ("S100k-yaw x 8: layer3_ 128->128 @4", "S100k-yaw", 8, 0.02, 4, 3, 1, False, 128, 128, True),
    ("S100k-yaw x 8: down3 128->256 @4->8", "S100k-yaw", 8, 0.02, 4, 3, 2, False, 128, 256, True),
    ("S200k x 4 @0.01: layer1 64->64 @2", "S200k", 4, 0.01, 2, 3, 1, False, 64, 64, True),
    ("S200k x 4 @0.01: layer3_ 128->128 @4", "S200k", 4, 0.01, 4, 3, 1, False, 128, 128, True),
    ("S200k x 4 @0.01: out convT k2 256->256 @4->2", "S200k", 4, 0.01, 4, 2, 2, True, 256, 256, True),
]


@pytest.mark.parametrize("name,cfg,scenes,voxel,in_stride,ks,cstride,transpose,cin,cout,tile", OTHER, ids=[l[0] for l in OTHER])
def test_layers_of_the_other_configs_match_oracle(oracle, hip, monkeypatch, name, cfg, scenes, voxel, in_stride, ks, cstride,
                                                  transpose, cin, cout, tile):
    _check_layer(oracle, hip, monkeypatch, bench_coords(cfg, scenes, voxel), name, in_stride, ks, cstride, transpose, cin, cout, tile)


def _check_layer(oracle, hip, monkeypatch, coords, name, in_stride, ks, cstride, transpose, cin, cout, expect_tile):
    monkeypatch.setattr(me, "PRECISION", 1)
    tile_calls, wgrad_calls = [], []
    real_tile = me._conv_tile
    with _lib.use_library(oracle):
        ref = _layer_case(coords, in_stride, ks, cstride, transpose, cin, cout, seed=cin + cout + ks)
    with _lib.use_library(hip):
        monkeypatch.setattr(me, "_conv_tile", lambda *a, **k: (tile_calls.append((a[2].ntile, a[5])), real_tile(*a, **k))[1])
        real_call = hip.call

        def spy(name_, *a):
            if name_ == "cg3d_spconv_pairs_wgrad":
                wgrad_calls.append(int(a[-2].value))
            return real_call(name_, *a)
        monkeypatch.setattr(hip, "call", spy)
        out = _layer_case(coords.cuda(), in_stride, ks, cstride, transpose, cin, cout, seed=cin + cout + ks)
        torch.cuda.synchronize()
    assert ref[3] == out[3], "map sizes differ between oracle and device"
    n_in, n_out, P = out[3]
    # the device really took the benchmark's kernels: tile kernel forward + data gradient, bf16-row weight gradient
    if expect_tile:
        assert len(tile_calls) == 2, "forward and data gradient must run k_spconv_tile (got %d)" % len(tile_calls)
    assert wgrad_calls == [2], "the weight gradient must run the bf16-row kernel (precision 2)"
    if expect_tile and in_stride <= 4 and not transpose:
        units = max(t * max(c // 128, 1) for t, c in tile_calls)
        assert units > 2 * 256, "benchmark-size map: every persistent workgroup walks several units (%d units)" % units
    for nm, r, o in zip(("y", "dx", "dw"), ref[:3], out[:3]):
        scale = max(float(r.abs().max()), 1.0)
        try:
            torch.testing.assert_close(o.cpu(), r, rtol=RTOL, atol=ATOL * scale)
        except AssertionError as e:  # pragma: no cover
            raise AssertionError("%s: %s differs (n_in %d, n_out %d, pairs %d): %s" % (name, nm, n_in, n_out, P, e))


def _class_map_case(coords, B, G, ks, seed, vs):
    """The 9^3 class-branch convolution (cagroup_head.py:259) on a benchmark-shaped class map: the stride-2 voxels, every
    class c re-quantised at its own voxel size into batch index c*B + b of ONE coordinate map, grouped weights."""
    dev = coords.device
    # the class coordinates are made on the host for both sides (a float division by a scalar is a multiplication by its
    # reciprocal on the device: floor() of it can differ from the host's in the last voxel)
    ch = coords.cpu()
    c2 = torch.unique(torch.cat([ch[:, :1], torch.div(ch[:, 1:], 2, rounding_mode="floor") * 2], 1), dim=0)     # tensor stride 2
    xyz = c2[:, 1:].float() * 0.02
    rows = []
    for c in range(G):
        sel = (torch.div(c2[:, 1], 16, rounding_mode="floor") + 5 * torch.div(c2[:, 2], 16, rounding_mode="floor")) % G == c   # 32 cm columns
        q = torch.floor(xyz[sel] / vs[c])
        rows.append(torch.cat([(c * B + c2[sel, :1]).float(), q], 1))
    fine = torch.cat(rows).int().contiguous().to(dev)
    cls_map = me.SparseTensor(coordinates=fine, features=torch.zeros(fine.shape[0], 1, device=dev))
    key = cls_map.coordinate_map_key
    cm = cls_map.coordinate_manager
    km = cm.kernel_map(key, key, ks, 1, False)
    n = km.n_out
    per_class = torch.bincount(cls_map.C[:, 0].long() // B, minlength=G).cpu().numpy()
    bounds = (0,) + tuple(np.cumsum(per_class).tolist())
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(n, 64, generator=g).to(dev).requires_grad_(True)
    ws = [(torch.randn(ks ** 3, 64, 64, generator=g) * 0.02).to(dev).requires_grad_(True) for _ in range(G)]
    dy = torch.randn(n, 64, generator=g).to(dev)
    y = me.grouped_conv(feats, ws, km, bounds, closed=True)
    (y * dy).sum().backward()
    return y.detach(), feats.grad, torch.stack([w.grad for w in ws]), (n, int((km.nbr >= 0).sum()))


@pytest.mark.parametrize("ks", [9, 5])
def test_benchmark_class_map_convolution_matches_oracle(oracle, hip, monkeypatch, ks):
    coords = bench_coords()
    G, B = 18, 4
    vs = [0.08 + 0.02 * (c % 5) for c in range(G)] if ks == 9 else [0.30 + 0.05 * (c % 5) for c in range(G)]
    monkeypatch.setattr(me, "PRECISION", 1)
    tile_calls = []
    real_tile = me._conv_tile
    with _lib.use_library(hip):
        monkeypatch.setattr(me, "_conv_tile", lambda *a, **k: (tile_calls.append(a[2].K), real_tile(*a, **k))[1])
        out = _class_map_case(coords.cuda(), B, G, ks, seed=ks, vs=vs)
        torch.cuda.synchronize()
        monkeypatch.setattr(me, "_conv_tile", real_tile)
    with _lib.use_library(oracle):
        # (the grouped bf16 path is a device path; on the oracle the same operator runs as the stacked-weight conv)
        ref = _class_map_case(coords, B, G, ks, seed=ks, vs=vs)
    assert ref[3] == out[3]
    assert tile_calls == [ks ** 3, ks ** 3], "grouped forward and data gradient must run k_spconv_tile"
    for nm, r, o in zip(("y", "dx", "dw"), ref[:3], out[:3]):
        scale = max(float(r.abs().max()), 1.0)
        torch.testing.assert_close(o.cpu(), r, rtol=RTOL, atol=ATOL * scale, msg=lambda m: "%s (K=%d): %s" % (nm, ks ** 3, m))


# ------------------------------------------------------------------ the 1x1x1 convolutions (cg3d_linear_fwd) at the benchmark's row counts
@pytest.mark.parametrize("n,cin,cout", [(155773, 64, 64), (82107, 128, 256), (23015, 256, 128), (155773, 64, 128)])
def test_benchmark_linear_layer_matches_oracle(oracle, hip, n, cin, cout):
    """me.linear in the bench precision on the row counts of the S50k x 4 maps: forward (bf16 rows x fragment-ordered weights
    + bias), data gradient (the other weight copy), weight gradient (bf16-rows pair kernel on the identity list) and the
    BatchNorm statistics the kernel leaves in its epilogue, all against the oracle's restatement of the same arithmetic."""
    g = torch.Generator().manual_seed(n + cin)
    x, w, b = torch.randn(n, cin, generator=g), torch.randn(cin, cout, generator=g) / cin ** 0.5, torch.randn(cout, generator=g)
    dy = torch.randn(n, cout, generator=g)
    # operands that bf16 represents exactly: the oracle side's weight gradient is the fp32 product x^T dy
    x, w, dy = (t.to(torch.bfloat16).float() for t in (x, w, dy))

    def run(lib, dev):
        old = me.PRECISION, me.WANT_BN_STATS
        me.PRECISION, me.WANT_BN_STATS = 1, True
        calls = []
        orig = lib.call
        lib.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
        try:
            with _lib.use_library(lib):
                me.zero_arena().reset()
                xs, ws, bs = (t.to(dev).clone().requires_grad_(True) for t in (x, w, b))
                y = me.linear(xs, ws, bs)
                st = me._STATS.pop(y.data_ptr(), None)
                sums = st[0].view(me.BN_SLOTS, 2, cout).sum(0).cpu() if st is not None else None
                y.backward(dy.to(dev))
                return [y.detach().cpu(), xs.grad.cpu(), ws.grad.cpu(), bs.grad.cpu(), sums], calls
        finally:
            lib.call = orig
            me.PRECISION, me.WANT_BN_STATS = old
            me._STATS.clear()
    want, _ = run(oracle, "cpu")
    got, calls = run(hip, "cuda") if hip is not None else run(oracle, "cpu")      # (CG3D_PARITY_SELFTEST: the test code alone)
    assert calls.count("cg3d_linear_fwd") == 2, calls               # forward + dX on the own kernel
    assert hip is None or "cg3d_spconv_pairs_wgrad" in calls, calls
    for name, a, r in zip(("y", "dx", "dw", "db", "statistics"), got, want):
        assert a is not None and r is not None, name
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(a, r, rtol=1e-4, atol=1e-5 * scale * (30 if name in ("dw", "db", "statistics") else 1), msg=lambda m: name + ": " + m)
