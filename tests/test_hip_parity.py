"""-m gpu: the hand-written HIP kernels against the CPU oracle on the same seeded inputs, called
through the same C-ABI binding.  Integer / index / mask outputs must be BIT-EXACT; floating-point
feature outputs are compared with the tolerance SURVEY.md section 8(d) states:
rtol 1e-4, atol 1e-5 (scaled by the magnitude of the accumulated sum).
"""
import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, me
from cagroup3d_amd.ops import iou3d_nms_utils, knn as knn_mod, rotated_iou
from util import rand_boxes, rand_coords, surface_coords

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-5


def both(oracle, hip, fn, *args):
    """Run fn(*args) once on the oracle (CPU tensors) and once on the HIP library (cuda tensors)."""
    def mv(x, dev):
        return x.to(dev) if torch.is_tensor(x) else x
    with _lib.use_library(oracle):
        ref = fn(*[mv(a, "cpu") for a in args])
    if hip is None:  # CG3D_PARITY_SELFTEST=1 (no GPU): oracle against itself, checks the test code only
        with _lib.use_library(oracle):
            return ref, fn(*[mv(a, "cpu") for a in args])
    with _lib.use_library(hip):
        out = fn(*[mv(a, "cuda") for a in args])
    torch.cuda.synchronize()
    return ref, out


def eq(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.equal(a.cpu(), b.cpu())


def close(ref, out, scale=1.0):
    torch.testing.assert_close(out.cpu(), ref.cpu(), rtol=RTOL, atol=ATOL * max(scale, 1.0))


# ------------------------------------------------------------------ coordinate maps
@pytest.mark.parametrize("n,qs", [(0, 1), (1, 1), (63, 1), (5000, 1), (5000, 2), (70000, 4), (300000, 1)])
def test_coord_map_build_bit_exact(oracle, hip, n, qs):
    coords = rand_coords(n, batch=3, extent=60, seed=n + qs)

    def fn(c):
        out, keys, vals, cap, uniq, inv = me._build_map(c, qs)
        return out, uniq, inv
    ref, out = both(oracle, hip, fn, coords)
    for r, o in zip(ref, out):
        eq(r, o)


def _maps(coords, kernel_size, stride, transpose=False, given=None):
    x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1, device=coords.device))
    mgr = x.coordinate_manager
    if given is not None:
        out_key, _, _ = mgr.insert(given, 1)
    elif transpose:
        fine = x.coordinate_map_key
        coarse = mgr.stride(fine, stride)
        km = mgr.kernel_map(coarse, fine, kernel_size, 1, True)
        return km.nbr, km.nbrT
    else:
        out_key = mgr.stride(x.coordinate_map_key, stride) if stride > 1 else x.coordinate_map_key
    km = mgr.kernel_map(x.coordinate_map_key, out_key, kernel_size, 1, False)
    return km.nbr, km.nbrT


@pytest.mark.parametrize("ks,stride,tr", [(3, 1, False), (3, 2, False), (1, 2, False), (5, 1, False), (9, 1, False),
                                          (2, 2, True), (3, 3, True)])
def test_kernel_map_bit_exact(oracle, hip, ks, stride, tr):
    coords = surface_coords(6000, batch=2, extent=30, seed=ks * 10 + stride)
    ref, out = both(oracle, hip, lambda c: _maps(c, ks, stride, tr), coords)
    eq(ref[0], out[0])
    eq(ref[1], out[1])


def test_kernel_map_at_given_coordinates(oracle, hip):
    coords = surface_coords(4000, seed=5)
    given = rand_coords(3000, batch=2, extent=30, seed=9, dup=0.0)
    given = torch.unique(given, dim=0).int().contiguous()
    ref, out = both(oracle, hip, lambda c, g: _maps(c, 5, 1, False, g), coords, given)
    eq(ref[0], out[0])
    eq(ref[1], out[1])


def test_pair_lists_bit_exact(oracle, hip):
    coords = surface_coords(6000, batch=2, extent=30, seed=77)

    def fn(c):
        x = me.SparseTensor(coordinates=c, features=torch.zeros(c.shape[0], 1, device=c.device))
        mgr = x.coordinate_manager
        km = mgr.kernel_map(x.coordinate_map_key, mgr.stride(x.coordinate_map_key, 2), 3, 1, False)
        pin, pout, off, P = km.pairs()
        return pin[:P], pout[:P], torch.from_numpy(off)
    ref, out = both(oracle, hip, fn, coords)
    for r, o in zip(ref, out):
        eq(r, o)


# ------------------------------------------------------------------ sparse convolution
def _conv_case(coords, feats, w, bias, dy, ks, stride, fn=None):
    x = me.SparseTensor(coordinates=coords, features=feats)
    mgr = x.coordinate_manager
    out_key = mgr.stride(x.coordinate_map_key, stride) if stride > 1 else x.coordinate_map_key
    km = mgr.kernel_map(x.coordinate_map_key, out_key, ks, 1, False)
    xf = x.F.detach().clone().requires_grad_(True)
    wp = w.detach().clone().requires_grad_(True)
    bp = bias.detach().clone().requires_grad_(True) if bias is not None else None
    y = (fn or me.SparseConvFunction).apply(xf, wp, bp, km)
    g = dy[: y.shape[0]]
    (y * g).sum().backward()
    res = [y.detach(), xf.grad, wp.grad]
    if bp is not None:
        res.append(bp.grad)
    return res


@pytest.mark.parametrize("cin,cout,ks,stride,n", [
    (3, 64, 3, 1, 9000), (64, 64, 3, 1, 9000), (64, 64, 3, 2, 9000), (64, 128, 3, 2, 6000),
    (128, 128, 3, 1, 4000), (256, 256, 3, 1, 1500), (128, 256, 3, 2, 3000), (64, 64, 9, 1, 700),
    (64, 128, 5, 1, 1500), (5, 7, 3, 1, 1000), (64, 192, 3, 1, 1000), (64, 64, 1, 2, 5000), (16, 24, 3, 1, 130),
])
@pytest.mark.parametrize("form", ["pairs", "implicit"])
def test_spconv_fwd_bwd_matches_oracle(oracle, hip, cin, cout, ks, stride, n, form):
    torch.manual_seed(cin * 1000 + cout + ks)
    coords = surface_coords(n, batch=2, extent=max(8, int(n ** 0.5) // 3), seed=n + ks)
    feats = torch.randn(coords.shape[0], cin)
    w = torch.randn(ks ** 3, cin, cout) / (cin * min(ks, 3) ** 3) ** 0.5
    bias = torch.randn(cout) if cout % 3 == 0 else None
    dy = torch.randn(coords.shape[0], cout)
    fn = me.SparseConvFunction if form == "pairs" else me.ImplicitConvFunction
    ref, out = both(oracle, hip, _conv_case, coords, feats, w, bias, dy, ks, stride, fn)
    names = ["y", "dx", "dw", "db"]
    for nm, r, o in zip(names, ref, out):
        scale = float(r.abs().max()) if r.numel() else 1.0
        try:
            close(r, o, scale)
        except AssertionError as e:  # pragma: no cover
            raise AssertionError("%s mismatch (cin=%d cout=%d ks=%d): %s" % (nm, cin, cout, ks, e))


@pytest.mark.parametrize("cin,cout,ks,stride,n", [(64, 64, 3, 1, 9000), (64, 128, 3, 2, 6000), (128, 128, 3, 1, 4000),
                                                  (256, 256, 3, 1, 1500), (128, 64, 3, 1, 3000), (64, 192, 3, 1, 1000), (16, 24, 3, 1, 130)])
def test_spconv_bf16_operands_match_oracle_emulation(oracle, hip, cin, cout, ks, stride, n):
    """precision 1: bf16 (RNE) operands, fp32 accumulate -- against the oracle's bit-level emulation of the
    same rounding, so only the fp32 summation order differs."""
    torch.manual_seed(cin + cout)
    coords = surface_coords(n, batch=2, extent=max(8, int(n ** 0.5) // 3), seed=n + ks)
    feats = torch.randn(coords.shape[0], cin)
    w = torch.randn(ks ** 3, cin, cout) / (cin * 27) ** 0.5
    dy = torch.randn(coords.shape[0], cout)
    me.PRECISION = 1
    try:
        ref, out = both(oracle, hip, _conv_case, coords, feats, w, None, dy, ks, stride)
    finally:
        me.PRECISION = 0
    ref32, _ = both(oracle, None, _conv_case, coords, feats, w, None, dy, ks, stride) if hip is None else (ref, None)
    for nm, r, o in zip(["y", "dx", "dw"], ref, out):
        close(r, o, float(r.abs().max()))
    # and the bf16 result is a bf16-accurate approximation of the fp32 one
    with _lib.use_library(oracle):
        full = _conv_case(coords, feats, w, None, dy, ks, stride)
    assert (ref[0] - full[0]).abs().max() <= 2e-2 * float(full[0].abs().max())


@pytest.mark.parametrize("cin,cout,n", [(64, 64, 4000), (128, 128, 4000), (128, 256, 2500), (256, 128, 2500), (64, 32, 900),
                                       (96, 64, 1500), (72, 48, 900), (192, 80, 1200)])     # partial 64-channel chunk / column tile
def test_spconv_output_stationary_bf16_matches_pair_form_oracle(oracle, hip, cin, cout, n):
    """bf16 mode picks the atomic-free output-stationary kernel on dense-enough maps: same operator."""
    torch.manual_seed(cin * 3 + cout)
    coords = rand_coords(n, batch=1, extent=6, seed=n, dup=0.0)      # compact blob: ~50 % neighbourhood occupancy
    feats = torch.randn(coords.shape[0], cin)
    w = torch.randn(27, cin, cout) / (cin * 27) ** 0.5
    bias = torch.randn(cout)
    dy = torch.randn(coords.shape[0], cout)
    me.PRECISION = 1
    old, old_t = me.IMPLICIT_MIN_OCCUPANCY, me.IMPLICIT_MIN_TILES
    me.IMPLICIT_MIN_TILES = 0
    try:
        me.IMPLICIT_MIN_OCCUPANCY = 0.0          # force the output-stationary kernel on the device
        ref, out = both(oracle, hip, _conv_case, coords, feats, w, bias, dy, 3, 1)
        me.IMPLICIT_MIN_OCCUPANCY = 2.0          # and the pair form
        _, out_pairs = both(oracle, hip, _conv_case, coords, feats, w, bias, dy, 3, 1)
    finally:
        me.PRECISION, me.IMPLICIT_MIN_OCCUPANCY, me.IMPLICIT_MIN_TILES = 0, old, old_t
    for r, o, q in zip(ref, out, out_pairs):
        close(r, o, float(r.abs().max()))
        close(q, o, float(r.abs().max()))


@pytest.mark.parametrize("cin,cout,prec", [(64, 64, 0), (64, 3, 0), (128, 18, 0), (64, 64, 1), (256, 128, 1)])
def test_linear_split_row_weight_gradient(oracle, hip, cin, cout, prec):
    """1x1x1 convolutions: forward / data gradient (fp32: library GEMMs; bench precision with 64-multiple channels:
    cg3d_linear_fwd on bf16 rows, on both sides), the split-over-rows wgrad kernel backward == x^T @ dy."""
    torch.manual_seed(cin + cout)
    n = 20000
    x, w, b, dy = torch.randn(n, cin), torch.randn(cin, cout) / cin ** 0.5, torch.randn(cout), torch.randn(n, cout)

    def fn(x, w, b, dy):
        x, w, b = [t.clone().requires_grad_(True) for t in (x, w, b)]
        me.PRECISION = prec
        try:
            y = me.linear(x, w, b)
            (y * dy).sum().backward()
        finally:
            me.PRECISION = 0
        return y.detach(), x.grad, w.grad, b.grad
    ref, out = both(oracle, hip, fn, x, w, b, dy)
    exact = (x.t().double() @ dy.double()).float()
    for i, (r, o) in enumerate(zip(ref, out)):
        if i == 2 and prec:
            continue            # bf16 operands on the device only (the oracle side of this op is the fp32 GEMM)
        close(r, o, float(r.abs().max()))
    tol = (2e-2 if prec else 2e-4) * float(exact.abs().max())
    assert (out[2].cpu() - exact).abs().max() <= tol


@pytest.mark.parametrize("ks,frac", [(3, (0, 0.2, 0.2, 0.6, 1.0)), (5, (0, 0.0005, 0.05, 1.0)), (9, (0, 0.3, 1.0))])
def test_grouped_conv_row_groups_with_their_own_weights(oracle, hip, ks, frac):
    """me.grouped_conv (the class branches): every row group convolves with its own weights; bf16 mode runs the
    LDS-staged tile kernel on group-aligned tiles on the device (forward, and the data gradient on the SAME plan with
    the weight slots reversed), the oracle side the stacked pair form -- same operator, per-group weight gradients come
    back as separate tensors."""
    cin, cout, G = 64, 64, len(frac) - 1
    coords = torch.unique(rand_coords(4000, batch=1, extent=7, seed=ks, dup=0.0), dim=0)     # ~60 % of the cells: dense map
    n = coords.shape[0]
    bounds = tuple(int(round(f * n)) for f in frac)           # includes an empty group and a 1-2 row group
    for g in range(G):                                        # every group its own batch index: no pair crosses groups
        coords[bounds[g]:bounds[g + 1], 0] = g
    torch.manual_seed(ks)
    feats, dy = torch.randn(n, cin), torch.randn(n, cout)
    ws = [torch.randn(ks ** 3, cin, cout) / (cin * ks ** 3) ** 0.5 for _ in range(G)]

    def fn(coords, feats, dy, *ws):
        x = me.SparseTensor(coordinates=coords.float(), features=feats)
        km = x.coordinate_manager.kernel_map(x.coordinate_map_key, x.coordinate_map_key, ks, 1, False)
        f = x.F.clone().requires_grad_(True)
        wl = [w.clone().requires_grad_(True) for w in ws]
        me.PRECISION = 1
        try:
            y = me.grouped_conv(f, wl, km, bounds, closed=True)
            (y * dy).sum().backward()
        finally:
            me.PRECISION = 0
        return (y.detach(), f.grad) + tuple(w.grad for w in wl)
    ref, out = both(oracle, hip, fn, coords, feats, dy, *ws)
    for r, o in zip(ref, out):
        close(r, o, float(r.abs().max()))


def test_spconv_empty_and_tiny(oracle, hip):
    for n in (1, 2, 33):
        coords = rand_coords(n, batch=1, extent=2, seed=n, dup=0.0)
        feats = torch.randn(n, 8)
        w = torch.randn(27, 8, 8)
        dy = torch.randn(n, 8)
        ref, out = both(oracle, hip, _conv_case, coords, feats, w, None, dy, 3, 1)
        for r, o in zip(ref, out):
            close(r, o, float(r.abs().max()))


# ------------------------------------------------------------------ interpolation / pooling
def _interp_case(coords, feats, q, ts, dout):
    x = me.SparseTensor(coordinates=coords, features=feats)
    mgr = x.coordinate_manager
    key = mgr.stride(x.coordinate_map_key, ts) if ts > 1 else x.coordinate_map_key
    src = mgr.get(key)
    f = torch.randn(src.n, feats.shape[1], generator=torch.Generator().manual_seed(3)).to(feats.device)
    f.requires_grad_(True)
    st = me.SparseTensor(features=f, coordinate_map_key=key, coordinate_manager=mgr)
    lib = _lib.get()
    from ctypes import c_int32, c_int64
    idx = torch.empty((q.shape[0], 8), dtype=torch.int32, device=q.device)
    w = torch.empty((q.shape[0], 8), dtype=torch.float32, device=q.device)
    lib.call("cg3d_interp_map", _lib.ptr(q), c_int64(q.shape[0]), c_int32(ts), _lib.ptr(src.keys), _lib.ptr(src.vals),
             c_int64(src.cap), _lib.ptr(idx), _lib.ptr(w), lib.stream())
    out = st.features_at_coordinates(q)
    (out * dout).sum().backward()
    return idx, w, out.detach(), f.grad


@pytest.mark.parametrize("ts,c", [(2, 128), (4, 128), (8, 256), (2, 7)])
def test_interpolation_matches_oracle(oracle, hip, ts, c):
    coords = surface_coords(5000, extent=40, seed=ts)
    feats = torch.zeros(coords.shape[0], c)
    q = torch.unique(coords, dim=0).float().contiguous()
    q[:, 1:] += torch.rand(q.shape[0], 3, generator=torch.Generator().manual_seed(1)) * 0.999  # off-lattice
    dout = torch.randn(q.shape[0], c)
    ref, out = both(oracle, hip, _interp_case, coords, feats, q, ts, dout)
    eq(ref[0], out[0])          # corner rows: bit-exact
    eq(ref[1], out[1])          # weights: bit-exact (same fp32 op order)
    close(ref[2], out[2], float(ref[2].abs().max()))
    close(ref[3], out[3], float(ref[3].abs().max()))


def _pool_case(coords, feats, ks, stride, dout):
    x = me.SparseTensor(coordinates=coords, features=feats)
    xf = x.F.detach().clone().requires_grad_(True)
    xs = x._like(xf)
    pool = me.MinkowskiAvgPooling(kernel_size=ks, stride=stride)
    y = pool(xs)
    (y.F * dout[: len(y)]).sum().backward()
    pmap = x.coordinate_manager._kmaps[("pool", x.coordinate_map_key, y.coordinate_map_key, ks)]
    return pmap, y.C, y.F.detach(), xf.grad


@pytest.mark.parametrize("ks,stride", [(5, 2), (9, 4), (17, 8), (33, 16)])
def test_avgpool_matches_oracle(oracle, hip, ks, stride):
    coords = surface_coords(3000, extent=50, seed=ks)
    feats = torch.randn(coords.shape[0], 64)
    dout = torch.randn(coords.shape[0], 64)
    ref, out = both(oracle, hip, _pool_case, coords, feats, ks, stride, dout)
    eq(ref[0], out[0])
    eq(ref[1], out[1])
    close(ref[2], out[2], float(ref[2].abs().max()))
    close(ref[3], out[3], float(ref[3].abs().max()))


def test_quantise_average_matches_oracle(oracle, hip):
    coords = rand_coords(20000, batch=4, extent=12, seed=4, dup=0.5).float()
    feats = torch.randn(coords.shape[0], 64)

    def fn(c, f):
        f = f.clone().requires_grad_(True)
        t = me.SparseTensor(coordinates=c, features=f, quantization_mode=me.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
        t.F.square().sum().backward()
        return t.C, t.inverse_mapping, t.F.detach(), f.grad
    ref, out = both(oracle, hip, fn, coords, feats)
    eq(ref[0], out[0])
    eq(ref[1], out[1])
    close(ref[2], out[2], 4.0)
    close(ref[3], out[3], 8.0)


# ------------------------------------------------------------------ fused BN (+res) (+act)
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("c,bounds", [(64, (0, 30000)), (128, (0, 1, 700, 701, 5000)), (1024, (0, 1229)), (256, (0, 257, 257, 9000))])
def test_fused_bn_act_matches_oracle(oracle, hip, act, c, bounds):
    torch.manual_seed(c + act)
    n, G = bounds[-1], len(bounds) - 1
    x, res, dy = torch.randn(n, c) * 2 + 0.5, torch.randn(n, c), torch.randn(n, c)
    gamma, beta = torch.rand(G, c) + 0.5, torch.randn(G, c)

    def fn(x, res, dy, gamma, beta):
        x, res, gamma, beta = [t.clone().requires_grad_(True) for t in (x, res, gamma, beta)]
        y, mean, var, _ = me.FusedBNActFunction.apply(x, gamma, beta, res, bounds, act, True, None, None, 1e-5)
        (y * dy).sum().backward()
        return y.detach(), mean, var, x.grad, res.grad, gamma.grad, beta.grad
    ref, out = both(oracle, hip, fn, x, res, dy, gamma, beta)
    for r, o in zip(ref, out):
        close(r, o, float(r.abs().max()))


@pytest.mark.parametrize("G", [1, 3])
def test_fused_bn_running_statistics_match_torch_batchnorm(oracle, hip, G):
    """The statistics launch also updates running_mean / running_var / num_batches_tracked (G == 1) exactly as
    nn.BatchNorm1d does; grouped calls go through the foreach path -- same contract."""
    torch.manual_seed(G)
    c, bounds = 64, tuple(range(0, 700 * G + 1, 700))
    x = torch.randn(bounds[-1], c) * 3 + 1

    def fn(x):
        bns = [torch.nn.BatchNorm1d(c).to(x.device) for _ in range(G)]
        refs = [torch.nn.BatchNorm1d(c).to(x.device) for _ in range(G)]
        for _ in range(2):
            y = me.fused_bn_act(x, bns, bounds, me.ACT_RELU)
            yr = torch.cat([torch.relu(refs[g](x[bounds[g]:bounds[g + 1]])) for g in range(G)])
        assert (y - yr).abs().max() < 1e-4
        for b, r in zip(bns, refs):
            assert (b.running_mean - r.running_mean).abs().max() < 1e-5
            assert (b.running_var - r.running_var).abs().max() < 1e-4
            assert int(b.num_batches_tracked) == int(r.num_batches_tracked) == 2
        return torch.stack([b.running_mean for b in bns]), torch.stack([b.running_var for b in bns])
    ref, out = both(oracle, hip, fn, x)
    for r, o in zip(ref, out):
        close(r, o, 1.0)


@pytest.mark.parametrize("n,g,seg", [(0, 4, False), (5, 0, False), (3000, 37, False), (20000, 64, True)])
def test_points_in_boxes_bit_exact(oracle, hip, n, g, seg):
    """find_points_in_boxes as one fused op: identical masks on both libraries, points placed ON faces included."""
    from cagroup3d_amd.ops.iou3d_nms_utils import points_in_boxes
    boxes = rand_boxes(g, seed=g + 1)
    rs = np.random.RandomState(n + g)
    pts = torch.from_numpy(rs.uniform(-6, 6, (n, 3)).astype(np.float32))
    if n and g:
        k = min(n, g)
        pts[:k] = boxes[:k, :3]                                   # box centres: inside
        pts[k:2 * k, :] = boxes[:k, :3][: max(0, min(k, n - k))]   # and points exactly on a face: not strictly inside
        pts[k:2 * k, 2] += boxes[: max(0, min(k, n - k)), 5] / 2
    ps = torch.from_numpy(rs.randint(0, 4, n).astype(np.int32)) if seg else None
    bs = torch.from_numpy(rs.randint(0, 4, g).astype(np.int32)) if seg else None
    ref, out = both(oracle, hip, points_in_boxes, pts, boxes, ps, bs)
    eq(ref, out)
    if n and g and not seg:
        assert bool(ref[0, 0]) and ref.shape == (n, g) and ref.dtype == torch.bool


# ------------------------------------------------------------------ iou3d_nms
@pytest.mark.parametrize("na,nb", [(0, 5), (1, 1), (17, 33), (300, 257)])
def test_boxes_overlap_and_iou_bit_exact(oracle, hip, na, nb):
    a, b = rand_boxes(na, seed=na), rand_boxes(nb, seed=nb + 100)
    if na and nb:
        b[0] = a[0]                       # identical boxes
        b[-1, :6] = a[-1, :6]             # same box, other heading
    for fn in (iou3d_nms_utils.boxes_overlap_bev, iou3d_nms_utils.boxes_iou_bev):
        ref, out = both(oracle, hip, fn, a, b)
        eq(ref, out)


@pytest.mark.parametrize("n", [0, 1, 64, 65, 1000, 3000])
@pytest.mark.parametrize("rotated", [False, True])
def test_nms_keep_bit_exact(oracle, hip, n, rotated):
    boxes = rand_boxes(n, seed=n, yaw=rotated, extent=3.0)
    scores = torch.rand(n, generator=torch.Generator().manual_seed(n))
    fn = iou3d_nms_utils.nms_gpu if rotated else iou3d_nms_utils.nms_normal_gpu
    ref, out = both(oracle, hip, lambda b, s: fn(b, s, 0.5)[0], boxes, scores)
    eq(ref, out)
    if n:
        assert 0 < out.numel() <= n


@pytest.mark.parametrize("n,rotated", [(0, True), (1, False), (300, True), (777, False)])
def test_literal_nms_entry_points_match_the_device_form(oracle, hip, n, rotated):
    """cg3d_nms_gpu / cg3d_nms_normal_gpu (the reference's `nms_gpu(boxes, keep, thresh) -> num` with a HOST keep tensor,
    iou3d_nms.h:9-12): same kept indices as cg3d_nms, on the oracle and on the device."""
    boxes = rand_boxes(n, seed=n + 11, yaw=rotated, extent=3.0)
    scores = torch.rand(n, generator=torch.Generator().manual_seed(n))
    b_sorted = boxes[scores.sort(0, descending=True)[1]].contiguous()
    ext = iou3d_nms_utils.iou3d_nms_cuda

    def fn(b):
        keep = torch.full((max(n, 1),), -1, dtype=torch.int64)
        num = (ext.nms_gpu if rotated else ext.nms_normal_gpu)(b, keep, 0.5)
        dk, dn, _ = iou3d_nms_utils._nms_sorted(b, 0.5, rotated)
        return keep[:num], dk[: int(dn.item())]
    ref, out = both(oracle, hip, fn, b_sorted)
    eq(ref[0], out[0])
    eq(out[0], out[1].cpu())
    eq(ref[0], ref[1])


def test_nms_mask_words_bit_exact(oracle, hip):
    n = 500
    boxes = rand_boxes(n, seed=3, yaw=True, extent=2.0)

    def fn(b):
        keep, num, mask = iou3d_nms_utils._nms_sorted(b, 0.3, True)
        return keep[: int(num.item())], mask
    ref, out = both(oracle, hip, fn, boxes)
    eq(ref[0], out[0])
    cb = (n + 63) // 64
    mr, mo = ref[1].view(n, cb), out[1].cpu().view(n, cb)
    for i in range(n):       # only tiles on/above the diagonal are produced by the HIP kernel
        assert torch.equal(mr[i, i // 64:], mo[i, i // 64:])


def test_nms_batched_matches_single(oracle, hip):
    sizes = [0, 5, 64, 130, 1, 999]
    boxes = [rand_boxes(s, seed=s + 7, yaw=False, extent=2.5) for s in sizes]
    seg = [0]
    for s in sizes:
        seg.append(seg[-1] + s)
    ref, out = both(oracle, hip, lambda b: iou3d_nms_utils.nms_batched_sorted(b, seg, 0.5, False), torch.cat(boxes))
    eq(ref[1], out[1])
    for g, s in enumerate(sizes):
        k = int(ref[1][g])
        eq(ref[0][seg[g]: seg[g] + k], out[0][seg[g]: seg[g] + k])
        if s:
            sr, so = both(oracle, hip, lambda b: iou3d_nms_utils._nms_sorted(b, 0.5, False)[:2], boxes[g])
            eq(so[0][: int(so[1].item())], out[0][seg[g]: seg[g] + k])


# ------------------------------------------------------------------ knn / sort_vertices
@pytest.mark.parametrize("b,n,m,k", [(1, 5000, 4000, 1), (2, 1500, 700, 1), (1, 300, 257, 5), (2, 64, 10, 16), (1, 1, 3, 1)])
def test_knn_bit_exact(oracle, hip, b, n, m, k):
    g = torch.Generator().manual_seed(n + m)
    xyz = torch.rand(b, n, 3, generator=g) * 4
    q = torch.rand(b, m, 3, generator=g) * 4
    if n > 10:
        xyz[:, 5] = xyz[:, 2]            # exact ties: the lower index must win
        q[:, 0] = xyz[:, 2]
    ref, out = both(oracle, hip, lambda x, c: knn_mod.knn_with_dist(k, x, c), xyz, q)
    eq(ref[0], out[0])
    eq(ref[1], out[1])
    if n > 10 and k == 1:
        assert int(out[0][0, 0, 0]) == 2


def test_boxes_bev_iou_cpu_entry_point_equals_device_kernel(oracle, hip):
    """boxes_iou_bev_cpu (iou3d_nms_api.cpp:16): host pointers through the HIP library's host instantiation of the same
    functions == its kernel == the oracle, bit for bit."""
    a, b = rand_boxes(150, seed=3), rand_boxes(90, seed=4)
    with _lib.use_library(hip):
        host = iou3d_nms_utils.boxes_bev_iou_cpu(a, b)
        dev = iou3d_nms_utils.boxes_iou_bev(a.cuda(), b.cuda())
        assert isinstance(iou3d_nms_utils.boxes_bev_iou_cpu(a.numpy(), b.numpy()), np.ndarray)
    with _lib.use_library(oracle):
        ref = iou3d_nms_utils.boxes_iou_bev(a, b)
    eq(host, dev)
    eq(host, ref)


@pytest.mark.parametrize("b,n,m,radius,nsample", [(1, 5000, 3000, 0.3, 16), (2, 700, 300, 0.5, 64), (1, 200, 65, 0.05, 8),
                                                  (2, 64, 10, 10.0, 100), (1, 1, 3, 0.1, 4), (1, 50000, 2000, 0.2, 32)])
def test_ball_query_bit_exact(oracle, hip, b, n, m, radius, nsample):
    """f4: the wave-ballot ball query == the reference's one-thread-per-query walk (first nsample hits in index order, the
    first hit repeated in unfilled slots, zeros when a ball is empty), including exact ties at the radius."""
    from cagroup3d_amd.ops.ball_query import ball_query
    g = torch.Generator().manual_seed(n + m)
    xyz = torch.rand(b, n, 3, generator=g) * 4
    q = torch.rand(b, m, 3, generator=g) * 4
    q[:, 0] = 100.0                                                   # an empty ball
    ref, out = both(oracle, hip, lambda x, c: ball_query(radius, nsample, x, c), xyz.contiguous(), q.contiguous())
    eq(ref, out)
    assert int(out[0, 0].abs().sum()) == 0
    # independent check of the semantics on one batch element
    d2 = ((q[0, :, None, :] - xyz[0, None, :, :]) ** 2).sum(-1)
    for qi in (1, m - 1):
        hits = torch.nonzero(d2[qi] < radius * radius).view(-1)[:nsample].tolist()
        exp = (hits + [hits[0]] * (nsample - len(hits))) if hits else [0] * nsample
        got = out[0, qi].cpu().tolist()
        # (torch's broadcast expression may round a distance at the radius differently: only such rows may differ)
        assert got == exp or any(abs(float(d2[qi, k]) - radius * radius) < 1e-5 for k in set(got) ^ set(exp))


@pytest.mark.parametrize("b,n,m,far", [(1, 6000, 2000, 0.0), (2, 9000, 5000, 0.2), (1, 20000, 30000, 1.0)])
def test_knn1_uniform_grid_is_exact(oracle, hip, b, n, m, far):
    """k = 1 on large problems goes through the cell grid: same indices and distances as the exhaustive scan, including
    exact ties (lowest index), clustered points, and queries far from every point (redone exhaustively)."""
    g = torch.Generator().manual_seed(n + m)
    xyz = torch.rand(b, n, 3, generator=g) * torch.tensor([8.0, 6.0, 3.0])
    xyz[:, : n // 4] = xyz[:, : n // 4] * 0.02 + 1.0                  # a dense cluster: many points per cell
    xyz[:, 7] = xyz[:, 3]                                            # exact duplicates
    q = xyz[:, torch.randint(0, n, (m,), generator=g)] + torch.randn(b, m, 3, generator=g) * 0.01
    nfar = int(m * far)
    if nfar:
        q[:, :nfar] = torch.rand(b, nfar, 3, generator=g) * 40 - 20    # far outside the point cloud
    q[:, -1] = xyz[:, 3]
    ref, out = both(oracle, hip, lambda x, c: knn_mod.knn_with_dist(1, x, c), xyz, q)
    eq(ref[0], out[0])
    eq(ref[1], out[1])
    assert int(out[0][0, -1, 0]) == 3


def test_sort_vertices_bit_exact(oracle, hip):
    g = torch.Generator().manual_seed(0)
    v = torch.rand(2, 1500, 24, 2, generator=g)
    m = torch.rand(2, 1500, 24, generator=g) > 0.8
    # cap the number of valid vertices at 8 like real polygon candidates
    csum = m.int().cumsum(-1)
    m = m & (csum <= 8)
    nv = m.int().sum(-1).int()
    mean = (v * m.unsqueeze(-1)).sum(2, keepdim=True) / nv.clamp(min=1).view(2, 1500, 1, 1)
    v = v - mean
    ref, out = both(oracle, hip, rotated_iou.sort_v, v, m, nv)
    eq(ref, out)


def test_rotated_iou3d_pairs(oracle, hip):
    a = rand_boxes(800, seed=1).view(1, -1, 7)
    b = rand_boxes(800, seed=2).view(1, -1, 7)
    b[0, :100] = a[0, :100]
    b[0, :100, :3] += 0.05
    ref, out = both(oracle, hip, rotated_iou.cal_iou_3d, a, b)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
def test_pinned_staging_ring_survives_wraparound(hip):
    """me.h2d stages small host tables in two fixed pinned halves; shrink them so that a few hundred copies wrap the
    ring many times while the stream is busy, and check every table arrived intact."""
    half = me._PinnedStage.HALF
    me._PinnedStage.HALF = 4096
    me._PinnedStage._rings.clear()
    try:
        busy = torch.randn(2048, 2048, device="cuda")
        outs, refs = [], []
        g = torch.Generator().manual_seed(0)
        for i in range(300):
            if i % 10 == 0:
                busy = busy @ busy * 1e-3          # keep the stream behind the host
            n = int(torch.randint(1, 700, (1,), generator=g))
            t = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64).to(torch.int32)
            refs.append(t)
            outs.append(me.h2d(t, torch.int32, "cuda"))
        big = torch.arange(5000, dtype=torch.int32)                # larger than a half: the pin_memory() fallback
        assert torch.equal(me.h2d(big, torch.int32, "cuda").cpu(), big)
        torch.cuda.synchronize()
        for o, r in zip(outs, refs):
            assert torch.equal(o.cpu(), r)
    finally:
        me._PinnedStage.HALF = half
        me._PinnedStage._rings.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["defer", "stream"])
def test_weight_plan_early_and_late_rows_equal_the_single_launch(hip, mode):
    """me.set_early_weights + prepare_weights(split=True): the early weights' rows lead the table and are converted now, the
    others when me.run_late() says so (defer) or on the late stream (stream); afterwards every copy holds the bits of the
    one-launch conversion.  A look-up of a late weight while its rows are deferred launches them by itself, and an unsplit call
    (split=False) flushes first."""
    me._WeightPlan.reset()
    me.PRECISION = 1
    keep = (me.LATE_MODE, me.LATE_WEIGHTS)
    me.LATE_MODE, me.LATE_WEIGHTS = mode, True
    try:
        torch.manual_seed(0)
        ws = [torch.randn(27, 64, 64, device="cuda"), torch.randn(8, 72, 48, device="cuda"), torch.randn(27, 128, 256, device="cuda")]
        grp = [torch.randn(125, 64, 64, device="cuda") for _ in range(3)]
        for w in ws:
            me._prep_bf16_both(w)
        me._prep_bf16_group(grp, True), me._prep_bf16_group(grp, False)
        me.prepare_weights()
        ref = [tuple(t.clone() for t in me._prep_bf16_both(w)) for w in ws]
        ref_g = (me._prep_bf16_group(grp, True).clone(), me._prep_bf16_group(grp, False).clone())
        me.finish_weights()
        nrows = me._WeightPlan.nrows
        me.set_early_weights([ws[1]])                        # (not the first recorded: its rows must move to the front)
        for w in ws + grp:
            w.add_(0.5)
        for step in range(2):
            me.prepare_weights(training=True, split=True)
            assert me._WeightPlan.nrows == nrows and me._WeightPlan.n_early == 8 * 2, (me._WeightPlan.nrows, me._WeightPlan.n_early)
            assert (len(me._DEFERRED) == 1) if mode == "defer" else bool(me._LATE_PENDING), "the late rows must not have run here"
            early = me._prep_bf16_both(ws[1])                # early: valid without anything else
            assert (len(me._DEFERRED) == 1) if mode == "defer" else bool(me._LATE_PENDING)
            if step == 0:
                me.run_late()
            got = [me._prep_bf16_both(w) for w in ws]        # (step 1: the look-up of a late weight launches the deferred rows)
            assert not me._DEFERRED
            got_g = (me._prep_bf16_group(grp, True), me._prep_bf16_group(grp, False))
            me.finish_weights()
            assert not me._LATE_PENDING
            with torch.no_grad():
                exp = [me._prep_bf16_both(w) for w in ws]    # converted on the spot (the arena is not trusted outside a forward)
            for (a_t, a_p), (b_t, b_p), (r_t, r_p) in zip(got, exp, ref):
                assert torch.equal(a_t, b_t) and torch.equal(a_p, b_p) and not torch.equal(a_t, r_t)
            assert torch.equal(early[0], exp[1][0])
            assert not torch.equal(got_g[0], ref_g[0]) and not torch.equal(got_g[1], ref_g[1])
            for w in ws + grp:
                w.add_(0.25)
        me.prepare_weights(training=True, split=True)
        assert me._DEFERRED or me._LATE_PENDING
        me.prepare_weights(training=True)                    # unsplit: flushes first, one launch
        assert not me._DEFERRED and not me._LATE_PENDING
        me.finish_weights()
    finally:
        me._DEFERRED.clear()
        me.LATE_MODE, me.LATE_WEIGHTS = keep
        me.set_early_weights(None)
        me._WeightPlan.reset()
        me.PRECISION = 0


@pytest.mark.gpu
def test_weight_plan_single_launch_equals_per_layer_conversion(hip):
    """me.prepare_weights(): every recorded conv weight converted by ONE table-driven launch into the arena -- the same
    bits as the per-layer launches, refreshed exactly when a weight's version changes."""
    me._WeightPlan.reset()
    me.PRECISION = 1
    try:
        torch.manual_seed(0)
        ws = [torch.randn(27, 64, 64, device="cuda"), torch.randn(8, 72, 48, device="cuda"), torch.randn(27, 128, 256, device="cuda")]
        grp = [torch.randn(125, 64, 64, device="cuda") for _ in range(3)]
        first = [me._prep_bf16_both(w) for w in ws[:2]] + [(me._prep_bf16_t(ws[2]), None)]        # per-layer launches; recorded
        g_t, g_p = me._prep_bf16_group(grp, True), me._prep_bf16_group(grp, False)
        wf = torch.randn(27, 128, 64, device="cuda")         # the tile kernel's operands: both copies in MFMA fragment order
        f_t, f_p = me._prep_frag(wf, True, True)
        gf_t, gf_p = me._prep_bf16_group(grp, True, True), me._prep_bf16_group(grp, False, True)
        assert me._WeightPlan.dirty and me._WeightPlan.table is None
        me.prepare_weights()
        assert me._WeightPlan.nrows == 27 + 8 * 2 + 27 * 2 * 4 + 2 * 3 * 125 + 27 * 2 + 2 * 3 * 125
        a_t, a_p = me._prep_frag(wf, True, True)
        assert a_t.data_ptr() != f_t.data_ptr() and torch.equal(a_t.view(-1), f_t.view(-1)) and torch.equal(a_p.view(-1), f_p.view(-1))
        assert torch.equal(me._prep_bf16_group(grp, True, True).view(-1), gf_t.view(-1))
        assert torch.equal(me._prep_bf16_group(grp, False, True).view(-1), gf_p.view(-1))
        again = [me._prep_bf16_both(w) for w in ws[:2]] + [(me._prep_bf16_t(ws[2]), None)]
        for (a_t, a_p), (b_t, b_p), w in zip(first, again, ws):
            assert b_t.data_ptr() != a_t.data_ptr() and torch.equal(a_t, b_t)                 # answered from the arena
            assert (a_p is None and b_p is None) or torch.equal(a_p, b_p)
        assert torch.equal(me._prep_bf16_group(grp, True), g_t) and torch.equal(me._prep_bf16_group(grp, False), g_p)
        arena_ptr = again[0][0].data_ptr()
        me.finish_weights()                                  # outside a detector forward the arena is never trusted
        assert me._prep_bf16_both(ws[0])[0].data_ptr() != arena_ptr
        # a fused optimizer step changes the weights WITHOUT bumping the version: a training forward converts always
        ws[0].data.copy_(ws[0].data + 1.0)
        with torch.no_grad():
            expect = me._prep_bf16_both(ws[0])               # converted on the spot
        me.prepare_weights(training=True)
        fresh = me._prep_bf16_both(ws[0])
        assert fresh[0].data_ptr() == arena_ptr and torch.equal(fresh[0], expect[0]) and torch.equal(fresh[1], expect[1])
        assert not torch.equal(fresh[0], first[0][0])
        me.finish_weights()
        # inference: converts once after a training forward, then only when a version changes
        me.prepare_weights(training=False)
        n0 = me._WeightPlan.pending
        assert n0 is False
        ws[1].add_(1.0)
        me.prepare_weights(training=False)
        again2 = me._prep_bf16_both(ws[1])
        me.finish_weights()
        assert torch.equal(again2[0], me._prep_bf16_both(ws[1])[0])
    finally:
        me.PRECISION = 0
        me._WeightPlan.reset()


@pytest.mark.parametrize("n,c,two", [(5000, 64, True), (777, 128, True), (82107, 128, True), (3000, 256, False), (1, 64, True)])
def test_add_relu_rows_is_exact(oracle, hip, n, c, two):
    """me.add_relu (cg3d_bn_apply with the identity normalisation): relu(a + b), its bf16 row copy and the masked gradient,
    bit for bit what torch computes, on the oracle and on the device."""
    g = torch.Generator().manual_seed(n + c)
    a, b, dy = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)

    def fn(a, b, dy):
        me.PRECISION = 1
        try:
            xa, xb = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
            y = me.add_relu(xa, xb if two else None)
            y16 = me.rows16_of(y.detach() if False else y, True)
            (y * dy).sum().backward()
            return y.detach(), xa.grad, (xb.grad if two else xa.grad), (y16.clone() if y16 is not None else torch.zeros(1))
        finally:
            me.PRECISION = 0
    ref, out = both(oracle, hip, fn, a, b, dy)
    want = torch.relu(a + b if two else a)
    for r, o in zip(ref, out):
        eq(r, o)
    assert torch.equal(out[0].cpu(), want) and torch.equal(out[1].cpu(), dy * (want > 0))
    assert torch.equal(out[3].cpu().view(torch.bfloat16).float(), want.to(torch.bfloat16).float())
