"""not gpu: the CPU oracle against INDEPENDENT references (dense conv3d, explicit loops, torch ops).
MinkowskiEngine is not available, so these are what pins the sparse-engine semantics."""
import itertools

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cagroup3d_amd import _lib, me
from cagroup3d_amd.ops import iou3d_nms_utils, knn as knn_mod, rotated_iou
from util import morton_keys, rand_boxes, rand_coords, surface_coords


@pytest.fixture(autouse=True)
def _bind_oracle(oracle):
    with _lib.use_library(oracle):
        yield


def dense_of(x, G, pad=0):
    B = int(x.C[:, 0].max()) + 1
    C = x.C.long()
    d = torch.zeros(B, x.F.shape[1], G + 2 * pad, G + 2 * pad, G + 2 * pad)
    o = G // 2 + pad
    d[C[:, 0], :, C[:, 1] + o, C[:, 2] + o, C[:, 3] + o] = x.F.detach()
    return d, o


def test_coordinate_map_first_occurrence_order(monkeypatch):
    """cg3d_coord_map_build itself: duplicates merged, representative = first occurrence, rows in first-occurrence order."""
    monkeypatch.setattr(me, "MORTON_ROWS", False)
    c = torch.tensor([[0, 5, 5, 5], [0, 1, 1, 1], [0, 5, 5, 5], [1, 1, 1, 1], [0, 1, 1, 1], [0, -3, 2, 9]], dtype=torch.int32)
    f = torch.arange(6, dtype=torch.float32).view(6, 1)
    x = me.SparseTensor(coordinates=c, features=f)
    assert x.C.tolist() == [[0, 5, 5, 5], [0, 1, 1, 1], [1, 1, 1, 1], [0, -3, 2, 9]]
    assert x.unique_index.tolist() == [0, 1, 3, 5] and x.inverse_mapping.tolist() == [0, 1, 0, 2, 1, 3]
    assert x.F.view(-1).tolist() == [0., 1., 3., 5.]                     # first row of each voxel
    xa = me.SparseTensor(coordinates=c, features=f, quantization_mode=me.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
    assert xa.F.view(-1).tolist() == [1.0, 2.5, 3.0, 5.0]                # mean over the voxel's rows


def test_inserted_maps_are_in_batch_morton_order():
    """The engine inserts every map in (batch, Morton) order; unique_index / inverse_mapping keep the caller's row numbers,
    the representative of a voxel stays its first occurrence, and strided maps inherit the order."""
    assert me.MORTON_ROWS
    c = torch.tensor([[0, 5, 5, 5], [0, 1, 1, 1], [0, 5, 5, 5], [1, 1, 1, 1], [0, 1, 1, 1], [0, -3, 2, 9]], dtype=torch.int32)
    f = torch.arange(6, dtype=torch.float32).view(6, 1)
    x = me.SparseTensor(coordinates=c, features=f)
    assert x.C.tolist() == [[0, -3, 2, 9], [0, 1, 1, 1], [0, 5, 5, 5], [1, 1, 1, 1]]
    assert x.unique_index.tolist() == [5, 1, 0, 3] and x.inverse_mapping.tolist() == [2, 1, 2, 3, 1, 0]
    assert x.F.view(-1).tolist() == [5., 1., 0., 3.]
    xa = me.SparseTensor(coordinates=c, features=f, quantization_mode=me.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE)
    assert xa.F.view(-1).tolist() == [5.0, 2.5, 1.0, 3.0]
    # a bigger map against numpy: sorted by key, first occurrence per voxel, strided maps sorted by THEIR keys as well
    coords = rand_coords(4000, batch=3, extent=40, seed=3)
    x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1))
    keys = morton_keys(x.C.numpy())
    assert (np.diff(keys.astype(np.float64)) > 0).all() and len(keys) == len(np.unique(coords.numpy(), axis=0))
    first = {}
    for i, row in enumerate(map(tuple, coords.tolist())):
        first.setdefault(row, i)
    assert x.unique_index.tolist() == [first[tuple(r)] for r in x.C.tolist()]
    assert torch.equal(x.C[x.inverse_mapping.long()], coords)
    key2 = x.coordinate_manager.stride(x.coordinate_map_key, 2)
    k2 = morton_keys(x.coordinate_manager.get(key2).coords.numpy())
    assert (np.diff(k2.astype(np.float64)) > 0).all()
    # float coordinates are floored (negative values too)
    xf = me.SparseTensor(coordinates=torch.tensor([[0, -0.5, 0.5, 1.99], [0, -1.0, 0.0, 1.0]]), features=torch.ones(2, 1))
    assert xf.C.tolist() == [[0, -1, 0, 1]]


def test_stride_map_floors_negative_coordinates(monkeypatch):
    monkeypatch.setattr(me, "MORTON_ROWS", False)
    c = torch.tensor([[0, -1, -2, -3], [0, 0, 1, 3], [0, -4, 2, 2], [0, 1, 0, 2]], dtype=torch.int32)
    x = me.SparseTensor(coordinates=c, features=torch.ones(4, 1))
    key = x.coordinate_manager.stride(x.coordinate_map_key, 2)
    assert x.coordinate_manager.get(key).coords.tolist() == [[0, -2, -2, -4], [0, 0, 0, 2], [0, -4, 2, 2]]
    assert x.coordinate_manager.stride(x.coordinate_map_key, 2) is key         # cached / shared by later layers


@pytest.mark.parametrize("ks,stride,cin,cout", [(3, 1, 5, 7), (3, 2, 4, 6), (5, 1, 3, 4), (1, 2, 4, 4)])
def test_sparse_conv_equals_dense_conv3d(ks, stride, cin, cout):
    torch.manual_seed(ks * 10 + stride)
    G = 12
    coords = rand_coords(500, batch=2, extent=G // 2, seed=ks)
    x = me.SparseTensor(coordinates=coords, features=torch.randn(coords.shape[0], cin))
    conv = me.MinkowskiConvolution(cin, cout, kernel_size=ks, stride=stride, bias=True)
    xf = x.F.clone().requires_grad_(True)
    y = conv(x._like(xf))
    pad = ks // 2
    dense, o = dense_of(x, G, pad)
    dense.requires_grad_(True)
    w = conv._w3().detach().clone().requires_grad_(True)
    wd = w.view(ks, ks, ks, cin, cout).permute(4, 3, 0, 1, 2)               # offset index k = (ix, iy, iz), iz fastest
    yd = F.conv3d(dense, wd, padding=0) + conv.bias.detach().view(1, -1, 1, 1, 1)
    Co = y.C.long()
    yref = yd[Co[:, 0], :, Co[:, 1] + o - pad, Co[:, 2] + o - pad, Co[:, 3] + o - pad]
    torch.testing.assert_close(y.F, yref, rtol=1e-4, atol=1e-5)
    if stride > 1:      # output coordinates are the even lattice, each exactly once
        assert (y.C[:, 1:] % stride == 0).all() and len(torch.unique(y.C, dim=0)) == len(y)
    g = torch.randn_like(y.F)
    (y.F * g).sum().backward()
    (yref * g).sum().backward()
    C = x.C.long()
    torch.testing.assert_close(xf.grad, dense.grad[C[:, 0], :, C[:, 1] + o, C[:, 2] + o, C[:, 3] + o], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(conv.kernel.grad.view_as(w), w.grad, rtol=1e-4, atol=1e-4)


def test_transposed_conv_is_adjoint_of_strided_conv():
    """<convT(y), x> == <y, conv(x)> with the same [K, Cin, Cout] kernel transposed -- even kernel k2 s2."""
    torch.manual_seed(0)
    coords = surface_coords(900, batch=2, extent=10, seed=3)
    x = me.SparseTensor(coordinates=coords, features=torch.randn(coords.shape[0], 6))
    down = me.MinkowskiConvolution(6, 4, kernel_size=2, stride=2)
    y = down(x)
    up = me.MinkowskiConvolutionTranspose(4, 6, kernel_size=2, stride=2)
    with torch.no_grad():
        up.kernel.copy_(down.kernel.transpose(1, 2))
    z = up(y._like(torch.randn(len(y), 4)))
    assert z.coordinate_map_key == x.coordinate_map_key                    # lands on the existing finer map
    yy = torch.randn(len(y), 4)
    lhs = (up(y._like(yy)).F * x.F).sum()
    rhs = (yy * y.F).sum()
    torch.testing.assert_close(lhs, rhs, rtol=1e-4, atol=1e-4)


def test_generative_transpose_onto_given_coordinates():
    coords = torch.tensor([[0, 0, 0, 0], [0, 3, 0, 0]], dtype=torch.int32)
    x = me.SparseTensor(coordinates=coords, features=torch.tensor([[1.0], [10.0]]), tensor_stride=3)
    up = me.MinkowskiGenerativeConvolutionTranspose(1, 1, kernel_size=3, stride=3)
    with torch.no_grad():
        up.kernel.copy_(torch.arange(27.).view(27, 1, 1))
    tgt = torch.tensor([[0, 1, 0, 0], [0, 2, 0, 0], [0, 0, 0, -1], [0, 7, 7, 7]], dtype=torch.int32)
    z = up(x, tgt)
    assert z.C.tolist() == tgt.tolist()
    # o = i + off: (1,0,0) = (0,0,0)+(+1,0,0) -> k=(2,1,1)=22 ; (2,0,0) = (3,0,0)+(-1,0,0) -> k=(0,1,1)=4, x10
    assert z.F.view(-1).tolist() == [22.0, 40.0, 12.0, 0.0]


def test_conv_at_given_coordinates_uses_input_stride_spacing():
    coords = torch.tensor([[0, 0, 0, 0], [0, 2, 0, 0], [0, 4, 0, 0]], dtype=torch.int32)
    x = me.SparseTensor(coordinates=coords, features=torch.tensor([[1.0], [2.0], [4.0]]), tensor_stride=2)
    conv = me.MinkowskiConvolution(1, 1, kernel_size=3)
    with torch.no_grad():
        conv.kernel.fill_(1.0)
    y = conv(x, torch.tensor([[0, 2, 0, 0], [0, 6, 0, 0], [0, 1, 0, 0]], dtype=torch.int32))
    assert y.F.view(-1).tolist() == [7.0, 4.0, 0.0]       # offsets are multiples of 2; (1,0,0) sees nothing


def test_trilinear_interpolation_against_loops():
    torch.manual_seed(0)
    coords = surface_coords(600, batch=2, extent=10, seed=1)
    x = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1))
    mgr = x.coordinate_manager
    key = mgr.stride(x.coordinate_map_key, 4)
    src = mgr.get(key)
    f = torch.randn(src.n, 3, dtype=torch.float64)
    st = me.SparseTensor(features=f.float(), coordinate_map_key=key, coordinate_manager=mgr)
    q = torch.unique(coords, dim=0).float()[:200]
    q[:, 1:] += torch.rand(200, 3) * 0.9
    out = st.features_at_coordinates(q)
    table = {tuple(c): i for i, c in enumerate(src.coords.tolist())}
    ref = torch.zeros(200, 3, dtype=torch.float64)
    for i in range(200):
        b = int(q[i, 0])
        base = [int(np.floor(float(q[i, d + 1]) / 4)) * 4 for d in range(3)]
        for dx, dy, dz in itertools.product((0, 1), repeat=3):
            c = (b, base[0] + 4 * dx, base[1] + 4 * dy, base[2] + 4 * dz)
            if c in table:
                wgt = 1.0
                for d, dd in enumerate((dx, dy, dz)):
                    r = (float(q[i, d + 1]) - base[d]) / 4
                    wgt *= r if dd else 1 - r
                ref[i] += wgt * f[table[c]]
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-6)
    # on-lattice queries return the stored feature
    on = st.features_at_coordinates(src.coords.float())
    torch.testing.assert_close(on, f.float(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("ks,stride", [(5, 2), (9, 4), (17, 8)])
def test_avg_pool_counts_present_inputs_only(ks, stride):
    coords = surface_coords(400, batch=2, extent=12, seed=ks)
    x = me.SparseTensor(coordinates=coords, features=torch.randn(coords.shape[0], 2))
    y = me.MinkowskiAvgPooling(kernel_size=ks, stride=stride)(x)
    half = (ks - 1) // 2
    xin = x.C.tolist()
    for o, row in zip(y.C.tolist(), y.F):
        sel = [i for i, c in enumerate(xin) if c[0] == o[0] and all(abs(c[d] - o[d]) <= half for d in (1, 2, 3))]
        assert sel, "output coordinate without inputs"
        torch.testing.assert_close(row, x.F[sel].mean(0), rtol=1e-5, atol=1e-6)
    assert (y.C[:, 1:] % stride == 0).all()


def naive_nms(boxes, thr, iou_fn):
    keep, alive = [], [True] * len(boxes)
    for i in range(len(boxes)):
        if not alive[i]:
            continue
        keep.append(i)
        for j in range(i + 1, len(boxes)):
            if alive[j] and iou_fn(boxes[i], boxes[j]) > thr:
                alive[j] = False
    return keep


def test_nms_normal_against_python_loop():
    boxes = rand_boxes(300, seed=5, yaw=False, extent=2.0)
    scores = torch.rand(300, generator=torch.Generator().manual_seed(5))

    def iou(a, b):
        l, r = max(a[0] - a[3] / 2, b[0] - b[3] / 2), min(a[0] + a[3] / 2, b[0] + b[3] / 2)
        t, bt = max(a[1] - a[4] / 2, b[1] - b[4] / 2), min(a[1] + a[4] / 2, b[1] + b[4] / 2)
        inter = max(r - l, 0) * max(bt - t, 0)
        return inter / max(a[3] * a[4] + b[3] * b[4] - inter, 1e-8)
    order = scores.sort(0, descending=True)[1]
    ref = naive_nms(boxes[order].double().tolist(), 0.25, iou)
    keep, _ = iou3d_nms_utils.nms_normal_gpu(boxes, scores, 0.25)
    assert keep.tolist() == order[ref].tolist()


def test_nms_rotated_consistent_with_iou_matrix():
    boxes = rand_boxes(200, seed=9, yaw=True, extent=1.5)
    scores = torch.rand(200, generator=torch.Generator().manual_seed(9))
    order = scores.sort(0, descending=True)[1]
    iou = iou3d_nms_utils.boxes_iou_bev(boxes[order], boxes[order])
    ref = naive_nms(list(range(200)), 0.3, lambda i, j: float(iou[i, j]))
    keep, _ = iou3d_nms_utils.nms_gpu(boxes, scores, 0.3)
    assert keep.tolist() == order[ref].tolist()
    # idempotence: NMS of the kept set keeps everything
    keep2, _ = iou3d_nms_utils.nms_gpu(boxes[keep], scores[keep], 0.3)
    assert len(keep2) == len(keep)


def test_bev_overlap_known_answers():
    sq = torch.tensor([[0, 0, 0, 2, 2, 1, 0.]])
    r45 = torch.tensor([[0, 0, 0, 2, 2, 1, np.pi / 4]])
    sh = torch.tensor([[1, 0, 0, 2, 2, 1, 0.]])
    far = torch.tensor([[9, 9, 0, 2, 2, 1, 0.3]])
    assert abs(float(iou3d_nms_utils.boxes_overlap_bev(sq, sq)) - 4.0) < 1e-5
    assert abs(float(iou3d_nms_utils.boxes_overlap_bev(sq, r45)) - 8 * (2 ** 0.5 - 1)) < 1e-4     # regular octagon
    assert abs(float(iou3d_nms_utils.boxes_iou_bev(sq, sh)) - 1 / 3) < 1e-6
    assert float(iou3d_nms_utils.boxes_overlap_bev(sq, far)) == 0.0


def _rect(b):
    """corners of the BEV rectangle of box (x, y, z, dx, dy, dz, heading), anticlockwise, float64"""
    c, s = np.cos(b[6]), np.sin(b[6])
    loc = np.array([[1, 1], [-1, 1], [-1, -1], [1, -1]], np.float64) * np.array([b[3], b[4]], np.float64) / 2
    return loc @ np.array([[c, s], [-s, c]]) + np.array([b[0], b[1]], np.float64)


def _clip_area(P, Q):
    """area of the intersection of two convex anticlockwise polygons: Sutherland-Hodgman clipping of P by the edges of Q,
    shoelace formula -- nothing in common with the reference's vertex collection + angular sort"""
    out = [tuple(p) for p in P]
    for i in range(len(Q)):
        a, b = Q[i], Q[(i + 1) % len(Q)]
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        src, out = out, []
        for j in range(len(src)):
            p, q = src[j], src[(j + 1) % len(src)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        if not out:
            return 0.0
    x, y = np.array([p[0] for p in out]), np.array([p[1] for p in out])
    return 0.5 * abs(float(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1))))


def test_bev_overlap_and_iou_against_polygon_clipping():
    """the restated iou3d_nms_kernel.cu:104-372 arithmetic (fp32, the reference's operation order) against an independent
    float64 computation, over boxes that overlap, touch, contain each other and miss"""
    g = torch.Generator().manual_seed(5)
    a, b = rand_boxes(60, 5, extent=1.2), rand_boxes(50, 6, extent=1.2)
    b[:10, :3] = a[:10, :3] + 0.05 * torch.randn(10, 3, generator=g)      # near-coincident centres
    b[10:14] = a[10:14]                                                       # identical boxes
    b[14:18, 3:6] = a[14:18, 3:6] * 0.3                                       # ... and contained ones
    b[14:18, :3] = a[14:18, :3]
    ov = iou3d_nms_utils.boxes_overlap_bev(a, b).numpy()
    iou = iou3d_nms_utils.boxes_iou_bev(a, b).numpy()
    an, bn = a.double().numpy(), b.double().numpy()
    ref = np.array([[_clip_area(_rect(x), _rect(y)) for y in bn] for x in an])
    area_a, area_b = an[:, 3] * an[:, 4], bn[:, 3] * bn[:, 4]
    assert (ref > 0).mean() > 0.1 and (ref == 0).mean() > 0.1
    # The reference counts a corner as inside the other box up to MARGIN = 1e-2 outside it (iou3d_nms_kernel.cu:53-60), so
    # its polygon can reach a centimetre past the true intersection: areas agree to MARGIN x (a side length), not to fp32
    # rounding (measured here: max 3e-3, mean 1e-4 on sides of 0.2 ... 1.7) -- still two orders below what a swapped
    # dx / dy, a mirrored heading or a half-extent slip produces (checked below on the same boxes).
    err = np.abs(ov - ref)
    assert err.max() <= 1e-2 and err.mean() <= 5e-4, (err.max(), err.mean())
    ref_iou = ref / np.maximum(area_a[:, None] + area_b[None, :] - ref, 1e-8)
    assert np.abs(iou - ref_iou).max() <= 2e-2
    assert (ov[10:14, 10:14].diagonal() - area_a[10:14]).__abs__().max() <= 1e-2          # identical boxes: the box itself
    assert np.abs(ov[14:18, 14:18].diagonal() - area_b[14:18]).max() <= 1e-2               # contained: the inner box
    mirrored = bn.copy()
    mirrored[:, 6] *= -1
    wrong = np.array([[_clip_area(_rect(x), _rect(y)) for y in mirrored] for x in an])
    assert np.abs(ov - wrong).max() > 0.1


@pytest.mark.parametrize("k", [1, 4])
def test_knn_against_cdist(k):
    g = torch.Generator().manual_seed(k)
    xyz, q = torch.rand(2, 500, 3, generator=g), torch.rand(2, 77, 3, generator=g)
    idx, d2 = knn_mod.knn_with_dist(k, xyz, q)
    ref_d, ref_i = torch.cdist(q.double(), xyz.double()).pow(2).topk(k, dim=2, largest=False)
    assert torch.equal(idx.long(), ref_i)
    torch.testing.assert_close(d2.double(), ref_d, rtol=1e-4, atol=1e-6)
    assert knn_mod.knn(k, xyz, q).shape == (2, k, 77)                    # the reference wrapper's layout


def test_sort_vertices_is_anticlockwise_from_positive_x_axis():
    g = torch.Generator().manual_seed(3)
    n = 300
    ang = torch.rand(1, n, 24, generator=g) * 2 * np.pi
    rad = torch.rand(1, n, 24, generator=g) * 0.5 + 0.5
    v = torch.stack([rad * torch.cos(ang), rad * torch.sin(ang)], -1)
    m = torch.rand(1, n, 24, generator=g) > 0.75
    m = m & (m.int().cumsum(-1) <= 8)
    nv = m.int().sum(-1).int()
    idx = rotated_iou.sort_v(v, m, nv)
    for i in range(n):
        k = int(nv[0, i])
        if k < 3:
            assert len(set(idx[0, i].tolist())) == 1 and not bool(m[0, i, idx[0, i, 0]])      # all padding
            continue
        order = idx[0, i, :k].tolist()
        assert sorted(order) == sorted(torch.nonzero(m[0, i]).view(-1).tolist())
        a = ang[0, i, order]
        # ascending angle starting just above -0 ... the first vertex is the one closest after the +x axis
        assert bool((a[1:] > a[:-1]).all()), (a, order)
        assert idx[0, i, k] == idx[0, i, 0]


def test_ball_query_oracle_against_torch():
    """f4: first nsample in-radius reference rows in index order, first hit repeated, zeros for an empty ball."""
    from cagroup3d_amd.ops.ball_query import ball_query
    g = torch.Generator().manual_seed(1)
    xyz, q = torch.rand(2, 400, 3, generator=g) * 2, torch.rand(2, 90, 3, generator=g) * 2
    q[0, 3] = 50.0
    r, ns = 0.35, 12
    idx = ball_query(r, ns, xyz.contiguous(), q.contiguous())
    assert idx.shape == (2, 90, ns) and idx.dtype == torch.int32
    d2 = ((q[:, :, None, :] - xyz[:, None, :, :]) ** 2).sum(-1)
    for b in range(2):
        for j in range(90):
            hits = torch.nonzero(d2[b, j] < r * r).view(-1)[:ns].tolist()
            exp = (hits + [hits[0]] * (ns - len(hits))) if hits else [0] * ns
            assert idx[b, j].tolist() == exp, (b, j)
