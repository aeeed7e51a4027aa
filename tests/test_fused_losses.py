"""Fused loss terms (ops/fused_losses.py): the oracle's restatement against the torch mirror of the reference lines it
replaces (CPU), the HIP kernels against the oracle (-m gpu), and the dense head's loss with and without them."""
import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib
from cagroup3d_amd.ops import fused_losses
from cagroup3d_amd.pcdet.utils.iou3d_loss import axis_aligned_iou_loss


def _case(n=3000, B=4, seed=0, frac=0.3):
    g = torch.Generator().manual_seed(seed)
    points = (torch.rand(n, 3, generator=g) - 0.5) * 8
    bbox_pred = torch.rand(n, 6, generator=g) * 1.5 + 0.05            # exp(...) of the head: positive face distances
    centerness = torch.randn(n, 1, generator=g)
    ctr_t = torch.rand(n, generator=g)
    # targets: boxes around the points, some far away (no overlap), some tiny
    bbox_t = torch.cat([points + (torch.rand(n, 3, generator=g) - 0.5) * 1.2, torch.rand(n, 3, generator=g) * 2 + 0.1,
                        torch.zeros(n, 1)], 1)
    bbox_t[::17, :3] += 50.0                                           # disjoint boxes: overlap 0
    scene = torch.randint(0, B, (n,), generator=g)
    pos = torch.nonzero(torch.rand(n, generator=g) < frac).squeeze(1)
    n_pos = torch.bincount(scene[pos], minlength=B).float().clamp(min=1.)
    ctr_den = torch.zeros(B).index_add_(0, scene[pos], ctr_t[pos]).clamp(min=1e-6)
    return points, bbox_pred, centerness, ctr_t, bbox_t, scene, pos, n_pos, ctr_den, B


def _torch_chain(points, bbox_pred, centerness, ctr_t, bbox_t, scene, pos, n_pos, ctr_den, B, wc, wb, eps):
    """The lines of `_loss_batched` the fused op replaces (the mirror of reference cagroup_head.py:532-546)."""
    ps = scene[pos]
    pc, pb = centerness[pos], bbox_pred[pos]
    ct = ctr_t[pos].unsqueeze(1)
    bce = torch.nn.functional.binary_cross_entropy_with_logits(pc, ct, reduction="none")
    loss_c = (bce.squeeze(1) * wc / (n_pos[ps] + eps)).sum()
    lo, hi = pb[:, 0:6:2], pb[:, 1:6:2]
    boxes = torch.cat([points[pos] + (hi - lo) / 2, lo + hi], dim=1)
    iou_el = axis_aligned_iou_loss(boxes, bbox_t[pos][:, :6], None, reduction="none")
    loss_b = (iou_el * ct.squeeze(1) * wb / ctr_den[ps]).sum()
    return loss_c, loss_b


def _run_fused(dev, points, bbox_pred, centerness, ctr_t, bbox_t, scene, pos, n_pos, ctr_den, B, wc, wb, eps, gw):
    t = [x.to(dev) for x in (points, bbox_pred, centerness, ctr_t, bbox_t, scene, pos, n_pos, ctr_den)]
    bp, ce = t[1].clone().requires_grad_(True), t[2].clone().requires_grad_(True)
    out = fused_losses.positives_loss(ce, bp, t[0], t[3], t[4], t[5], t[7], t[8], t[6], wc, wb, eps)
    (out[0] * gw[0] + out[1] * gw[1]).backward()
    return out.detach().cpu(), ce.grad.cpu(), bp.grad.cpu()


@pytest.mark.parametrize("seed,frac", [(0, 0.3), (1, 0.02), (2, 1.0)])
def test_oracle_positives_loss_equals_the_torch_chain(oracle, seed, frac):
    c = _case(seed=seed, frac=frac)
    wc, wb, eps, gw = 1.0 / c[-1], 1.0 / c[-1], float(torch.finfo(torch.float32).eps), (0.7, 1.3)
    with _lib.use_library(oracle):
        out, dce, dbp = _run_fused("cpu", *c, wc, wb, eps, gw)
    bp, ce = c[1].clone().requires_grad_(True), c[2].clone().requires_grad_(True)
    lc, lb = _torch_chain(c[0], bp, ce, *c[3:], wc, wb, eps)
    (lc * gw[0] + lb * gw[1]).backward()
    torch.testing.assert_close(out, torch.stack([lc, lb]).detach(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dce, ce.grad, rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(dbp, bp.grad, rtol=1e-4, atol=1e-7)
    other = torch.ones(dbp.shape[0], dtype=torch.bool); other[c[6]] = False
    assert float(dbp[c[6]].abs().sum()) > 0 and float(dbp[other].abs().sum()) == 0.0 and float(dce[other].abs().sum()) == 0.0     # positives only


def test_oracle_positives_loss_without_positives(oracle):
    c = list(_case(seed=3))
    c[6] = c[6][:0]
    with _lib.use_library(oracle):
        out, dce, dbp = _run_fused("cpu", *c, 1.0, 1.0, 1e-7, (1.0, 1.0))
    assert float(out.abs().sum()) == 0.0 and float(dce.abs().sum()) == 0.0 and float(dbp.abs().sum()) == 0.0


@pytest.mark.parametrize("n,d,beta", [(5000, 3, 0.04), (700, 9, 1.0 / 9), (1, 3, 0.04)])
def test_oracle_smooth_l1_rows_equals_torch(oracle, n, d, beta):
    g = torch.Generator().manual_seed(n)
    pred, tgt, w = torch.randn(n, d, generator=g) * 0.1, torch.randn(n, d, generator=g) * 0.1, torch.rand(n, generator=g)
    tgt[::5] = pred[::5]                                               # exact zeros of the difference
    p0 = pred.clone().requires_grad_(True)
    dd = torch.abs(p0 - tgt)
    ref = (torch.where(dd < beta, 0.5 * dd * dd / beta, dd - 0.5 * beta) * w.unsqueeze(1)).sum()
    (ref * 1.7).backward()
    with _lib.use_library(oracle):
        p1 = pred.clone().requires_grad_(True)
        out = fused_losses.smooth_l1_rows(p1, tgt, w.unsqueeze(1), beta)
        (out * 1.7).backward()
    torch.testing.assert_close(out.detach(), ref.detach(), rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(p1.grad, p0.grad, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("dataset,cfgname,ltol,gtol", [("scannet", "S5k", 1e-5, 1e-4), ("sunrgbd", "S5k-yaw", 1e-4, 5e-3)])
def test_dense_head_loss_is_the_same_with_and_without_the_fused_terms(oracle, dataset, cfgname, ltol, gtol):
    """One training step of the detector on the oracle: every loss term and the backbone gradients agree between the fused
    ops and the torch chains they replace (SUN RGB-D: the yaw form, whose oracle gradient is a central difference)."""
    from cagroup3d_amd import build_model
    from cagroup3d_amd.pcdet.models.dense_heads import cagroup_head as H
    res = []
    for fused in (True, False):
        H.FUSED_LOSSES = fused
        try:
            with _lib.use_library(oracle):
                model, _ = build_model.build_cagroup3d(dataset, seed=0)
                model.train()
                model.dense_head.force_gt_selection = True
                model.dense_head.force_class_logit_boost = 6.0
                torch.manual_seed(1); np.random.seed(1)
                ret, tb, _ = model(build_model.synthetic_batch(cfgname, 2, device="cpu"))
                ret["loss"].backward()
        finally:
            H.FUSED_LOSSES = True
        res.append((tb, {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
    (tb1, g1), (tb0, g0) = res
    for k in tb0:
        assert abs(tb0[k] - tb1[k]) <= ltol * max(1.0, abs(tb0[k])), (k, tb0[k], tb1[k])
    assert tb0["loss_bbox"] > 0
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    assert (num / den) ** 0.5 < gtol, (num / den) ** 0.5


@pytest.mark.gpu
@pytest.mark.parametrize("seed,frac,n", [(0, 0.3, 3000), (1, 0.02, 40000), (2, 1.0, 700)])
def test_hip_positives_loss_matches_oracle(oracle, hip, seed, frac, n):
    c = _case(n=n, seed=seed, frac=frac)
    wc, wb, eps, gw = 0.25, 0.25, float(torch.finfo(torch.float32).eps), (0.7, 1.3)
    with _lib.use_library(oracle):
        ref = _run_fused("cpu", *c, wc, wb, eps, gw)
    with _lib.use_library(hip):
        out = _run_fused("cuda", *c, wc, wb, eps, gw)
    torch.testing.assert_close(out[0], ref[0], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out[1], ref[1], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(out[2], ref[2], rtol=2e-4, atol=1e-6)


@pytest.mark.gpu
def test_hip_smooth_l1_rows_matches_oracle(oracle, hip):
    g = torch.Generator().manual_seed(5)
    pred, tgt, w = torch.randn(90000, 3, generator=g) * 0.1, torch.randn(90000, 3, generator=g) * 0.1, torch.rand(90000, generator=g)
    res = []
    for lib, dev in ((oracle, "cpu"), (hip, "cuda")):
        with _lib.use_library(lib):
            p = pred.detach().clone().to(dev).requires_grad_(True)       # a fresh leaf on either device
            out = fused_losses.smooth_l1_rows(p, tgt.to(dev), w.to(dev), 0.04)
            (out * 0.5).backward()
            res.append((out.detach().cpu(), p.grad.cpu()))
    torch.testing.assert_close(res[1][0], res[0][0], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(res[1][1], res[0][1], rtol=1e-5, atol=1e-8)


# ---------------------------------------------------------------------------------------------- fused FCOS assignment
def _assign_case(n=6000, m=40, n_cls=6, B=3, seed=0, yaw=True):
    g = torch.Generator().manual_seed(seed)
    gt = torch.cat([(torch.rand(m, 3, generator=g) - 0.5) * 6, torch.rand(m, 3, generator=g) * 2.0 + 0.3,
                    ((torch.rand(m, 1, generator=g) - 0.5) * 6.0) if yaw else torch.zeros(m, 1)], 1)
    gl = torch.randint(0, n_cls, (m,), generator=g)
    gs = torch.randint(0, B, (m,), generator=g)
    # points: half of them inside some box of their class / scene, the rest anywhere
    pick = torch.randint(0, m, (n,), generator=g)
    pts = gt[pick, :3] + (torch.rand(n, 3, generator=g) - 0.5) * gt[pick, 3:6] * 1.3
    pts[n // 2:] = (torch.rand(n - n // 2, 3, generator=g) - 0.5) * 8
    pc, ps = gl[pick].clone(), gs[pick].clone()
    pc[n // 2:] = torch.randint(0, n_cls, (n - n // 2,), generator=g)
    ps[n // 2:] = torch.randint(0, B, (n - n // 2,), generator=g)
    n_map = torch.full((m,), n // (n_cls * B), dtype=torch.long)
    n_map[::7] = 5                                         # a few boxes compete on tiny maps: k = 5 < TOPK + 1
    return pts, pc, ps, gt, gl, gs, n_map


def _assign(dev, fused, pts, pc, ps, gt, gl, gs, n_map):
    from cagroup3d_amd.pcdet.models.dense_heads.target_assigner import cagroup3d_assigner as A

    class Cfg(dict):
        __getattr__ = dict.__getitem__
    a = A.CAGroup3DAssigner(Cfg(LIMIT=27, TOPK=18, N_SCALES=4))
    old, A.FUSED_ASSIGN = A.FUSED_ASSIGN, fused
    try:
        out = a.assign_all_classes([pts.to(dev)], gt.to(dev), gl.to(dev), pt_cls=pc.to(dev), pt_scene=ps.to(dev), gt_scene=gs.to(dev),
                                   n_map=n_map.to(dev))
    finally:
        A.FUSED_ASSIGN = old
    return [o.cpu() for o in out]


@pytest.mark.parametrize("yaw,seed", [(False, 0), (True, 1), (True, 2)])
def test_oracle_fused_assignment_equals_the_torch_chain(oracle, yaw, seed):
    c = _assign_case(seed=seed, yaw=yaw)
    with _lib.use_library(oracle):
        ctr1, box1, lab1 = _assign("cpu", True, *c)
        ctr0, box0, lab0 = _assign("cpu", False, *c)
    assert torch.equal(lab1, lab0) and int((lab0 >= 0).sum()) > 200
    pos = lab0 >= 0
    torch.testing.assert_close(ctr1[pos], ctr0[pos], rtol=1e-5, atol=1e-6)
    assert torch.equal(box1[pos], box0[pos])


@pytest.mark.gpu
@pytest.mark.parametrize("yaw,seed,n,m", [(False, 0, 28000, 80), (True, 1, 6000, 40), (True, 2, 700, 3)])
def test_hip_fused_assignment_is_bit_identical_to_the_oracle(oracle, hip, yaw, seed, n, m):
    c = _assign_case(n=n, m=m, seed=seed, yaw=yaw)
    with _lib.use_library(oracle):
        ref = _assign("cpu", True, *c)
    with _lib.use_library(hip):
        out = _assign("cuda", True, *c)
    assert torch.equal(out[2], ref[2])
    pos = ref[2] >= 0
    assert torch.equal(out[0][pos], ref[0][pos]) and torch.equal(out[1][pos], ref[1][pos])
