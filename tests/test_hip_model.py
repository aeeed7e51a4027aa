"""-m gpu: the full detector on the HIP library against the same detector on the CPU oracle (S5k scenes),
and size-independent properties of the hot-path ops at BASELINE.json's full S50k sizes."""
import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, build_model, me, synthetic
from cagroup3d_amd.ops import iou3d_nms_utils, knn as knn_mod
from util import rand_boxes

pytestmark = pytest.mark.gpu


def _step(model, cfgname, dev):
    model.zero_grad()
    torch.manual_seed(1)
    np.random.seed(1)
    batch = build_model.synthetic_batch(cfgname, 2, device=dev)
    ret, tb, _ = model(batch)
    ret["loss"].backward()
    return batch, tb, {n: p.grad.detach().cpu() for n, p in model.named_parameters()}


@pytest.mark.parametrize("dataset,cfgname", [("scannet", "S5k"), ("sunrgbd", "S5k-yaw")])
def test_full_detector_hip_matches_oracle(oracle, hip, dataset, cfgname):
    model, cfg = build_model.build_cagroup3d(dataset, seed=0)
    model.train()
    model.dense_head.force_gt_selection = True
    model.dense_head.force_class_logit_boost = 6.0
    with _lib.use_library(oracle):
        b0, tb0, g0 = _step(model, cfgname, "cpu")
    state = {k: v.clone() for k, v in model.state_dict().items()}       # BN running stats moved during step 0
    model2, _ = build_model.build_cagroup3d(dataset, seed=0)
    model2.dense_head.force_gt_selection = True
    model2.dense_head.force_class_logit_boost = 6.0
    model2 = model2.cuda().train()
    with _lib.use_library(hip):
        b1, tb1, g1 = _step(model2, cfgname, "cuda")
    # voxelisation: bit-exact coordinates and row order
    assert torch.equal(b0["sp_tensor"].C, b1["sp_tensor"].C.cpu())
    # shared head part (same coordinates on both sides): fp32 tolerance
    sem0, off0 = b0["one_stage_results"][1], b0["one_stage_results"][2]
    sem1, off1 = b1["one_stage_results"][1], b1["one_stage_results"][2]
    torch.testing.assert_close(sem1.F.cpu(), sem0.F, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(off1.F.cpu(), off0.F, rtol=1e-3, atol=1e-4)
    # class-branch maps quantise the VOTED positions floor((xyz + offset) / v_c): a 1e-6 difference in an offset
    # can move a vote across a voxel boundary, so voxel counts may differ by a few rows -- not more
    x0, x1 = b0["one_stage_results"][0], b1["one_stage_results"][0]
    n0 = sum(x0[3][c][s].shape[0] for c in range(len(x0[3])) for s in range(2))
    n1 = sum(x1[3][c][s].shape[0] for c in range(len(x1[3])) for s in range(2))
    assert abs(n0 - n1) <= max(3, 0.005 * n0), (n0, n1)
    print("loss terms oracle vs hip:", {k: (round(tb0[k], 6), round(tb1[k], 6)) for k in tb0})
    for k in tb0:                                       # SURVEY 8(d) asks 1 %; measured: every term agrees to 1e-6
        assert abs(tb0[k] - tb1[k]) <= 1e-3 * max(1.0, abs(tb0[k])), (k, tb0[k], tb1[k])
    p0, p1 = [len(p[0]) for p in b0["pred_bbox_list"]], [len(p[0]) for p in b1["pred_bbox_list"]]
    assert all(abs(a - b) <= max(3, 0.05 * a) for a, b in zip(p0, p1)), (p0, p1)
    # backbone gradients (independent of the vote quantisation noise up to the loss terms it feeds)
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    print("whole-model gradient, hip vs oracle, relative L2: %.3e" % (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 5e-3, (num / den) ** 0.5         # measured 1.0e-3 / 1.2e-3 (fp32 atomics, ~60 layers deep)


def test_bf16_mode_stays_within_stated_tolerance_of_fp32(hip):
    """The bench precision (bf16 MFMA operands, fp32 accumulate/storage) against the fp32 parity configuration on
    the same weights and scenes: shared-head features within 1e-2 (semantic) / 4e-2 (offset) of their scale at the worst
    row, 2e-3 / 4e-2 relative RMS, every loss term within
    2 % (the build's own tolerance -- the reference has no bf16 behaviour)."""
    outs = []
    for prec in (0, 1):
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        model.dense_head.force_gt_selection = True
        model.dense_head.force_class_logit_boost = 6.0
        model = model.cuda().train()
        me.PRECISION = prec
        try:
            with _lib.use_library(hip):
                b, tb, g = _step(model, "S5k", "cuda")
        finally:
            me.PRECISION = 0
        outs.append((b["one_stage_results"][1].F.detach(), b["one_stage_results"][2].F.detach(), tb, g))
    (sem0, off0, tb0, g0), (sem1, off1, tb1, g1) = outs
    def rel(a, b):
        scale = max(float(b.abs().max()), 1.0)
        return float((a - b).abs().max()) / scale, float((a - b).pow(2).mean().sqrt()) / float(b.pow(2).mean().sqrt())
    (sem_max, sem_rms), (off_max, off_rms) = rel(sem1, sem0), rel(off1, off0)
    print("bf16 vs fp32 features: semantic max %.2e rms %.2e, offset max %.2e rms %.2e" % (sem_max, sem_rms, off_max, off_rms))
    # measured over 8 runs (round 3: every convolution incl. the 1x1x1 ones on bf16 operands): semantic max 1.9-2.6e-3;
    # offset max 1.6-2.3e-2 of the largest offset (one row out of 5 k decides it; fp32 atomics move it run to run)
    assert sem_max <= 1e-2 and off_max <= 4e-2, (sem_max, off_max)
    assert sem_rms <= 2e-3 and off_rms <= 4e-2, (sem_rms, off_rms)      # measured 1.6e-4 / 1.8e-2 (offsets at initialisation are small numbers)
    print("loss terms fp32 vs bf16:", {k: (round(tb0[k], 5), round(tb1[k], 5)) for k in tb0})
    for k in tb0:
        assert abs(tb0[k] - tb1[k]) <= 2e-2 * max(1.0, abs(tb0[k])), (k, tb0[k], tb1[k])
    # the total loss: within 1 % (the bar VERDICT r01 item 3 names)
    tot = [k for k in tb0 if k in ("loss_all", "loss")]
    for k in tot:
        assert abs(tb0[k] - tb1[k]) <= 1e-2 * abs(tb0[k]), (k, tb0[k], tb1[k])
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    # gradients: a sanity bound, not a precision claim -- bf16 rounding moves some votes across class-voxel
    # boundaries (discrete changes of the class maps), which dominates this number (0.16 measured on S5k)
    print("bf16 vs fp32 whole-model gradient, relative L2: %.3f" % (num / den) ** 0.5)
    assert (num / den) ** 0.5 < 0.3, (num / den) ** 0.5


def test_prefetched_coordinates_give_the_same_step(hip):
    """CAGroup3D.prefetch_coordinates (side stream, coordinate-only dry run of the backbone) must hand the forward
    exactly the structures it would have built itself."""
    res = []
    for use_prefetch in (False, True):
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        model.dense_head.force_gt_selection = True
        model = model.cuda().train()
        torch.manual_seed(1)
        np.random.seed(1)
        batch = build_model.synthetic_batch("S5k", 2, device="cuda")
        with _lib.use_library(hip):
            if use_prefetch:
                batch["prepared"] = model.prefetch_coordinates(batch)
                assert batch["prepared"] is not None
            ret, tb, _ = model(batch)
            ret["loss"].backward()
        torch.cuda.synchronize()
        res.append((batch["sp_tensor"].C.clone(), batch["one_stage_results"][1].F.detach().clone(), tb,
                    {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
    (c0, f0, tb0, g0), (c1, f1, tb1, g1) = res
    assert torch.equal(c0, c1)
    torch.testing.assert_close(f1, f0, rtol=1e-4, atol=1e-5)
    for k in tb0:
        assert abs(tb0[k] - tb1[k]) <= 1e-3 * max(1.0, abs(tb0[k])), (k, tb0[k], tb1[k])
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    assert (num / den) ** 0.5 < 2e-2


def test_worker_thread_prefetch_while_another_step_runs(hip):
    """prefetch_coordinates_async: the dry run of batch B on the worker thread WHILE the main thread runs a full training
    step on batch A (its own coordinate structures, autograd, the per-thread COORDS_ONLY flag) -- then B's step with the
    prefetched structures == B's step without them, from the same weights."""
    res = []
    for use_prefetch in (False, True):
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        model.dense_head.force_gt_selection = True
        model = model.cuda().train()
        a = build_model.synthetic_batch("S5k", 2, device="cuda")
        b = build_model.synthetic_batch("S5k", 2, first_scene=2, device="cuda")   # other scenes
        with _lib.use_library(hip):
            handle = model.prefetch_coordinates_async(b) if use_prefetch else None
            ret, _, _ = model(a)
            ret["loss"].backward()                                            # A's step; gradients are discarded below
            assert not me.coords_only(), "the worker's COORDS_ONLY flag must not leak into this thread"
            for p in model.parameters():
                p.grad = None
            if use_prefetch:
                b["prepared"] = handle.result()
                assert b["prepared"] is not None
            ret, tb, _ = model(b)
            ret["loss"].backward()
        torch.cuda.synchronize()
        res.append((b["sp_tensor"].C.clone(), tb, {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    (c0, tb0, g0), (c1, tb1, g1) = res
    assert torch.equal(c0, c1)
    for k in tb0:
        assert abs(tb0[k] - tb1[k]) <= 1e-3 * max(1.0, abs(tb0[k])), (k, tb0[k], tb1[k])
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    assert (num / den) ** 0.5 < 2e-2


def _s50k_tensor():
    batch = synthetic.make_batch("S50k", 4)
    pts = torch.from_numpy(batch["points"]).cuda()
    coords = pts[:, :4].clone()
    coords[:, 1:] /= 0.02
    return me.SparseTensor(coordinates=coords, features=pts[:, 4:] / 255.), pts


def test_full_size_coordinate_and_kernel_map_properties(hip):
    x, pts = _s50k_tensor()
    n = len(x)
    assert 150000 < n < 200000
    # idempotence: re-inserting the unique coordinates is the identity
    out, keys, vals, cap, uniq, inv = me._build_map(x.C, 1)
    assert torch.equal(out, x.C) and torch.equal(uniq.long(), torch.arange(n, device="cuda")) and torch.equal(inv, uniq)
    # every input point maps to a voxel holding its floored coordinate
    vox = torch.floor(pts[:, 1:4] / 0.02).int()
    assert torch.equal(x.C[x.inverse_mapping.long(), 1:], vox)
    # stride-1 k3 kernel map is symmetric: nbr[k][o] = i  <=>  nbr[26-k][i] = o
    km = x.coordinate_manager.kernel_map(x.coordinate_map_key, x.coordinate_map_key, 3, 1, False)
    nbr = km.nbr.long()
    assert torch.equal(nbr[13], torch.arange(n, device="cuda"))
    for k in (0, 5, 12):
        o = torch.nonzero(nbr[k] >= 0).squeeze(1)
        assert torch.equal(nbr[26 - k][nbr[k][o]], o)
    assert torch.equal(km.nbrT, km.nbr.flip(0))
    pin, pout, off, P = km.pairs()
    assert P == int((km.nbr >= 0).sum()) and off[-1] == P
    assert torch.equal(nbr[(torch.arange(P, device="cuda") >= torch.tensor(off[1:-1], device="cuda").view(-1, 1)).sum(0), pout[:P].long()], pin[:P].long())


def test_full_size_conv_linearity_and_forms_agree(hip):
    x, _ = _s50k_tensor()
    mgr = x.coordinate_manager
    k2 = mgr.stride(x.coordinate_map_key, 2)
    km = mgr.kernel_map(x.coordinate_map_key, k2, 3, 1, False)
    torch.manual_seed(0)
    a, b = torch.randn(len(x), 64, device="cuda"), torch.randn(len(x), 64, device="cuda")
    w = torch.randn(27, 64, 64, device="cuda") * 0.05
    f = lambda t: me.SparseConvFunction.apply(t, w, None, km)
    torch.testing.assert_close(f(2.0 * a - 3.0 * b), 2.0 * f(a) - 3.0 * f(b), rtol=1e-3, atol=1e-3)
    # the atomic pair-list form and the output-stationary implicit form are the same operator
    torch.testing.assert_close(f(a), me.ImplicitConvFunction.apply(a, w, None, km), rtol=1e-4, atol=1e-4)


def test_full_size_bf16_direct_operand_form_agrees_with_pair_form(hip):
    """k_spconv_implicit_bf16_ad at a launch of > 768 workgroups (no offset split, three workgroups per CU) against the
    atomic pair-list bf16 kernel on the same bf16 rows and weights: same products, fp32 sums in another order."""
    x, _ = _s50k_tensor()
    mgr = x.coordinate_manager
    k2 = mgr.stride(x.coordinate_map_key, 2)
    km = mgr.kernel_map(k2, k2, 3, 1, False)
    assert km.n_out > 128 * 768
    torch.manual_seed(1)
    prec, me.PRECISION = me.PRECISION, 1
    try:
        for cin, cout in ((64, 64), (128, 128), (64, 128)):
            xin = me._to_bf16(torch.randn(km.n_in, cin, device="cuda"))
            w = torch.randn(27, cin, cout, device="cuda") * 0.05
            P = int((km.nbr >= 0).sum())
            y_ad = me._conv_implicit_bf16(xin, me._prep_bf16_t(w), km.nbr, None, km.n_out, cin, cout, P)
            pin, pout, off, _ = km.pairs()
            seg, nseg = km.segments(128)
            y_pairs = me._conv_pairs(xin, w, pin, pout, seg, nseg, None, km.n_out, P)
            torch.testing.assert_close(y_ad, y_pairs, rtol=1e-3, atol=1e-3)
    finally:
        me.PRECISION = prec


def test_full_size_knn_self_query_and_nms_idempotence(hip):
    _, pts = _s50k_tensor()
    xyz = pts[pts[:, 0] == 0, 1:4].contiguous()[None]
    xyz = torch.unique(xyz[0], dim=0)[None].contiguous()
    idx, d2 = knn_mod.knn_with_dist(1, xyz, xyz)
    assert torch.equal(idx.view(-1).long(), torch.arange(xyz.shape[1], device="cuda")) and float(d2.max()) == 0.0
    boxes = rand_boxes(18000, seed=1, yaw=False, extent=6.0, device="cuda")
    scores = torch.rand(18000, device="cuda")
    keep, _ = iou3d_nms_utils.nms_normal_gpu(boxes, scores, 0.5)
    keep2, _ = iou3d_nms_utils.nms_normal_gpu(boxes[keep], scores[keep], 0.5)
    assert 0 < len(keep) < 18000 and len(keep2) == len(keep)
    kb = boxes[keep][:3000]
    lo, hi = kb[:, :2] - kb[:, 3:5] / 2, kb[:, :2] + kb[:, 3:5] / 2                # axis-aligned BEV IoU (iou_normal)
    wh = (torch.min(hi[:, None], hi[None]) - torch.max(lo[:, None], lo[None])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    area = kb[:, 3] * kb[:, 4]
    iou = inter / (area[:, None] + area[None] - inter).clamp(min=1e-8)
    iou.fill_diagonal_(0)
    assert float(iou.max()) <= 0.5 + 1e-6           # no surviving pair overlaps more than the threshold


def _backbone_stage_outputs(model, batch):
    """Features after every stage of BiResNet (forward hooks on its direct children), keyed by module name."""
    feats, hooks = {}, []
    for name, mod in model.backbone_3d.named_children():
        def hook(m, inp, out, name=name):
            t = out.F if hasattr(out, "F") else (out["sp_tensor"].F if isinstance(out, dict) and "sp_tensor" in out else None)
            if t is not None:
                feats[name] = t.detach().float().clone()
        hooks.append(mod.register_forward_hook(hook))
    try:
        ret, tb, _ = model(batch)
    finally:
        for h in hooks:
            h.remove()
    return feats, ret, tb


def _backbone_functional_grads(model, dev):
    """Gradients of a fixed linear functional of the backbone output (no vote quantisation, class-map membership or NMS in
    the path): {parameter name: gradient}."""
    model.zero_grad()
    batch = build_model.synthetic_batch("S5k", 2, device=dev)
    batch["points"][:, -3:] = batch["points"][:, -3:] / 255.
    sp = model.voxelization(batch["points"])
    out = model.backbone_3d({"sp_tensor": sp, "batch_size": 2})["sp_tensor"]
    proj = torch.randn(out.F.shape, generator=torch.Generator().manual_seed(3)).to(dev)
    (out.F * proj).sum().backward()
    return {n: p.grad.detach().cpu().clone() for n, p in model.backbone_3d.named_parameters() if p.grad is not None}


def _rel_l2(g1, g0):
    num = sum(float((g1[n] - g0[n]).pow(2).sum()) for n in g0)
    den = sum(float(g0[n].pow(2).sum()) for n in g0)
    return (num / den) ** 0.5


def test_bf16_per_stage_features_and_backbone_gradients(oracle, hip):
    """SURVEY 8(d) bf16 tolerance, stage by stage: every BiResNet stage output within 2e-2 of its scale, the end-to-end loss
    within 1 %.

    Gradients: bf16 vs fp32 differ by ~0.2 in relative L2 even for a fixed linear functional of the backbone output (no
    vote quantisation / class membership / NMS in the path), i.e. the deviation of the full step (0.16, test above) IS
    operand rounding -- ~60 convolutions in series, each rounding its input rows and output gradients to 8 mantissa bits,
    through BatchNorm backward passes of an untrained net whose parameter gradients are sums with heavy cancellation.
    It is a property of the arithmetic, not of the kernels: the CPU oracle's independent bit-level emulation of the same
    rounding deviates from ITS fp32 run by the same amount, and the HIP bf16 gradients agree with the oracle's bf16
    gradients to 1e-2."""
    res = []
    for lib, dev, prec in ((hip, "cuda", 0), (hip, "cuda", 1), (oracle, "cpu", 0), (oracle, "cpu", 1)):
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        model.dense_head.force_gt_selection = True
        model.dense_head.force_class_logit_boost = 6.0
        model = model.to(dev).train()
        me.PRECISION = prec
        try:
            with _lib.use_library(lib):
                feats = tb = None
                if dev == "cuda":
                    torch.manual_seed(1)
                    np.random.seed(1)
                    feats, ret, tb = _backbone_stage_outputs(model, build_model.synthetic_batch("S5k", 2, device=dev))
                grads = _backbone_functional_grads(model, dev)
        finally:
            me.PRECISION = 0
        res.append((feats, tb, grads))
    (f0, tb0, g_hip32), (f1, tb1, g_hip16), (_, _, g_or32), (_, _, g_or16) = res
    assert len(f0) >= 8, sorted(f0)
    worst = {}
    for name in f0:
        if f0[name].shape == f1[name].shape:
            worst[name] = float((f1[name] - f0[name]).abs().max()) / max(float(f0[name].abs().max()), 1e-6)
    assert max(worst.values()) <= 2e-2, worst
    assert abs(tb0["loss_all"] - tb1["loss_all"]) <= 1e-2 * abs(tb0["loss_all"]), (tb0["loss_all"], tb1["loss_all"])
    dev_hip, dev_or = _rel_l2(g_hip16, g_hip32), _rel_l2(g_or16, g_or32)
    fp32_pair, bf16_pair = _rel_l2(g_hip32, g_or32), _rel_l2(g_hip16, g_or16)
    report = dict(hip_bf16_vs_fp32=dev_hip, oracle_bf16_vs_fp32=dev_or, hip_vs_oracle_fp32=fp32_pair, hip_vs_oracle_bf16=bf16_pair)
    print("backbone gradient deviations (relative L2):", report)
    # fp32: the two implementations differ by summation order only (1e-7 per operation) -- and the gradients already by
    # 3e-3: the net amplifies a perturbation ~10^4-fold on its way back through ~60 layers and their BatchNorms
    assert fp32_pair < 5e-3, report          # measured 2.3e-3
    assert dev_hip < 0.35 and dev_or < 0.35 and abs(dev_hip - dev_or) < 0.08, report
    assert bf16_pair < max(dev_hip, dev_or), report              # two bf16 implementations are closer to each other than to fp32


@pytest.mark.parametrize("dataset,cfgname,bs", [("sunrgbd", "S100k-yaw", 8), ("scannet", "S200k", 4)])
def test_full_size_training_step_of_the_other_configs(hip, dataset, cfgname, bs):
    """BASELINE.json configs[3] (SUN RGB-D shaped: 8 single-view 100k-point scenes with yaw, 10 classes) and configs[4]
    (200k points at 0.01 m): one full training step in the bench precision at full size -- finite loss and gradients --
    plus size-independent properties of the structures it builds (voxel rows == distinct voxels, rows in (batch, Morton)
    order, every strided map a subset lattice of its parent, proposals inside the scene)."""
    from cagroup3d_amd.pcdet.config import cfg_from_yaml_file  # noqa: F401
    model, cfg = build_model.build_cagroup3d(dataset, seed=0, voxel_size=build_model.VOXEL_SIZE_OF_CONFIG.get(cfgname))
    assert float(model.voxel_size) == (0.01 if cfgname == "S200k" else 0.02)
    model.dense_head.force_gt_selection = True
    model.dense_head.force_class_logit_boost = 6.0
    model = model.cuda().train()
    me.PRECISION = 1
    try:
        with _lib.use_library(hip):
            batch = build_model.synthetic_batch(cfgname, bs, device="cuda")
            n_pts = batch["points"].shape[0]
            ret, tb, _ = model(batch)
            ret["loss"].backward()
            torch.cuda.synchronize()
    finally:
        me.PRECISION = 0
    assert np.isfinite(tb["loss_all"]) and tb["loss_all"] > 0
    gn = torch.stack([p.grad.float().norm() for p in model.parameters() if p.grad is not None])
    assert torch.isfinite(gn).all() and float(gn.sum()) > 0
    assert n_pts == bs * int(cfgname[1:4]) * 1000
    from util import morton_keys
    vs = float(model.voxel_size)
    with _lib.use_library(hip):
        vox = model.voxelization(batch["points"])                   # the input map of the step (batch["sp_tensor"] is the backbone's OUTPUT by now)
    C = vox.C.cpu().numpy()
    keys = morton_keys(C)
    assert (np.diff(keys.astype(np.float64)) > 0).all(), "distinct voxels in (batch, Morton) order"
    pts = batch["points"][:, :4]
    ref = torch.unique(torch.cat([pts[:, :1], torch.floor(pts[:, 1:4] / vs)], 1).int(), dim=0)
    assert C.shape[0] == ref.shape[0], (C.shape[0], ref.shape[0])  # both divisions run on the device: the same voxels
    # the backbone's output map (stride 2): exactly the parent voxels' cells of the coarser lattice
    out_C = batch["sp_tensor"].C
    assert (out_C[:, 1:] % 2 == 0).all()
    parent = torch.unique(torch.cat([ref[:, :1], torch.div(ref[:, 1:], 2, rounding_mode="floor") * 2], 1), dim=0)
    assert out_C.shape[0] == parent.shape[0], (out_C.shape[0], parent.shape[0])
    out = batch["middle_feature_list"][3]
    assert (out.C[:, 1:] % 2 == 0).all() and out.F.shape[1] == 64
    boxes = torch.cat([p[0] for p in batch["pred_bbox_list"]])
    assert torch.isfinite(boxes).all() and boxes.shape[1] == 7
