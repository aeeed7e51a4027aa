"""The backbone's launch program (cagroup3d_amd/engine.py, include/cagroup3d_program.h) against the per-layer path it
replaces: same output rows, same gradient for every parameter, same running statistics.  CPU: both paths on the oracle in the
fp32 parity mode (every product through the generic pair kernels); gpu: both paths on the HIP library in the bench precision."""
import os

import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, build_model, engine, me


def _backbone_step(model, batch, use_engine, dev):
    """One forward + backward of the backbone alone (a fixed random upstream gradient); returns what both paths must agree on."""
    net = model.backbone_3d
    net.train()
    for p in net.parameters():
        p.grad = None
    os.environ["CG3D_ENGINE_ANY"] = "1"
    engine.ENABLED = bool(use_engine)
    try:
        pts = batch["points"].clone()
        pts[:, -3:] = pts[:, -3:] / 255.
        sp = model.voxelization(pts)
        me.prepare_weights(True)
        out = net({"sp_tensor": sp, "batch_size": batch["batch_size"]})["sp_tensor"]
        me.finish_weights()
        g = torch.Generator().manual_seed(5)
        up = torch.randn(out.F.shape, generator=g).to(dev)
        (out.F * up).sum().backward()
    finally:
        engine.ENABLED = True
        os.environ.pop("CG3D_ENGINE_ANY", None)
    grads = {n: p.grad.detach().clone().cpu() for n, p in net.named_parameters() if p.grad is not None}
    bufs = {n: b.detach().clone().cpu() for n, b in net.named_buffers()}
    return out.C.cpu(), out.F.detach().cpu(), grads, bufs


def _l2(a, b):
    return float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30))


def _compare(a, b, tol):
    """Relative L2 error per tensor (an untrained BatchNorm-heavy net amplifies rounding differences element-wise; a missing
    or doubled gradient contribution shows as an O(1) error in this measure)."""
    assert torch.equal(a[0], b[0])
    assert _l2(a[1], b[1]) <= tol, _l2(a[1], b[1])
    assert set(a[2]) == set(b[2]) and len(a[2]) > 100
    bad = {k: _l2(a[2][k], b[2][k]) for k in a[2] if float(a[2][k].norm()) > 1e-3 and _l2(a[2][k], b[2][k]) > tol}
    assert not bad, bad
    for k in a[3]:
        assert _l2(a[3][k].float(), b[3][k].float()) <= tol, k


def test_program_equals_per_layer_path_on_the_oracle(oracle):
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            batch = build_model.synthetic_batch("S5k", 2, device="cpu")
            state = {k: v.clone() for k, v in model.state_dict().items()}
            # the per-layer path's 1x1x1 products through the pair kernel as well (its CPU default is the library GEMM): with
            # the same arithmetic on both sides the two paths must agree to rounding of the few re-ordered additions -- a
            # sharp test, where a library-vs-kernel rounding difference is amplified to 1e-3 by this untrained net
            skinny = me.LinearFunction._skinny
            me.LinearFunction._skinny = staticmethod(lambda n, a, b: True)
            try:
                ref = _backbone_step(model, batch, False, "cpu")
            finally:
                me.LinearFunction._skinny = skinny
            model.load_state_dict(state)                     # (the running statistics moved)
            seen = []
            run = engine.run_backbone
            engine.run_backbone = lambda *a, **k: (seen.append(1), run(*a, **k))[1]
            try:
                got = _backbone_step(model, batch, True, "cpu")
            finally:
                engine.run_backbone = run
            assert seen, "the engine path did not run"
        finally:
            me.PRECISION = prec
    _compare(ref, got, 1e-5)


def test_program_table_is_plain_data(oracle):
    """A compiled pass is two int64 tables + region sizes: rows of CG3D_PROG_STRIDE, known opcodes, region tags only in 1..6."""
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        os.environ["CG3D_ENGINE_ANY"] = "1"
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            model.train()
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            pts = batch["points"].clone()
            sp = model.voxelization(pts)
            comp = engine.compile_backbone(model.backbone_3d, sp, mid_mark=True)
        finally:
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)
    for tab in (comp.fwd, comp.bwd):
        assert tab.dtype == np.int64 and tab.shape[1] == engine.STRIDE and tab.shape[0] > 100
        op, lane = tab[:, 0] & engine.OPCODE_MASK, tab[:, 0] >> engine.LANE_SHIFT
        assert ((op > 0) & (op <= engine.OP_EVENT_WAIT)).all() and int(lane.max()) < engine.MAX_LANES
        assert int((tab >> engine.TAG).max()) <= 6 and int(tab.min()) >= 0
    assert 0 < comp.marks["mid"] < comp.bwd.shape[0]
    assert comp.size[engine.R_PG] >= 4 * sum(p.numel() for p in model.backbone_3d.parameters())


def test_a_program_node_refuses_a_second_backward(oracle):
    """The node's backward releases the arena its rows point into; nothing is saved through save_for_backward, so autograd
    would let a second backward (retain_graph=True) re-run the table on freed memory.  It must raise instead."""
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        os.environ["CG3D_ENGINE_ANY"] = "1"
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            model.train()
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            pts = batch["points"].clone()
            pts[:, -3:] = pts[:, -3:] / 255.
            sp = model.voxelization(pts)
            out = engine.run_backbone(model.backbone_3d, sp)
            loss = out.F.sum()
            loss.backward(retain_graph=True)
            first = {n: p.grad.clone() for n, p in model.backbone_3d.named_parameters() if p.grad is not None}
            with pytest.raises(RuntimeError, match="backward already ran"):
                loss.backward()
            # ... and the first pass's gradients are untouched by the refused one
            for n, p in model.backbone_3d.named_parameters():
                if n in first:
                    assert torch.equal(p.grad, first[n]), n
        finally:
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)


def test_frozen_parameters_get_no_gradient_from_a_program(oracle):
    """requires_grad = False on a layer: the per-layer autograd path leaves its .grad None; so must the program's backward
    (clip_grad_norm_ over model.parameters() would otherwise see gradients the optimizer never asked for)."""
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        os.environ["CG3D_ENGINE_ANY"] = "1"
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            model.train()
            net = model.backbone_3d
            frozen = [p for n, p in net.named_parameters() if n.startswith("layer1.")]
            assert frozen
            for p in frozen:
                p.requires_grad_(False)
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            pts = batch["points"].clone()
            pts[:, -3:] = pts[:, -3:] / 255.
            out = engine.run_backbone(net, model.voxelization(pts))
            out.F.sum().backward()
            assert all(p.grad is None for p in frozen)
            assert sum(p.grad is not None for p in net.parameters()) > 100
        finally:
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)


@pytest.mark.gpu
def test_a_program_is_not_run_on_a_stale_weight_arena(hip):
    """Detector step, optimizer step, then the backbone ALONE (no prepare_weights: the arena still holds the pre-step bf16
    weights).  The program must step aside (NotReady -> per-layer path, which converts on the spot), so the output follows
    the UPDATED weights."""
    prec, me.PRECISION = me.PRECISION, 1
    try:
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        model = model.cuda().train()
        net = model.backbone_3d
        batch = build_model.synthetic_batch("S5k", 2, device="cuda")
        for e in (False, False, True, True):
            _backbone_step(model, batch, e, "cuda")                    # every weight variant enters the arena
        pts = batch["points"].clone()
        pts[:, -3:] = pts[:, -3:] / 255.
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.5)                                            # "optimizer.step()": in place, version-blind as fused AdamW
        os.environ["CG3D_ENGINE_ANY"] = "1"
        before = dict(engine.STATS)
        assert not me._WeightPlan.live
        out_prog = net({"sp_tensor": model.voxelization(pts.clone()), "batch_size": 2})["sp_tensor"].F.detach().clone()
        assert engine.STATS["not_ready"] == before["not_ready"] + 1 and engine.STATS["program_passes"] == before["program_passes"]
        engine.ENABLED = False
        out_ref = net({"sp_tensor": model.voxelization(pts.clone()), "batch_size": 2})["sp_tensor"].F.detach().clone()
        engine.ENABLED = True
        assert _l2(out_prog, out_ref) < 0.05, _l2(out_prog, out_ref)   # (atomics' run-to-run noise; stale weights are O(1) off)
    finally:
        engine.ENABLED = True
        os.environ.pop("CG3D_ENGINE_ANY", None)
        me.PRECISION = prec


@pytest.mark.gpu
@pytest.mark.parametrize("cfgname,n", [("S5k", 2), ("S50k", 1)])
def test_program_equals_per_layer_path_on_the_device(hip, cfgname, n):
    prec, me.PRECISION = me.PRECISION, 1
    # (the per-layer path stores fp32 rows: it is the specification of the program with fp32 row storage; the bf16-storage
    # program is compared with THAT program in tests/test_act_bf16.py)
    act16, engine.ACT_BF16 = engine.ACT_BF16, False
    try:
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        model = model.cuda()
        batch = build_model.synthetic_batch(cfgname, n, device="cuda")
        state = {k: v.clone() for k, v in model.state_dict().items()}
        # warm-up passes of both paths: the weights (every variant either path asks for) enter the step's bf16 arena
        for e in (False, False, True, True):
            model.load_state_dict(state)
            _backbone_step(model, batch, e, "cuda")

        def run(use_engine):
            model.load_state_dict(state)
            return _backbone_step(model, batch, use_engine, "cuda")
        ref, ref2 = run(False), run(False)
        before = engine.STATS["program_passes"]
        got = run(True)
        assert engine.STATS["program_passes"] == before + 1, "the engine path did not run on the device"
    finally:
        me.PRECISION = prec
        engine.ACT_BF16 = act16
    # Same kernels, same operands: what differs is the order of fp32 atomic additions -- which this untrained BatchNorm-heavy
    # net amplifies enormously (two runs of the per-layer path itself differ by up to 0.2 in relative L2 on the small scenes).
    # The yardstick is therefore that run-to-run noise: the program may deviate from a per-layer run by no more than a
    # per-layer run deviates from another one (x 3), tensor by tensor.
    assert torch.equal(ref[0], got[0])
    assert _l2(ref[1], got[1]) <= 3 * _l2(ref[1], ref2[1]) + 1e-4
    bad = {}
    for k in ref[2]:
        if float(ref[2][k].norm()) > 1e-3:
            noise, err = _l2(ref[2][k], ref2[2][k]), _l2(ref[2][k], got[2][k])
            if err > 3 * noise + 1e-3:
                bad[k] = (err, noise)
    assert not bad, bad
    for k in ref[3]:
        assert _l2(ref[3][k].float(), got[3][k].float()) <= 3 * _l2(ref[3][k].float(), ref2[3][k].float()) + 1e-3, k


def test_compiled_program_owns_the_cached_tables_it_points_into(oracle):
    """me.py's host caches (chunk tables, identity pair lists, unit BatchNorm rows) are cleared when they grow -- in a training
    run with varying batches within tens of steps.  A program holds raw addresses into those tables, so it must hold the
    tensors themselves: every address in its rows that falls into a cached table must fall into a tensor of `comp.keep`."""
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        os.environ["CG3D_ENGINE_ANY"] = "1"
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            model.train()
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            sp = model.voxelization(batch["points"].clone())
            me._chunk_cache.clear(); me._ident_cache.clear()
            comp = engine.compile_backbone(model.backbone_3d, sp, mid_mark=False)
            cached = []
            me._walk_tensors = me._walk_tensors         # (same helper the stream hand-over uses)

            def leaves(o, out):
                if torch.is_tensor(o):
                    out.append(o)
                elif isinstance(o, (list, tuple)):
                    for v in o:
                        leaves(v, out)
            leaves(list(me._chunk_cache.values()) + list(me._ident_cache.values()), cached)
            kept = []
            leaves([comp.keep, comp.keep_cached], kept)
        finally:
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)
    assert len(cached) > 10
    kept_ptrs = {t.data_ptr() for t in kept}
    addrs = set(int(v) for v in np.concatenate([comp.fwd.reshape(-1), comp.bwd.reshape(-1)]))
    used = [t for t in cached if t.numel() > 0 and t.data_ptr() in addrs]
    assert len(used) > 10, "the program does not seem to use the cached tables at all"
    missing = [t.shape for t in used if t.data_ptr() not in kept_ptrs]
    assert not missing, "cached tables referenced by raw address only: %s" % missing[:5]


# ------------------------------------------------------------------------------------------------ the class branches
def _class_branch_step(head, fine, coarse, feat, up, use_program, B):
    """The feature side of all class branches (four grouped convolution + BatchNorm + ELU stages) on given coordinates, through the
    per-layer path or through the launch program; returns what both must agree on."""
    C, dev = head.n_classes, feat.device
    for p in head.parameters():
        p.grad = None
    x = feat.clone().requires_grad_(True)
    avg = me.SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE
    me.prepare_weights(True)
    cls_map = me.SparseTensor(coordinates=fine, features=x, quantization_mode=avg)
    cls_exp = me.SparseTensor(coordinates=coarse, features=x, tensor_stride=head.expand, quantization_mode=avg)
    fb, cb = cls_map.C[:, 0].long(), cls_exp.C[:, 0].long()
    fine_bounds = (0,) + tuple(np.cumsum(torch.bincount(fb // B, minlength=C).cpu().numpy()).tolist())
    coarse_bounds = (0,) + tuple(np.cumsum(torch.bincount(cb // B, minlength=C).cpu().numpy()).tolist())
    mgr, emgr = cls_map.coordinate_manager, cls_exp.coordinate_manager
    km9 = mgr.kernel_map(cls_map.coordinate_map_key, cls_map.coordinate_map_key, head.cls_kernel, 1, False)
    km5 = emgr.kernel_map(cls_exp.coordinate_map_key, cls_exp.coordinate_map_key, 5, 1, False)
    tgt_key, _, _ = emgr.insert(cls_map.C, 1)
    km_up = emgr.kernel_map(cls_exp.coordinate_map_key, tgt_key, head.expand, 1, True)
    ident = me.KernelMap.identity(cls_map.C.shape[0], dev)
    elu = torch.nn.functional.elu
    if use_program:
        assert engine.class_branches_applicable(head)
        f = engine.run_class_branches(head, cls_map.F, cls_exp.F, km9, km5, km_up, ident, fine_bounds, coarse_bounds)
    else:
        a = me.grouped_conv(cls_map.F, [m[0].kernel for m in head.cls_individual_out], km9, fine_bounds, closed=True)
        a = head._grouped_bn_act(a, fine_bounds, [m[1] for m in head.cls_individual_out], elu)
        e = me.grouped_conv(cls_exp.F, [m[0].kernel for m in head.cls_individual_expand_out], km5, coarse_bounds, closed=True)
        e = head._grouped_bn_act(e, coarse_bounds, [m[1] for m in head.cls_individual_expand_out], elu)
        u = me.grouped_conv(e, [m[0].kernel for m in head.cls_individual_up], km_up, fine_bounds)
        u = head._grouped_bn_act(u, fine_bounds, [m[1][0] for m in head.cls_individual_up], elu)
        f = me.grouped_conv(torch.cat([u, a], dim=1), [m[0].kernel for m in head.cls_individual_fuse], ident, fine_bounds)
        f = head._grouped_bn_act(f, fine_bounds, [m[1] for m in head.cls_individual_fuse], elu)
    me.finish_weights()
    fout = f.detach().cpu()
    loss = (f * up[:f.shape[0]]).sum()
    # (the head's frame is gone when backward runs: the graph alone must keep alive what the backward pass reads)
    del cls_map, cls_exp, mgr, emgr, km9, km5, km_up, ident, tgt_key, f, fb, cb
    import gc
    gc.collect()
    if feat.is_cuda:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        scratch = torch.full((64 << 20,), float("nan"), device=dev)      # what the allocator hands out next is poisoned
        del scratch
    loss.backward()
    names = ("cls_individual_out", "cls_individual_expand_out", "cls_individual_up", "cls_individual_fuse")
    grads = {n: p.grad.detach().clone().cpu() for n, p in head.named_parameters() if n.startswith(names)}
    bufs = {n: b.detach().clone().cpu() for n, b in head.named_buffers() if n.startswith(names)}
    return fout, x.grad.detach().cpu(), grads, bufs


def _class_branch_inputs(head, dev, B=2, base=150):
    C = head.n_classes
    g = torch.Generator().manual_seed(3)
    rows = []
    for c in range(C):
        for b in range(B):
            n = base + 40 * ((c + b) % 5)
            # a surface patch (most rows have in-plane neighbours) with a few duplicate voxels for the averaging quantisation
            xy = torch.randint(0, 24, (n, 2), generator=g)
            z = torch.randint(0, 3, (n, 1), generator=g)
            rows.append(torch.cat([torch.full((n, 1), c * B + b), xy, z], 1))
    fine = torch.cat(rows).float().to(dev)
    coarse = fine.clone()
    coarse[:, 1:] = torch.floor(fine[:, 1:] / head.expand) * head.expand
    ch = head.cls_individual_out[0][0].kernel.shape[-1]
    feat = torch.randn(fine.shape[0], ch, generator=g).to(dev)
    up = torch.randn(fine.shape[0], ch, generator=g).to(dev)
    return fine, coarse, feat, up


def test_class_branch_program_equals_per_layer_path_on_the_oracle(oracle):
    """CPU: both paths on the oracle library in the bf16-operand mode (sequential sums: the two paths issue the same calls with
    the same operands, so they agree to rounding of the few places where the order of additions differs)."""
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 1
        os.environ["CG3D_ENGINE_ANY"] = "1"
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            head = model.dense_head.train()
            fine, coarse, feat, up = _class_branch_inputs(head, "cpu", base=40)
            state = {k: v.clone() for k, v in head.state_dict().items()}

            def run(use_program):
                head.load_state_dict(state)
                return _class_branch_step(head, fine, coarse, feat, up, use_program, 2)
            ref, ref2 = run(False), run(False)
            before = engine.CLASS_STATS["program_passes"]
            got = run(True)
            assert engine.CLASS_STATS["program_passes"] == before + 1
        finally:
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)
            me._WeightPlan.reset()
    # On the CPU library me.grouped_conv takes the stacked-weight SparseConvFunction (pair lists throughout), the program the
    # grouped form the device takes (the 5^3 stage through cg3d_spconv_fwd_tiled): same products, another order of additions --
    # and with bf16 row copies between the stages a last-bit difference flips a rounding now and then.  Hence the absolute
    # allowances next to the (here: zero) rerun noise of the per-layer path.
    assert ref[0].shape == got[0].shape and len(ref[2]) == 18 * 12 and set(ref[2]) == set(got[2])
    print("output / input-gradient error: program %.2e / %.2e, per-layer rerun %.2e / %.2e" %
          (_l2(ref[0], got[0]), _l2(ref[1], got[1]), _l2(ref[0], ref2[0]), _l2(ref[1], ref2[1])))
    assert _l2(ref[0], got[0]) <= 3 * _l2(ref[0], ref2[0]) + 2e-4
    assert _l2(ref[1], got[1]) <= 3 * _l2(ref[1], ref2[1]) + 2e-3
    bad = {}
    for k in ref[2]:
        if float(ref[2][k].norm()) > 1e-3:
            noise, err = _l2(ref[2][k], ref2[2][k]), _l2(ref[2][k], got[2][k])
            if err > 3 * noise + 2e-3:
                bad[k] = (err, noise)
    assert not bad, bad
    for k in ref[3]:
        assert _l2(ref[3][k].float(), got[3][k].float()) <= 3 * _l2(ref[3][k].float(), ref2[3][k].float()) + 1e-4, k


@pytest.mark.gpu
def test_class_branch_program_equals_per_layer_path_on_the_device(hip):
    """engine.run_class_branches (one autograd node, one launch program each way) against the sequence of grouped autograd
    functions it replaces: same output rows, same gradient for the input rows and for every parameter of the four stages of
    all 18 classes, same running statistics.  Yardstick as for the backbone: the run-to-run noise of the per-layer path."""
    prec, me.PRECISION = me.PRECISION, 1
    try:
        model, _ = build_model.build_cagroup3d("scannet", seed=0)
        head = model.dense_head.cuda().train()
        B = 2
        fine, coarse, feat, up = _class_branch_inputs(head, "cuda")
        state = {k: v.clone() for k, v in head.state_dict().items()}

        def run(use_program):
            head.load_state_dict(state)
            return _class_branch_step(head, fine, coarse, feat, up, use_program, B)
        for e in (False, False, True, True):       # both paths' weight variants enter the step's bf16 arena
            run(e)
        ref, ref2 = run(False), run(False)
        before = engine.CLASS_STATS["program_passes"]
        got = run(True)
        assert engine.CLASS_STATS["program_passes"] == before + 1
    finally:
        me.PRECISION = prec
    assert ref[0].shape == got[0].shape and len(ref[2]) == 18 * 12 and set(ref[2]) == set(got[2])
    assert _l2(ref[0], got[0]) <= 3 * _l2(ref[0], ref2[0]) + 1e-4, (_l2(ref[0], got[0]), _l2(ref[0], ref2[0]))
    assert _l2(ref[1], got[1]) <= 3 * _l2(ref[1], ref2[1]) + 1e-3, (_l2(ref[1], got[1]), _l2(ref[1], ref2[1]))
    bad = {}
    for k in ref[2]:
        if float(ref[2][k].norm()) > 1e-3:
            noise, err = _l2(ref[2][k], ref2[2][k]), _l2(ref[2][k], got[2][k])
            if err > 3 * noise + 4e-3:      # the averaging quantisation in front of both paths sums duplicates with atomics: its own run-to-run flips reach 1.3e-3 on single kernels
                bad[k] = (err, noise)
    assert not bad, bad
    for k in ref[3]:
        assert _l2(ref[3][k].float(), got[3][k].float()) <= 3 * _l2(ref[3][k].float(), ref2[3][k].float()) + 1e-3, k


# ------------------------------------------------------------------------------------------------ the head's first layers
def _head_pre_step(head, sp_args, feat, ups, use_program):
    for p in head.parameters():
        p.grad = None
    x = feat.clone().requires_grad_(True)
    me._ROWS16.clear()
    me.prepare_weights(True)
    me.WANT_BN_STATS = True
    sp = me.SparseTensor(features=x, coordinate_map_key=sp_args[1], coordinate_manager=sp_args[0])
    if use_program:
        assert engine.head_pre_applicable(head)
        off, fo = engine.run_head_pre(head, sp)
    else:
        off, fo = head.offset_block(sp).F, head.feature_offset(sp).F
    me.finish_weights()
    ((off * ups[0]).sum() + (fo * ups[1]).sum()).backward()
    names = ("offset_block", "feature_offset")
    grads = {n: p.grad.detach().clone().cpu() for n, p in head.named_parameters() if n.startswith(names)}
    bufs = {n: b.detach().clone().cpu() for n, b in head.named_buffers() if n.startswith(names)}
    return torch.cat([off.detach().cpu(), fo.detach().cpu()], 1), x.grad.detach().cpu(), grads, bufs


@pytest.mark.gpu
@pytest.mark.parametrize("dataset,nrows", [("scannet", 20000), ("sunrgbd", 9000), ("scannet", 3000)])
def test_head_pre_program_equals_per_layer_path_on_the_device(hip, dataset, nrows):
    """engine.run_head_pre (the vote-offset block and the feature-offset block as one autograd node) against the module calls it
    replaces; 20 000 rows: the 64 -> 3 layer takes the streaming pair kernel (me.LinearFunction._skinny), 3 000: the library."""
    prec, me.PRECISION = me.PRECISION, 1
    keep_flag, engine.HEAD_PROGRAM = engine.HEAD_PROGRAM, True
    try:
        model, _ = build_model.build_cagroup3d(dataset, seed=0)
        head = model.dense_head.cuda().train()
        g = torch.Generator().manual_seed(4)
        coords = torch.cat([torch.randint(0, 2, (nrows * 2, 1), generator=g), torch.randint(0, 60, (nrows * 2, 2), generator=g),
                            torch.randint(0, 6, (nrows * 2, 1), generator=g)], 1).float().cuda()
        sp0 = me.SparseTensor(coordinates=coords, features=torch.zeros(coords.shape[0], 1, device="cuda"))
        n = sp0.C.shape[0]
        c = head.offset_block[0].kernel.shape[-2]
        feat = torch.randn(n, c, generator=g).cuda()
        n_vote = 3 if head.with_yaw else 1
        ups = (torch.randn(n, 3 * n_vote, generator=g).cuda(), torch.randn(n, c * n_vote, generator=g).cuda())
        state = {k: v.clone() for k, v in head.state_dict().items()}
        args = (sp0.coordinate_manager, sp0.coordinate_map_key)

        def run(use_program):
            head.load_state_dict(state)
            return _head_pre_step(head, args, feat, ups, use_program)
        for e in (False, False):
            run(e)
        for e in (True, True, True):               # the program's weight variants enter the arena one refusal at a time
            try:
                run(e)
            except engine.NotReady:
                pass
        ref, ref2 = run(False), run(False)
        before = engine.HEAD_STATS["program_passes"]
        got = run(True)
        assert engine.HEAD_STATS["program_passes"] == before + 1
    finally:
        me.PRECISION = prec
        engine.HEAD_PROGRAM = keep_flag
        me._WeightPlan.reset()
    assert ref[0].shape == got[0].shape and set(ref[2]) == set(got[2]) and len(ref[2]) == 10
    assert _l2(ref[0], got[0]) <= 3 * _l2(ref[0], ref2[0]) + 2e-4, (_l2(ref[0], got[0]), _l2(ref[0], ref2[0]))
    assert _l2(ref[1], got[1]) <= 3 * _l2(ref[1], ref2[1]) + 2e-3, (_l2(ref[1], got[1]), _l2(ref[1], ref2[1]))
    bad = {}
    for k in ref[2]:
        if float(ref[2][k].norm()) > 1e-3:
            noise, err = _l2(ref[2][k], ref2[2][k]), _l2(ref[2][k], got[2][k])
            if err > 3 * noise + 4e-3:
                bad[k] = (err, noise)
    assert not bad, bad
    for k in ref[3]:
        assert _l2(ref[3][k].float(), got[3][k].float()) <= 3 * _l2(ref[3][k].float(), ref2[3][k].float()) + 1e-3, k
