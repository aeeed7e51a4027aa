"""The path the benchmark TIMES against the CPU oracle's arithmetic (round 6; VERDICT round 5, "the timed path is only self-compared").

Every kernel is compared with the oracle one call at a time elsewhere (tests/test_hip_parity.py, test_full_size_parity.py).  Here
the launch PROGRAMS are: the backbone's forward + backward tables in the bench precision -- bf16 MFMA operands, activations and
activation gradients stored as bf16 rows, tile plans and fragment-ordered weights out of the step's weight arena, on the device
additionally spread over its lanes (several HIP streams with event edges) -- and the class branches' tables in the split
precision of the heads.  The oracle executes the SAME tables through its own entry points (`_lib.device_path`: a second handle
of liboracle.so that follows the device library's kernel selection; oracle/oracle_program.c walks the rows, oracle_tile.c /
oracle_conv.c / oracle_sparse.c restate the bf16 operands, the bf16 row storage and the split operands bit-level), so a wrong
event edge, a stale region base or a kernel that reads the wrong copy shows up as a difference from a sequential CPU run.

What the comparison can resolve.  The net is untrained and BatchNorm-heavy: a last-bit difference in an fp32 sum flips a bf16
rounding of a stored activation now and then (1 ulp = 4e-3 relative), and 70 layers amplify that.  The device itself differs from
run to run (fp32 atomics in the statistics, the pair kernels and the split contractions), so every bound below is
`K x (device run-to-run difference) + floor`, layer by layer: the oracle may not be further from a device run than device runs
are from each other, within a floor that a missing contribution (an O(1) error in this measure) exceeds by two orders of magnitude.
"""
import os

import pytest
import torch

from cagroup3d_amd import _lib, build_model, engine, me

from test_engine import _backbone_step, _class_branch_inputs, _class_branch_step, _l2


@pytest.fixture(autouse=True)
def _sixteen_threads():
    """The oracle's OpenMP loops anti-scale on a many-core host (256 hardware threads on the GPU box: 2.1 s per step at 16 threads,
    14 s at 128 -- profiles/r03_cpu_thread_scaling.txt; bench.py's cpu_baseline leg caps them the same way)."""
    import ctypes
    n = min(os.cpu_count() or 1, 16)
    keep = torch.get_num_threads()
    torch.set_num_threads(n)
    gomp = None
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        kept = gomp.omp_get_max_threads()
        gomp.omp_set_num_threads(n)
    except OSError:
        pass
    yield
    torch.set_num_threads(keep)
    if gomp is not None:
        gomp.omp_set_num_threads(kept)


@pytest.fixture(scope="module")
def oracle_dev(oracle):
    """A second handle of the oracle that takes the device library's kernel choices (the session's `oracle` stays as it is)."""
    return _lib.device_path(_lib.bind(oracle.path))


def _oracle_backbone(oracle_dev, model, batch, state, act16):
    """Forward + backward of the backbone's launch program on the oracle in the bench precision; returns _backbone_step's tuple."""
    keep = (me.PRECISION, engine.ACT_BF16)
    with _lib.use_library(oracle_dev):
        me.PRECISION, engine.ACT_BF16 = 1, act16
        try:
            got = None
            for _ in range(3):           # the first pass records the weights in the step's arena (engine.NotReady -> per-layer path)
                before = engine.STATS["program_passes"]
                model.load_state_dict(state)
                got = _backbone_step(model, batch, True, "cpu")
                if engine.STATS["program_passes"] == before + 1:
                    return got
            raise AssertionError("the oracle never ran the launch program: %r" % (engine.STATS,))
        finally:
            me.PRECISION, engine.ACT_BF16 = keep
            me._WeightPlan.reset()


def test_the_oracle_runs_the_bench_precision_program(oracle_dev):
    """CPU: the oracle executes the backbone's tables in the bench precision (tile plans, fragment weights, the step's weight
    arena).  With fp32 row storage the program equals the oracle's per-layer path (same calls, same operands, sequential sums);
    bf16 row storage moves the result by no more than rounded activations can (a few 1e-2 through 70 untrained layers)."""
    model, _ = build_model.build_cagroup3d("scannet", seed=0)
    batch = build_model.synthetic_batch("S2k", 1, device="cpu")
    state = {k: v.clone() for k, v in model.state_dict().items()}
    keep = (me.PRECISION, engine.ACT_BF16)
    with _lib.use_library(oracle_dev):
        me.PRECISION, engine.ACT_BF16 = 1, False
        try:
            # pass 1: nothing is in the weight arena yet -> engine.NotReady -> the per-layer path of me.py (the specification of the
            # fp32-row program), which records the weights; pass 2: the program
            before = dict(engine.STATS)
            ref = _backbone_step(model, batch, True, "cpu")
            assert engine.STATS["program_passes"] == before["program_passes"] and engine.STATS["not_ready"] == before["not_ready"] + 1
            model.load_state_dict(state)
            p32 = _backbone_step(model, batch, True, "cpu")
            assert engine.STATS["program_passes"] == before["program_passes"] + 1
            engine.ACT_BF16 = True
            model.load_state_dict(state)
            p16 = _backbone_step(model, batch, True, "cpu")
            assert engine.STATS["program_passes"] == before["program_passes"] + 2
        finally:
            me.PRECISION, engine.ACT_BF16 = keep
            me._WeightPlan.reset()
    assert torch.equal(ref[0], p32[0]) and torch.equal(ref[0], p16[0])
    assert _l2(ref[1], p32[1]) <= 2e-3, _l2(ref[1], p32[1])
    for k in ref[3]:
        assert _l2(ref[3][k].float(), p32[3][k].float()) <= 2e-3, k
    bad = {k: _l2(ref[2][k], p32[2][k]) for k in ref[2] if float(ref[2][k].norm()) > 1e-3 and _l2(ref[2][k], p32[2][k]) > 2e-2}
    assert not bad, bad
    assert _l2(p32[1], p16[1]) <= 1e-1, _l2(p32[1], p16[1])
    for k in ref[3]:
        if "running_" in k:
            assert _l2(p32[3][k].float(), p16[3][k].float()) <= 5e-2, k


def test_the_weight_plan_does_not_outlive_its_library(oracle, oracle_dev):
    """The step's weight plan (recorded weights + their arena) belongs to one library's memory: changing the active library resets
    it (`_lib.on_switch`).  It did not until round 6, and a plan that still listed the previous test's DEVICE weights handed the
    oracle a conversion table of device addresses."""
    w = torch.randn(27, 64, 64)
    with _lib.use_library(oracle_dev):
        keep, me.PRECISION = me.PRECISION, 1
        try:
            assert me._planned_single(w, False, True) is None and me._WeightPlan.singles       # recorded, converted from the next forward on
            gen = me._WeightPlan.gen
            with _lib.use_library(oracle):                                                      # another library: nothing carried over
                assert not me._WeightPlan.singles and me._WeightPlan.gen > gen
            assert not me._WeightPlan.singles                                                   # ... nor back
        finally:
            me.PRECISION = keep
    with _lib.use_library(oracle_dev):
        with _lib.use_library(oracle_dev):                                                      # the same library again: no reset
            me._WeightPlan.singles[("probe", 0)] = None
        assert ("probe", 0) in me._WeightPlan.singles
    me._WeightPlan.reset()


def _device_backbone(model_cuda, batch_cuda, state, runs=2):
    keep = me.PRECISION
    me.PRECISION = 1
    try:
        for e in (True, True):           # weights into the arena, programs compiled, lanes tuned
            model_cuda.load_state_dict(state)
            _backbone_step(model_cuda, batch_cuda, e, "cuda")
        out = []
        for _ in range(runs):
            model_cuda.load_state_dict(state)
            before = engine.STATS["program_passes"]
            out.append(_backbone_step(model_cuda, batch_cuda, True, "cuda"))
            assert engine.STATS["program_passes"] == before + 1, "the engine path did not run on the device"
        return out
    finally:
        me.PRECISION = keep


@pytest.mark.gpu
@pytest.mark.parametrize("cfgname,n", [("S5k", 2), (os.environ.get("CG3D_TIMED_PATH_CFG", "S50k"), 1)])
def test_backbone_program_on_lanes_with_bf16_rows_follows_the_oracle(oracle_dev, hip, cfgname, n):
    """The timed backbone pass (lanes on, bf16 row storage, bench precision) on the device against the same tables on the oracle:
    per-layer forward statistics (every BatchNorm's running mean / variance after the step: one check per convolution of the
    forward pass), the output rows, the `loss` the test backpropagates, and every parameter gradient."""
    assert engine.ACT_BF16 and engine.LANES, "this test is about the default (timed) configuration"
    model, _ = build_model.build_cagroup3d("scannet", seed=0)
    batch = build_model.synthetic_batch(cfgname, n, device="cpu")
    state = {k: v.clone() for k, v in model.state_dict().items()}
    import time
    t0 = time.time()
    ref = _oracle_backbone(oracle_dev, model, batch, state, True)
    print("%s x %d: oracle passes %.1f s" % (cfgname, n, time.time() - t0))
    model = model.cuda()
    dstate = {k: v.cuda() for k, v in state.items()}
    d1, d2 = _device_backbone(model, {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in batch.items()}, dstate)
    assert torch.equal(ref[0], d1[0]), "voxel rows differ"
    # forward, layer by layer
    worst = (0.0, None)
    for k in ref[3]:
        if "running_" not in k:
            continue
        noise, err = _l2(d1[3][k].float(), d2[3][k].float()), _l2(ref[3][k].float(), d1[3][k].float())
        worst = max(worst, (err, k))
        assert err <= 4 * noise + 2e-2, (k, err, noise)
    out_noise, out_err = _l2(d1[1], d2[1]), _l2(ref[1], d1[1])
    print("%s x %d: forward statistics worst %.2e (%s); output rows oracle-device %.2e, device-device %.2e" % (cfgname, n, worst[0], worst[1], out_err, out_noise))
    assert out_err <= 4 * out_noise + 2e-2, (out_err, out_noise)
    # the scalar the backward starts from (same fixed upstream gradient on both sides)
    g = torch.Generator().manual_seed(5)
    up = torch.randn(ref[1].shape, generator=g)
    lo, ld = float((ref[1] * up).sum()), float((d1[1] * up).sum())
    scale = float((ref[1].abs() * up.abs()).sum())
    assert abs(lo - ld) <= 1e-2 * scale, (lo, ld, scale)
    # backward: every parameter gradient
    assert set(ref[2]) == set(d1[2]) and len(ref[2]) > 100
    bad, worst_g = {}, (0.0, None)
    for k in ref[2]:
        if float(ref[2][k].norm()) > 1e-3:
            noise, err = _l2(d1[2][k], d2[2][k]), _l2(ref[2][k], d1[2][k])
            worst_g = max(worst_g, (err, k, noise))
            if err > 4 * noise + 5e-2:
                bad[k] = (err, noise)
    print("   parameter gradients: worst oracle-device %.2e (%s; device-device %.2e)" % worst_g)
    assert not bad, bad


@pytest.mark.gpu
def test_class_branch_program_in_the_split_precision_follows_the_fp32_oracle(oracle, hip):
    """The class branches' launch program as the benchmark runs it (split operands: fp32-accurate products from three bf16 passes)
    on the device against the oracle's per-layer path in FP32 arithmetic (sequential sums): four grouped convolution + BatchNorm
    + ELU stages of all 18 classes -- output rows, input gradient, every parameter gradient, running statistics at 1e-3."""
    model, _ = build_model.build_cagroup3d("scannet", seed=0)
    head = model.dense_head.train()
    state = {k: v.clone() for k, v in head.state_dict().items()}
    fine, coarse, feat, up = _class_branch_inputs(head, "cpu", base=60)
    prec = me.PRECISION
    with _lib.use_library(oracle):
        me.PRECISION = 0                 # (the class program has no fp32 form: the reference is the per-layer path in fp32 arithmetic)
        try:
            ref = _class_branch_step(head, fine, coarse, feat, up, False, 2)
        finally:
            me.PRECISION = prec
    head = head.cuda()
    dstate = {k: v.cuda() for k, v in state.items()}
    me.PRECISION = me.PREC_SPLIT
    try:
        got = None
        for i in range(3):               # weight variants into the arena first
            head.load_state_dict(dstate)
            before = engine.CLASS_STATS["program_passes"]
            got = _class_branch_step(head, fine.cuda(), coarse.cuda(), feat.cuda(), up.cuda(), True, 2)
        assert engine.CLASS_STATS["program_passes"] == before + 1
    finally:
        me.PRECISION = prec
        me._WeightPlan.reset()
    assert ref[0].shape == got[0].shape
    assert _l2(ref[0], got[0]) <= 1e-3, _l2(ref[0], got[0])
    assert _l2(ref[1], got[1]) <= 1e-3, _l2(ref[1], got[1])
    bad = {k: _l2(ref[2][k], got[2][k]) for k in ref[2] if float(ref[2][k].norm()) > 1e-3 and _l2(ref[2][k], got[2][k]) > 2e-3}
    assert not bad, bad
    for k in ref[3]:
        assert _l2(ref[3][k].float(), got[3][k].float()) <= 1e-3, k
