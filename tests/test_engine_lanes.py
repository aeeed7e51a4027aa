"""Lanes of a launch program (cagroup3d_amd/engine.py: _schedule, include/cagroup3d_program.h): the event edges between the
queues are DERIVED from what every row reads and writes, so they are checked three ways without a GPU:
  * the role table the derivation rests on against the `const` qualifiers of the entry points in include/cagroup3d_hip.h;
  * the scheduled tables of the backbone and the class branches by an independent happens-before walk (vector clocks over
    the RECORD / WAIT rows actually in the table);
  * by running the tables on the CPU oracle in OTHER orders the edges allow (side lanes as early / as late as they can go):
    every order must give bit-identical outputs and gradients -- a missing edge shows up as a row that ran before its input.
gpu: the lanes run on their own queues and must give what the one-queue run of the same tables gives.
"""
import bisect
import os
import re

import numpy as np
import pytest
import torch

from cagroup3d_amd import _lib, build_model, engine, me
from test_engine import _backbone_step, _compare

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_ENTRY = {
    engine.OP_TO_BF16: "cg3d_to_bf16", engine.OP_TILE_FWD: "cg3d_spconv_tile_fwd", engine.OP_SPCONV_FWD: "cg3d_spconv_fwd",
    engine.OP_SPCONV_FWD_TILED: "cg3d_spconv_fwd_tiled", engine.OP_PAIRS_FWD: "cg3d_spconv_pairs_fwd",
    engine.OP_PAIRS_WGRAD: "cg3d_spconv_pairs_wgrad", engine.OP_LINEAR_FWD: "cg3d_linear_fwd", engine.OP_BN_SUMS: "cg3d_bn_sums",
    engine.OP_BN_APPLY_SUMS: "cg3d_bn_apply_sums", engine.OP_BN_APPLY: "cg3d_bn_apply", engine.OP_BN_BWD_SUMS: "cg3d_bn_bwd_sums",
    engine.OP_BN_BWD_APPLY_SUMS: "cg3d_bn_bwd_apply_sums", engine.OP_BN_BWD_APPLY: "cg3d_bn_bwd_apply",
    engine.OP_INTERP_MAP: "cg3d_interp_map", engine.OP_INTERP_FWD: "cg3d_interp_fwd", engine.OP_INTERP_BWD: "cg3d_interp_bwd",
    engine.OP_GATHER_ROWS: "cg3d_gather_rows", engine.OP_SCATTER_ADD_ROWS: "cg3d_scatter_add_rows",
    engine.OP_SCATTER_MEAN_FWD: "cg3d_scatter_mean_fwd", engine.OP_SCATTER_MEAN_BWD: "cg3d_scatter_mean_bwd",
    engine.OP_TO_BF16_SPLIT: "cg3d_to_bf16_split", engine.OP_FROM_BF16: "cg3d_from_bf16",
}


def test_role_table_follows_the_const_qualifiers_of_the_header():
    src = open(os.path.join(ROOT, "include", "cagroup3d_hip.h")).read()
    for op, name in _ENTRY.items():
        m = re.search(r"\bint\s+" + name + r"\s*\(([^;]*?)\)\s*;", src, re.S)
        assert m, name
        args = [a.strip() for a in re.sub(r"\s+", " ", m.group(1)).split(",")]
        assert args[-1].startswith("cg3d_stream_t")
        rd = tuple(i + 1 for i, a in enumerate(args[:-1]) if "*" in a and a.startswith("const "))
        wr = tuple(i + 1 for i, a in enumerate(args[:-1]) if "*" in a and not a.startswith("const "))
        assert engine.ROLES[op] == (rd, wr), (name, engine.ROLES[op], rd, wr)
    assert engine.ROLES[engine.OP_MEMSET] == ((), (1,)) and engine.ROLES[engine.OP_COPY2D] == ((3,), (1,))
    assert set(engine.ROLES) == set(range(engine.OP_EVENT_WAIT + 1))
    prog_h = open(os.path.join(ROOT, "include", "cagroup3d_program.h")).read()
    assert "CG3D_OP_EVENT_WAIT = %d" % engine.OP_EVENT_WAIT in prog_h
    assert "#define CG3D_PROG_LANE_SHIFT %d" % engine.LANE_SHIFT in prog_h and "#define CG3D_PROG_MAX_LANES %d" % engine.MAX_LANES in prog_h


def test_timing_events_keep_every_row_once_and_in_order(oracle):
    """engine._with_events (bench.py's live timing of the conv launches): the profile entries of a table with lanes are recorded
    in emission order but refer to rows of the ISSUE order -- wrapped out of order, rows ran twice."""
    comp, _ = _compiled_backbone(oracle, 2)
    for tab, prof in ((comp.fwd, comp.fprof), (comp.bwd, comp.bprof)):
        assert len(prof) > 20
        rows = [p[0] for p in prof]
        if engine.LANES:
            assert rows != sorted(rows), "the round-robin issue order should have moved profiled rows past each other"
        with _lib.use_library(oracle):
            P, recs = engine._with_events(tab, prof, oracle)
        assert len(recs) == len(prof) and P.shape[0] == tab.shape[0] + 2 * len(prof)
        timing = ((P[:, 0] & engine.OPCODE_MASK) == engine.OP_EVENT_RECORD) & (P[:, 2] == 0)
        assert int(timing.sum()) == 2 * len(prof)
        assert np.array_equal(P[~timing], tab)                           # every row once, in the table's order
        at = np.nonzero(timing)[0]
        for a, b in zip(at[0::2], at[1::2]):                              # an event pair around ONE row, on that row's lane
            assert b == a + 2 and (P[a, 0] >> engine.LANE_SHIFT) == (P[a + 1, 0] >> engine.LANE_SHIFT) == (P[b, 0] >> engine.LANE_SHIFT)


# ------------------------------------------------------------------------------------------------ happens-before walk
def _blocks_of(comp_starts):
    def block(addr):
        tag = addr >> engine.TAG
        if not tag:
            return addr
        lst = comp_starts.get(tag << engine.TAG)
        if not lst:
            return (tag, 0)
        return (tag, lst[bisect.bisect_right(lst, addr - (tag << engine.TAG)) - 1])
    return block


def _check_happens_before(table, starts, cuts=()):
    """Vector clocks over the table as it will be issued.  clock[L][M] = number of rows of lane M known to be complete before
    the next row of lane L starts.  Returns the number of cross-lane dependences found (all of them ordered, else asserts)."""
    block = _blocks_of(starts)
    op = table[:, 0] & engine.OPCODE_MASK
    lane = (table[:, 0] >> engine.LANE_SHIFT).astype(int)
    NL = int(lane.max()) + 1
    bounds = [0] + sorted(set(cuts)) + [table.shape[0]]
    found = 0
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        clock = [[0] * NL for _ in range(NL)]
        done = [0] * NL
        events = {}
        lastw, lastr = {}, {}
        for i in range(lo, hi):
            L, o, row = lane[i], int(op[i]), table[i]
            if o == engine.OP_EVENT_RECORD and row[2] == 1:
                c = list(clock[L])
                c[L] = done[L]
                events[int(row[1])] = c
                continue
            if o == engine.OP_EVENT_WAIT:
                assert int(row[1]) in events, "row %d waits for an event no earlier row records" % i
                clock[L] = [max(a, b) for a, b in zip(clock[L], events[int(row[1])])]
                continue
            if L != 0 and done[L] == 0:
                # a side lane's first row: behind a wait for lane 0's position at the start of the part (the zero-fill precedes it)
                assert clock[L][0] >= 0 and any(int(op[j]) == engine.OP_EVENT_WAIT and lane[j] == L for j in range(lo, i)), \
                    "lane %d starts at row %d without a fork" % (L, i)
            rd, wr = engine.ROLES[o]
            br = [block(int(row[c])) for c in rd if row[c]]
            bw = [block(int(row[c])) for c in wr if row[c]]
            for bk in br:
                for M, q in enumerate(lastw.get(bk, ())):
                    if M != L and q:
                        found += 1
                        assert clock[L][M] >= q, "row %d (lane %d, opcode %d) reads a block lane %d wrote without an edge" % (i, L, o, M)
            for bk in bw:
                for tab in (lastw, lastr):
                    for M, q in enumerate(tab.get(bk, ())):
                        if M != L and q:
                            found += 1
                            assert clock[L][M] >= q, "row %d (lane %d, opcode %d) writes a block lane %d still uses" % (i, L, o, M)
            done[L] += 1
            for bk in br:
                lastr.setdefault(bk, [0] * NL)[L] = done[L]
            for bk in bw:
                lastw.setdefault(bk, [0] * NL)[L] = done[L]
                lastr.setdefault(bk, [0] * NL)[L] = done[L]
        for M in range(1, NL):                  # the part ends with lane 0 behind every other lane
            assert clock[0][M] >= done[M], "lane %d is not joined at row %d" % (M, hi)
    return found


def _compiled_backbone(oracle, wgrad_lane=0):
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        os.environ["CG3D_ENGINE_ANY"] = "1"
        wl, engine.WGRAD_LANE = engine.WGRAD_LANE, wgrad_lane
        captured = {}
        Builder = engine.Builder
        init = Builder.__init__

        def spy(self, *a, **k):
            init(self, *a, **k)
            captured["b"] = self
        Builder.__init__ = spy
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            model.train()
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            sp = model.voxelization(batch["points"].clone())
            comp = engine.compile_backbone(model.backbone_3d, sp, mid_mark=True)
        finally:
            Builder.__init__ = init
            engine.WGRAD_LANE = wl
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)
    return comp, captured["b"].starts


@pytest.mark.parametrize("wgrad_lane", [0, 2])
def test_every_cross_lane_dependence_of_the_backbone_tables_has_an_edge(oracle, wgrad_lane):
    if not engine.LANES:
        pytest.skip("CG3D_LANES=0")
    comp, starts = _compiled_backbone(oracle, wgrad_lane)
    assert comp.lanes and comp.nevents > 0
    lanes_f = set((comp.fwd[:, 0] >> engine.LANE_SHIFT).tolist())
    lanes_b = set((comp.bwd[:, 0] >> engine.LANE_SHIFT).tolist())
    side = set(engine.DAPPM_LANES)
    assert lanes_f == {0, 1} | side and lanes_b == {0, 1} | side | ({wgrad_lane} if wgrad_lane else set())
    nf = _check_happens_before(comp.fwd, starts)
    nb = _check_happens_before(comp.bwd, starts, [comp.marks["mid"]])
    # the three joins of the bilateral net, both ways, in both passes
    assert nf >= 6 and nb >= 6, (nf, nb)
    # the cut sits between the join of the first part and the second part's rows
    cut = comp.marks["mid"]
    assert 0 < cut < comp.bwd.shape[0]
    assert (comp.bwd[cut:, 0] >> engine.LANE_SHIFT).max() == wgrad_lane          # layer2 .. conv1: one chain (+ its weight gradients)


@pytest.mark.parametrize("wgrad_lane", [0, 2])
def test_the_library_derives_what_the_python_specification_derives(oracle, wgrad_lane):
    """cg3d_program_schedule (the product's path) against engine._schedule (its specification, checked above): same rows in
    the same order, same index map, same cut, same number of events -- and its role table is the one of engine.ROLES."""
    if not engine.LANES:
        pytest.skip("CG3D_LANES=0")
    import ctypes
    for op, (rd, wr) in engine.ROLES.items():
        a, b = ctypes.c_uint32(0), ctypes.c_uint32(0)
        assert oracle.raw("cg3d_program_roles")(op, ctypes.cast(ctypes.pointer(a), ctypes.c_void_p), ctypes.cast(ctypes.pointer(b), ctypes.c_void_p)) == 0
        assert a.value == sum(1 << c for c in rd) and b.value == sum(1 << c for c in wr), op
    native = engine.SCHED_NATIVE
    try:
        engine.SCHED_NATIVE = True
        c1, _ = _compiled_backbone(oracle, wgrad_lane)
        engine.SCHED_NATIVE = False
        c2, _ = _compiled_backbone(oracle, wgrad_lane)
    finally:
        engine.SCHED_NATIVE = native

    def plain(t):                                               # (absolute addresses differ between two compilations: tables of the
        t = t.copy()                                            #  coordinate manager are rebuilt; compare structure + region offsets)
        t[(t >> engine.TAG) == 0] = 0
        return t
    for other in (c2,):
        assert c1.nevents == other.nevents and c1.marks == other.marks
        for a, b in ((c1.fwd, other.fwd), (c1.bwd, other.bwd)):
            assert a.shape == b.shape
            assert np.array_equal(a[:, 0], b[:, 0])
            ev = np.isin(a[:, 0] & engine.OPCODE_MASK, (engine.OP_EVENT_RECORD, engine.OP_EVENT_WAIT))
            assert np.array_equal(a[ev], b[ev])
            assert np.array_equal(plain(a), plain(b))
        assert [r[0] for r in c1.fprof] == [r[0] for r in other.fprof] and [r[0] for r in c1.bprof] == [r[0] for r in other.bprof]
        assert [(r, c) for r, c, _ in c1.late_b] == [(r, c) for r, c, _ in other.late_b]


def test_a_missing_edge_is_found_by_the_walk(oracle):
    """The checker itself: drop one WAIT row of a scheduled table and it must object."""
    if not engine.LANES:
        pytest.skip("CG3D_LANES=0")
    comp, starts = _compiled_backbone(oracle)
    tab = comp.fwd
    waits = np.nonzero(((tab[:, 0] & engine.OPCODE_MASK) == engine.OP_EVENT_WAIT))[0]
    assert len(waits) >= 4
    broken = 0
    for w in waits[1:]:                              # (the first one is lane 1's fork)
        t = np.delete(tab, w, axis=0)
        try:
            _check_happens_before(t, starts)
        except AssertionError:
            broken += 1
    # (with more than two lanes some edges are implied by others -- a lane that waited for a lane that had waited)
    assert broken >= max(len(waits) // 2, 3), (broken, len(waits))


# ------------------------------------------------------------------------------------------------ other legal orders
def _reorder(P, priority):
    """The rows of `P` in the order a machine would run them that always advances the first lane of `priority` that can
    advance (a WAIT can advance once its RECORD has run).  Lane bits are cleared: the result is a one-queue table."""
    op = P[:, 0] & engine.OPCODE_MASK
    lane = (P[:, 0] >> engine.LANE_SHIFT).astype(int)
    queues = {l: [i for i in range(P.shape[0]) if lane[i] == l] for l in set(lane.tolist())}
    pos = {l: 0 for l in queues}
    priority = list(priority) + sorted(l for l in queues if l not in priority)
    recorded, order = set(), []
    while any(pos[l] < len(queues[l]) for l in queues):
        for l in priority:
            if l not in queues or pos[l] >= len(queues[l]):
                continue
            i = queues[l][pos[l]]
            if op[i] == engine.OP_EVENT_WAIT and int(P[i, 1]) not in recorded:
                continue
            if op[i] == engine.OP_EVENT_RECORD:
                recorded.add(int(P[i, 1]))
            order.append(i)
            pos[l] += 1
            break
        else:
            raise AssertionError("deadlock: every lane waits for an event nobody has recorded")
    Q = P[order].copy()
    Q[:, 0] &= engine.OPCODE_MASK
    return Q, order


@pytest.mark.parametrize("wgrad_lane", [2])
def test_other_orders_the_edges_allow_give_identical_results_on_the_oracle(oracle, wgrad_lane):
    if not engine.LANES:
        pytest.skip("CG3D_LANES=0")
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        wl, engine.WGRAD_LANE = engine.WGRAD_LANE, wgrad_lane
        run = engine._run
        bound, engine.BOUND = engine.BOUND, False              # (the passes go through engine._run, where this test re-orders them)
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            state = {k: v.clone() for k, v in model.state_dict().items()}
            results, moved = [], []
            for priority in (None, (2, 1, 0), (0, 1, 2), (1, 0, 2)):
                model.load_state_dict(state)

                def shuffled(lib, P, nrows=None, comp=None, lanes_run=None, priority=priority):
                    if priority is None or comp is None or not comp.lanes:
                        return run(lib, P, nrows, comp)
                    P = engine._bind_events(np.ascontiguousarray(P).copy(), lib, comp.nevents)
                    # (event rows now carry handles; RECORD / WAIT pair up by handle instead of slot, which _reorder reads from column 1 too)
                    Q, order = _reorder(P, priority)
                    moved.append(sum(1 for a, b in zip(order, range(len(order))) if a != b))
                    return run(lib, Q, nrows, None)
                engine._run = shuffled
                results.append(_backbone_step(model, batch, True, "cpu"))
            assert sum(1 for m in moved if m >= 20) >= 6, moved          # forward + backward table of each order, really reordered
        finally:
            engine._run = run
            engine.BOUND = bound
            engine.WGRAD_LANE = wl
            me.PRECISION = prec
    ref = results[0]
    for got in results[1:]:
        assert torch.equal(ref[1], got[1])
        assert set(ref[2]) == set(got[2])
        for k in ref[2]:
            assert torch.equal(ref[2][k], got[2][k]), k
        for k in ref[3]:
            assert torch.equal(ref[3][k], got[3][k]), k


def test_reorder_helper_detects_a_dropped_edge(oracle):
    """Counter-check of the order test: with ONE wait removed some legal-looking order runs a consumer before its producer and
    the outputs differ from the reference."""
    if not engine.LANES:
        pytest.skip("CG3D_LANES=0")
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        run = engine._run
        bound, engine.BOUND = engine.BOUND, False
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            state = {k: v.clone() for k, v in model.state_dict().items()}
            ref = _backbone_step(model, batch, True, "cpu")
            model.load_state_dict(state)
            calls = []

            def broken(lib, P, nrows=None, comp=None, lanes_run=None):
                if comp is None or not comp.lanes:
                    return run(lib, P, nrows, comp)
                P = engine._bind_events(np.ascontiguousarray(P).copy(), lib, comp.nevents)
                if not calls:                                        # the forward table: drop the waits of lane 0 on lane 1
                    op, lane = P[:, 0] & engine.OPCODE_MASK, P[:, 0] >> engine.LANE_SHIFT
                    drop = np.nonzero((op == engine.OP_EVENT_WAIT) & (lane == 0))[0]
                    P[drop, 0] = engine.OP_NOP
                calls.append(1)
                Q, _ = _reorder(P, (0, 1, 2))
                return run(lib, Q, nrows, None)
            engine._run = broken
            got = _backbone_step(model, batch, True, "cpu")
        finally:
            engine._run = run
            engine.BOUND = bound
            me.PRECISION = prec
    assert not torch.equal(ref[1], got[1])


def test_a_pass_issued_from_its_compiled_table_equals_the_resolved_copy(oracle):
    """cg3d_run_program_bound (region bases and event handles applied by the library, row by row) against the numpy path
    (engine._resolve + _bind_events + cg3d_run_program_lanes): bit-identical outputs, gradients and running statistics."""
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 0
        bound = engine.BOUND
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            batch = build_model.synthetic_batch("S5k", 1, device="cpu")
            state = {k: v.clone() for k, v in model.state_dict().items()}
            out = []
            for b in (True, False):
                engine.BOUND = b
                model.load_state_dict(state)
                out.append(_backbone_step(model, batch, True, "cpu"))
        finally:
            engine.BOUND = bound
            me.PRECISION = prec
    a, b = out
    assert torch.equal(a[1], b[1]) and set(a[2]) == set(b[2])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


def test_bound_run_applies_bases_and_events_and_refuses_bad_slots(oracle):
    """cg3d_run_program_bound on a hand-written table: the zero-fill first, region-relative words resolved through `bases`, event
    slots through `events` (an unknown slot is an argument error with its row in *fail_at), a lane beyond nstreams on stream 0."""
    import ctypes
    f = oracle.raw("cg3d_run_program_bound")
    buf = np.full(64, 7, dtype=np.uint8)
    src = np.arange(16, dtype=np.float32)
    dst = np.zeros(16, dtype=np.uint16)
    R = 3                                                           # a region tag
    tab = np.zeros((4, engine.STRIDE), dtype=np.int64)
    tab[0, :3] = (engine.OP_EVENT_RECORD, 0, 1)
    tab[1, :4] = (engine.OP_TO_BF16 | (1 << engine.LANE_SHIFT), (R << engine.TAG) + 0, (R << engine.TAG) + 64, 16)      # lane 1 of 1 stream
    tab[2, :3] = (engine.OP_EVENT_WAIT | (1 << engine.LANE_SHIFT), 0, 1)
    tab[3, :4] = (engine.OP_MEMSET, (2 << engine.TAG) + 8, 1, 4)
    arena = np.zeros(64 + 32, dtype=np.uint8)
    arena[:64] = src.view(np.uint8)
    bases = np.zeros(16, dtype=np.int64)
    bases[R], bases[2] = arena.ctypes.data, buf.ctypes.data
    events = np.array([1234], dtype=np.int64)
    streams = (ctypes.c_void_p * 8)(*([None] * 8))
    fail = ctypes.c_int64(-5)
    rc = f(tab.ctypes.data, 4, bases.ctypes.data, events.ctypes.data, 1, buf.ctypes.data, 32, ctypes.cast(streams, ctypes.c_void_p), 1,
           ctypes.cast(ctypes.pointer(fail), ctypes.c_void_p))
    assert rc == 0
    assert (buf[:8] == 0).all() and (buf[8:12] == 1).all() and (buf[12:32] == 0).all() and (buf[32:] == 7).all()
    got = arena[64:].view(np.uint16)
    assert np.array_equal(got, (src.view(np.uint32) >> 16).astype(np.uint16))        # exact for small integers
    assert tab[0, 1] == 0 and tab[0, 2] == 1, "the caller's table is not written to"
    rc = f(tab.ctypes.data, 4, bases.ctypes.data, events.ctypes.data, 0, None, 0, ctypes.cast(streams, ctypes.c_void_p), 1,
           ctypes.cast(ctypes.pointer(fail), ctypes.c_void_p))
    assert rc != 0 and fail.value == 0                              # slot 0 of zero events
    rc = f(tab.ctypes.data, 4, bases.ctypes.data, events.ctypes.data, 1, None, 0, ctypes.cast(streams, ctypes.c_void_p), 0,
           ctypes.cast(ctypes.pointer(fail), ctypes.c_void_p))
    assert rc != 0                                                  # no stream


# ------------------------------------------------------------------------------------------------ the tuner
class _FakeEvent:
    clock = [0.0]

    def __init__(self, enable_timing=False):
        self.t = None

    def record(self):
        self.t = _FakeEvent.clock[0]

    def elapsed_time(self, other):
        assert other.synced, "elapsed_time on an end event nobody synchronised"
        return other.t - self.t

    synced = False

    def synchronize(self):
        self.synced = True


def _tune(monkeypatch, ms_of):
    """Drive engine._LaneTuner with scripted pass times: ms_of(lanes_run, n_in, k) for the k-th measured pass."""
    class Lib:
        is_device = True

    class Comp:
        lanes = True

        def __init__(self, n):
            self.n_in = n
    monkeypatch.setattr(torch.cuda, "Event", _FakeEvent)
    monkeypatch.setattr(engine, "AUTOTUNE", True)
    monkeypatch.setattr(engine, "LANES_RUN", True)
    T = engine._LaneTuner
    monkeypatch.setattr(T, "skip", 1)
    monkeypatch.setattr(T, "samples", [])
    monkeypatch.setattr(T, "decided", False)
    monkeypatch.setattr(T, "verdict", None)
    modes, k = [], 0
    for i in range(12):
        comp = Comp(100000 + 9000 * (i % 3))                 # batches of different sizes, as in training
        tok = T.begin(comp, Lib())
        if tok is not None:
            modes.append(tok[0])
            _FakeEvent.clock[0] += ms_of(tok[0], comp.n_in, k)
            k += 1
        T.end(tok)
    return T, modes


def test_the_tuner_keeps_lanes_that_win_and_drops_lanes_that_lose(monkeypatch, capsys):
    T, modes = _tune(monkeypatch, lambda on, n, k: (4.0 if on else 4.7) * n / 100000.0)
    assert modes == [True, False, True, False] and T.decided and engine.LANES_RUN is True
    assert 0.80 < T.verdict[0] / T.verdict[1] < 0.90
    # two lanes on one hardware queue: slower than one stream -> off, and said so
    T, _ = _tune(monkeypatch, lambda on, n, k: (6.1 if on else 4.5) * n / 100000.0)
    assert T.decided and engine.LANES_RUN is False
    assert "lanes off" in capsys.readouterr().err
    # equal within 5 %: the lanes stay (they cost nothing when they do not win)
    T, _ = _tune(monkeypatch, lambda on, n, k: (4.6 if on else 4.5) * n / 100000.0)
    assert T.decided and engine.LANES_RUN is True
    # a profiled step is not sampled, a pass without lanes neither
    monkeypatch.setattr(T, "decided", False)
    monkeypatch.setattr(T, "samples", [])
    monkeypatch.setattr(T, "skip", 0)
    monkeypatch.setattr(me.KernelProfile, "enabled", True)

    class Lib:
        is_device = True

    class Comp:
        lanes, n_in = True, 1000
    assert T.begin(Comp(), Lib()) is None
    monkeypatch.setattr(me.KernelProfile, "enabled", False)
    Comp.lanes = False
    assert T.begin(Comp(), Lib()) is None


# ------------------------------------------------------------------------------------------------ the class branches
def test_class_branch_tables_have_their_edges(oracle):
    if not engine.LANES:
        pytest.skip("CG3D_LANES=0")
    from test_engine import _class_branch_inputs, _class_branch_step
    with _lib.use_library(oracle):
        prec, me.PRECISION = me.PRECISION, 1
        os.environ["CG3D_ENGINE_ANY"] = "1"
        captured, comps = {}, []
        init, compile_cb = engine.Builder.__init__, engine.compile_class_branches

        def spy(self, *a, **k):
            init(self, *a, **k)
            captured["b"] = self

        def grab(*a, **k):
            c = compile_cb(*a, **k)
            comps.append(c)
            return c
        engine.Builder.__init__, engine.compile_class_branches = spy, grab
        try:
            model, _ = build_model.build_cagroup3d("scannet", seed=0)
            head = model.dense_head.train()
            fine, coarse, feat, up = _class_branch_inputs(head, "cpu", base=40)
            _class_branch_step(head, fine, coarse, feat, up, True, 2)
        finally:
            engine.Builder.__init__, engine.compile_class_branches = init, compile_cb
            me.PRECISION = prec
            os.environ.pop("CG3D_ENGINE_ANY", None)
            me._WeightPlan.reset()
    assert comps, "the class-branch program did not compile"
    comp, starts = comps[-1], captured["b"].starts
    assert comp.lanes
    assert _check_happens_before(comp.fwd, starts) >= 1          # the concatenation reads both branches
    assert _check_happens_before(comp.bwd, starts) >= 1


# ------------------------------------------------------------------------------------------------ device
@pytest.mark.gpu
@pytest.mark.parametrize("wgrad_lane", [0, 2])
def test_lanes_on_their_queues_equal_the_one_queue_run_on_the_device(hip, wgrad_lane):
    """Same tables (event edges included) issued on one stream and on one stream per lane: the results agree to the run-to-run
    noise of the fp32 atomics (statistics, scatter, weight-gradient segments), measured by a second one-queue run."""
    if not engine.LANES:
        pytest.skip("CG3D_LANES=0")
    dev = "cuda"
    model, _ = build_model.build_cagroup3d("scannet", seed=0)
    model = model.to(dev)
    batch = build_model.synthetic_batch("S50k", 1, device=dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    prec, me.PRECISION = me.PRECISION, 1
    rows16, me.BF16_ROWS = me.BF16_ROWS, True
    wl, engine.WGRAD_LANE = engine.WGRAD_LANE, wgrad_lane
    lr, at, engine.AUTOTUNE = engine.LANES_RUN, engine.AUTOTUNE, False       # (the test chooses the queues itself)
    try:
        _backbone_step(model, batch, True, dev)                  # (first step: weights enter the arena)
        out = []
        for lanes_run in (False, False, True):
            model.load_state_dict(state)
            engine.LANES_RUN = lanes_run
            passes = engine.STATS["program_passes"]
            out.append(_backbone_step(model, batch, True, dev))
            assert engine.STATS["program_passes"] == passes + 1, "the program path did not run"
            torch.cuda.synchronize()
    finally:
        engine.LANES_RUN, engine.AUTOTUNE = lr, at
        engine.WGRAD_LANE = wl
        me.PRECISION, me.BF16_ROWS = prec, rows16
    a, b, c = out

    def l2(x, y):
        return float((x.double() - y.double()).norm() / (x.double().norm() + 1e-30))
    noise = max(l2(a[1], b[1]), 1e-6)
    assert l2(a[1], c[1]) <= 3 * noise + 1e-5, (l2(a[1], c[1]), noise)
    for k in a[2]:
        if float(a[2][k].norm()) > 1e-3:
            n = max(l2(a[2][k], b[2][k]), 1e-6)
            e = l2(a[2][k], c[2][k])
            assert e <= 3 * n + 4e-3, (k, e, n)
