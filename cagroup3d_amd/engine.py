"""Launch programs for the BiResNet backbone: one foreign call per pass (include/cagroup3d_program.h).

The per-layer host path of `me.py` -- an `autograd.Function`, three or four `torch.empty`, a ctypes call with twenty marshalled
arguments and a `SparseTensor` per op -- cost the launching thread ~110 us per layer: 15 ms of the 30 ms a training step took to
ISSUE, with the GPU waiting (profiles/r04_host_sections.txt).  Here the same sequence of C-ABI calls is written down once per
step as a table of rows (opcode + arguments) by plain integer arithmetic: every activation, gradient, statistics table and
scratch buffer of the pass is an offset into ONE arena, the coordinate structures (kernel maps, tile plans, pair lists, segment
tables) are the ones `me.CoordinateManager` builds anyway, the weights' bf16 copies are slices of the step's weight arena.
`cg3d_run_program` then issues the whole forward pass -- and, from the one autograd node that stands for the backbone, the whole
backward pass -- from C.

What a program contains is exactly what the per-layer path launches (same kernels, same arguments, same decisions: the
functions `me._use_tile`, `SparseConvFunction._implicit`, `_wgrad_seg_len`... are called from here), so the per-layer path stays
the specification: `tests/test_engine.py` runs both and compares every output and gradient.  Differences, all on purpose:
  * zero-initialised buffers (statistics tables, atomic-scatter outputs) live in one region per pass, cleared by ONE memset;
  * all parameter gradients of the backbone are slices of one zero-filled buffer per backward pass, accumulated into with
    CG3D_WGRAD_ACCUMULATE, and handed to the parameters directly (no AccumulateGrad node per parameter);
  * sums of gradient contributions are formed when the gradient is first read, one launch per extra contribution.

Reference: pcdet/models/backbones_3d/biresnet.py:8-406 (the module tree `emit` walks is cagroup3d_amd's mirror of it).
"""
import collections
import os
import struct

import numpy as np
import torch

from . import _lib
from . import me as ME

ENABLED = os.environ.get("CG3D_ENGINE", "1") != "0"
# how the backbone passes of this process were issued (bench.py prints it: a run that silently fell back to the per-layer
# path must not be read as an engine measurement)
STATS = {"program_passes": 0, "compiled_ahead": 0, "compiled_inline": 0, "not_ready": 0}

# opcodes of include/cagroup3d_program.h
(OP_NOP, OP_MEMSET, OP_COPY2D, OP_TO_BF16, OP_TILE_FWD, OP_SPCONV_FWD, OP_SPCONV_FWD_TILED, OP_PAIRS_FWD, OP_PAIRS_WGRAD,
 OP_LINEAR_FWD, OP_BN_SUMS, OP_BN_APPLY_SUMS, OP_BN_APPLY, OP_BN_BWD_SUMS, OP_BN_BWD_APPLY_SUMS, OP_BN_BWD_APPLY, OP_INTERP_MAP,
 OP_INTERP_FWD, OP_INTERP_BWD, OP_GATHER_ROWS, OP_SCATTER_ADD_ROWS, OP_SCATTER_MEAN_FWD, OP_SCATTER_MEAN_BWD,
 OP_EVENT_RECORD, OP_TO_BF16_SPLIT, OP_FROM_BF16, OP_EVENT_WAIT) = range(27)
# Lanes (include/cagroup3d_program.h): the two chains of the bilateral backbone -- and the two independent branches of the class
# program -- are issued on two queues, ordered by event edges that `_schedule` derives from what every row reads and writes.
# CG3D_LANES=0: everything on the one stream, as until round 5.
LANES = os.environ.get("CG3D_LANES", "1") != "0"
LANES_RUN = os.environ.get("CG3D_LANES_RUN", "1") != "0"     # 0: tables WITH their event edges, every lane on the one stream (A/B runs)
LANE_SHIFT = 32
MAX_LANES = 8               # CG3D_PROG_MAX_LANES
# The weight gradient of a layer feeds nothing inside the pass: on lane 2 it runs beside the data-gradient chain instead of in
# it (backbone backward 7.4 -> 7.0 ms for 4 x S50k, tools/backbone_lanes.py).  0: on the layer's lane.
WGRAD_LANE = int(os.environ.get("CG3D_WGRAD_LANE", "2"))
# lanes of DAPPM's pooled branches and shortcut (empty: with the rest of the coarse chain).  Two lanes: forward 4.35 -> 4.05 ms,
# backward 7.05 -> 6.6 ms; four lanes, or eight hardware queues (GPU_MAX_HW_QUEUES, default 4), were slower.
DAPPM_LANES = [int(x) for x in os.environ.get("CG3D_DAPPM_LANES", "3,2").split(",") if x]
OPCODE_MASK = (1 << LANE_SHIFT) - 1
# Pointer arguments of every opcode as (columns read, columns written) of a row (column 0 is the opcode): the `const T *` and the
# `T *` parameters of the entry point the opcode names (tests/test_engine.py checks this table against include/cagroup3d_hip.h).
# A written column counts as read too (accumulating kernels).
ROLES = {
    OP_NOP: ((), ()), OP_MEMSET: ((), (1,)), OP_COPY2D: ((3,), (1,)), OP_TO_BF16: ((1,), (2,)),
    OP_TILE_FWD: ((1, 2, 3, 4, 5, 6, 7, 10, 12, 13), (14, 22)), OP_SPCONV_FWD: ((1, 2, 3, 4), (5,)),
    OP_SPCONV_FWD_TILED: ((1, 2, 3, 4, 6), (7,)), OP_PAIRS_FWD: ((1, 2, 3, 4, 5, 7), (8,)),
    OP_PAIRS_WGRAD: ((1, 2, 3, 4, 5), (7,)), OP_LINEAR_FWD: ((1, 2, 3), (4, 9, 10)), OP_BN_SUMS: ((1, 2), (6,)),
    OP_BN_APPLY_SUMS: ((1, 2, 3, 7, 8, 10, 11), (13, 14, 15, 16, 17, 18, 19)), OP_BN_APPLY: ((1, 2, 3, 6, 7, 9, 10), (12, 13)),
    OP_BN_BWD_SUMS: ((1, 2, 3, 4, 8, 9), (12,)), OP_BN_BWD_APPLY_SUMS: ((1, 2, 3, 4, 8, 9, 11, 12, 13), (16, 17, 18, 19, 20)),
    OP_BN_BWD_APPLY: ((1, 2, 3, 4, 7, 8, 10, 11, 12, 13), (16, 17, 18)), OP_INTERP_MAP: ((1, 4, 5), (7, 8)),
    OP_INTERP_FWD: ((1, 2, 3), (4,)), OP_INTERP_BWD: ((1, 2, 3), (4,)), OP_GATHER_ROWS: ((1, 2), (3,)),
    OP_SCATTER_ADD_ROWS: ((1, 2), (3,)), OP_SCATTER_MEAN_FWD: ((1, 2), (4, 5)), OP_SCATTER_MEAN_BWD: ((1, 2, 3), (5,)),
    OP_EVENT_RECORD: ((), ()), OP_TO_BF16_SPLIT: ((1,), (2,)), OP_FROM_BF16: ((1,), (2,)), OP_EVENT_WAIT: ((), ()),
}
S16 = 0x100                 # CG3D_BN_STORE_BF16: the row matrices of a BatchNorm call are bf16 rows
TILE_OUT16 = 2              # CG3D_TILE_OUT_BF16 (bit of cg3d_spconv_tile_fwd's wrev argument)
LIN_OUT16 = 0x10000         # CG3D_LINEAR_OUT_BF16 (bit of cg3d_linear_fwd's ksplit argument)
# Activations and activation gradients of the BACKBONE program stored as bf16 only (BASELINE.json configs[1], "bf16 backbone"):
# the tile / linear convolutions write bf16 rows (+ fp32 statistics), the BatchNorm kernels read and write bf16 and compute in
# fp32, gradient contributions are summed from bf16 rows in fp32 registers.  A layer moves 6-8 bytes per element in the forward
# instead of 14-18 and 12-16 in the backward instead of 30-34 (include/cagroup3d_hip.h, CG3D_BN_STORE_BF16).  CG3D_ACT_BF16=0:
# fp32 storage with bf16 operand copies, the arithmetic of rounds 1-4 (and what the per-layer path of me.py computes).
ACT_BF16 = os.environ.get("CG3D_ACT_BF16", "1") != "0"
ACT16_MIN_ROWS = int(os.environ.get("CG3D_ACT16_MIN_ROWS", "4096"))
STRIDE = 24
WGRAD_ACC = 0x100

# An address inside a program row is either absolute (< 2^56: weights, coordinate structures, module buffers) or an offset into
# one of the regions below, tagged in the top bits and resolved when the pass is run.
TAG = 56
R_ACT, R_ZF, R_ZB, R_PG, R_DOUT, R_IN, R_IN2, R_DOUT2 = (r << TAG for r in range(1, 9))
#   R_ACT  activations, gradients, scratch          R_ZF  zero-filled at the start of the forward pass
#   R_ZB   zero-filled at the start of the backward pass
#   R_PG   the parameter-gradient buffer of a backward pass (fresh zeros per pass)
#   R_DOUT the gradient of the backbone output (known when backward runs)       R_IN the input features
#   R_IN2  the second input of a two-input program (the class branches: features on the coarse map; the head's first layers:
#          the bf16 copy of the input rows)                R_DOUT2 the gradient of a second output
ALIGN = 256


def _fbits(x):
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


class NotReady(Exception):
    """Something the program needs does not exist yet (a weight not recorded in the step's weight arena): the caller runs the
    per-layer path this step."""


class T:
    """A feature matrix inside a program: fp32 rows [n, c] at `p` (+ optional bf16 copy, + the BatchNorm statistics table its
    producer filled) and, during backward emission, the list of its gradient contributions."""
    __slots__ = ("p", "n", "c", "p16", "stats", "need", "gc", "gsum", "gsum16", "gdone")

    def __init__(self, p, n, c, need=True):
        self.p, self.n, self.c, self.p16, self.stats, self.need = p, n, c, 0, 0, need
        # (bf16 storage, Builder.act16: p may be 0 -- the rows exist as bf16 only, at p16)
        self.gc = None          # [(address or 0, bf16 address or 0)] gradient contributions (backward emission)
        self.gsum = None        # address of their sum once it has been formed (0: the sum exists as bf16 rows only)
        self.gsum16 = 0
        self.gdone = False


class _Events:
    """A pair of C-ABI timing events with the `elapsed_time` face of torch.cuda.Event (me.KernelProfile.records)."""

    def __init__(self, lib):
        self.lib = lib
        h = (ctypes_i64(), ctypes_i64())
        lib.call("cg3d_event_create", ctypes_ref(h[0]))
        lib.call("cg3d_event_create", ctypes_ref(h[1]))
        self.h = (h[0].value, h[1].value)

    def elapsed_time(self, _other=None):
        ms = ctypes_f32()
        self.lib.call("cg3d_event_elapsed_ms", self.h[0], self.h[1], ctypes_ref(ms))
        return float(ms.value)

    def __del__(self):
        try:
            self.lib.call("cg3d_event_destroy", self.h[0])
            self.lib.call("cg3d_event_destroy", self.h[1])
        except Exception:
            pass


def ctypes_i64():
    import ctypes
    return ctypes.c_int64(0)


def ctypes_f32():
    import ctypes
    return ctypes.c_float(0.0)


def ctypes_ref(x):
    import ctypes
    return ctypes.cast(ctypes.pointer(x), ctypes.c_void_p)


class _EvStart:          # (ev0 of a KernelProfile record: elapsed_time(ev1) is answered by the pair)
    def __init__(self, pair):
        self.pair = pair

    def elapsed_time(self, _ev1):
        return self.pair.elapsed_time()


class Program:
    """The rows of one pass, still with region-relative addresses."""

    def __init__(self):
        self.rows = []
        self.lanes = []         # lane of every row
        self.lane = 0           # lane of the rows being added (Builder.set_lane)
        self.prof = []          # (row index, flops, bytes, meta, per-pair bytes): conv launches (KernelProfile)
        self.nevents = 0        # ordering events the scheduled table refers to by slot number (_schedule)
        self.scheduled = None   # the table with its event edges, when the library derived them (_schedule_native)

    def add(self, *row):
        self.rows.append(row)
        self.lanes.append(self.lane)

    def table(self):
        if self.scheduled is not None:
            return self.scheduled
        P = np.zeros((len(self.rows), STRIDE), dtype=np.int64)
        for i, r in enumerate(self.rows):
            P[i, :len(r)] = r
        if any(self.lanes):
            P[:, 0] |= np.asarray(self.lanes, dtype=np.int64) << LANE_SHIFT
        return P


class _Tape(list):
    """The backward closures of a pass, each with the lane its forward rows were put on."""

    def __init__(self, builder):
        list.__init__(self)
        self.builder = builder

    def append(self, fn):
        list.append(self, (self.builder.lane, fn))


class Builder:
    def __init__(self, lib, device, gen):
        self.lib, self.dev, self.gen = lib, device, gen
        self.f, self.b = Program(), Program()
        self.lane = 0
        self.tape = _Tape(self)
        self.size = {R_ACT: 0, R_ZF: 0, R_ZB: 0, R_PG: 0}
        self.starts = {R_ACT: [], R_ZF: [], R_ZB: [], R_PG: []}     # block starts per region, ascending (_schedule)
        self.keep = []                  # tensors the rows point into (tables built at emission time)
        # tables out of me.py's host caches: held for their lifetime only (the caches publish an entry after its upload has
        # completed, and what frees one is this reference going away -- no stream hand-over needed: `keep` gets one)
        self.keep_cached, self._held = [], set()
        self.params = []                # (parameter, offset in R_PG)
        self.uses_arena = False         # some row addresses a bf16 weight copy of the step's arena (me._WeightPlan)
        self._pidx = {}
        self.late = []                  # (program, row, column, fn() -> tensor): operands that exist only at run time
        self.marks = {}                 # name -> backward row index (callbacks between two parts of the backward pass)
        self.mark_at = {}               # tape position -> name
        self.bf16 = ME._prec() in (1, 3) and ME.BF16_ROWS
        # split precision (ME.PREC_SPLIT, the two heads): operand rows are [hi | lo | hi] of 3 c channels written by their own
        # pass (the BatchNorm kernels' bf16 copies are not used), weights are three-part, contractions three times as long
        self.kx = ME._kx()
        self.prec = ME._prec()
        self.act16 = False              # set by compile_backbone: activations / gradients stored as bf16 only

    # ---------------------------------------------------------------- memory
    def alloc(self, nbytes, region=R_ACT):
        off = self.size[region]
        self.size[region] = off + ((int(nbytes) + ALIGN - 1) & ~(ALIGN - 1))
        self.starts[region].append(off)
        return region + off

    def wgrad_row(self, *row):
        """A weight-gradient row of the backward table (see WGRAD_LANE)."""
        prev = self.lane
        if WGRAD_LANE:
            self.set_lane(WGRAD_LANE)
        self.b.add(*row)
        self.set_lane(prev)

    def set_lane(self, lane):
        """Rows added from here on (to either table) go to `lane`; returns the lane that was current."""
        prev = self.lane
        self.lane = self.f.lane = self.b.lane = lane if LANES else 0
        return prev

    def new(self, n, c, need=True):
        return T(self.alloc(max(n, 1) * c * 4), n, c, need)

    def new16(self, n, c, need=True):
        """A feature matrix stored as bf16 rows only (act16)."""
        t = T(0, n, c, need)
        t.p16 = self.alloc(max(n, 1) * c * 2)
        return t

    def s16ok(self, c, n=1 << 30):
        """bf16 row storage of an n x c matrix in the BatchNorm kernels (CG3D_BN_STORE_BF16: power-of-two channel counts
        64 .. 1024 -- the rest of the backbone, DAPPM's 640-channel join, stays on fp32 rows).  Matrices of fewer than ACT16_MIN_ROWS
        rows stay fp32 as well: the stride-32 end of the net (layer5, DAPPM: 150-1 300 rows) is made of pooling, interpolation
        and pair-kernel layers that compute on fp32 rows -- bf16 storage there saves no time (every launch sits on its ~7 us
        floor) and cost ~40 conversion launches per step at the seams."""
        return self.act16 and n >= ACT16_MIN_ROWS and 64 <= c <= 1024 and (c & (c - 1)) == 0

    def out(self, n, c, need=True):
        """Where a kernel with both output forms puts a layer's rows: bf16 rows under act16, fp32 rows otherwise."""
        return self.new16(n, c, need) if self.act16 else self.new(n, c, need)

    def f32(self, t):
        """fp32 rows of `t` (forward pass): its own, or one widening pass over the bf16 rows (cg3d_from_bf16)."""
        if not t.p:
            t.p = self.alloc(max(t.n, 1) * t.c * 4)
            self.f.add(OP_FROM_BF16, t.p16, t.p, t.n * t.c)
        return t.p

    def pgrad(self, param):
        """Address (in R_PG) of the gradient of `param`."""
        k = id(param)
        off = self._pidx.get(k)
        if off is None:
            off = self._pidx[k] = self.alloc(param.numel() * 4, R_PG)
            self.params.append((param, off - R_PG))
        return off

    def rows16(self, t):
        """bf16 copy of the rows of `t` (written by its producer, or by one conversion pass)."""
        if not t.p16:
            t.p16 = self.alloc(max(t.n, 1) * t.c * 2 * self.kx)
            self._to16(self.f, t.p, t.p16, t.n, t.c)
        return t.p16

    def _to16(self, prog, p, p16, n, c):
        if self.kx == 3:
            prog.add(OP_TO_BF16_SPLIT, p, p16, n, c)
        else:
            prog.add(OP_TO_BF16, p, p16, n * c)

    def want16(self, c):
        """Whether an apply kernel should leave the plain bf16 copy of its output (never in the split precision)."""
        return self.bf16 and self.kx == 1 and ME._use_bf16(c)

    def host_table(self, data, dtype):
        t = ME.h2d(data, dtype, self.dev)
        self.keep.append(t)
        return t.data_ptr()

    # ---------------------------------------------------------------- gradients (backward emission)
    def gadd(self, t, p, p16=0):
        if t.need:
            if t.gc is None:
                t.gc = []
            t.gc.append((p, p16))
            t.gsum, t.gdone = None, False

    def grad(self, t, want16=False, want32=False):
        """(address or 0, bf16 address or 0) of the complete gradient of `t`, or None when no contribution arrived.
        want16 / want32: make sure that form exists (act16: a gradient may exist as bf16 rows only)."""
        if not t.gc:
            return None
        if not t.gdone:
            if len(t.gc) == 1:
                t.gsum, t.gsum16 = t.gc[0]
            elif self.s16ok(t.c, t.n):
                # every contribution as bf16 rows, added pairwise in fp32 registers (one rounding per sum)
                c16 = []
                for p, p16 in t.gc:
                    if not p16:
                        p16 = self.alloc(max(t.n, 1) * t.c * 2)
                        self.b.add(OP_TO_BF16, p, p16, t.n * t.c)
                    c16.append(p16)
                acc = c16[0]
                for q in c16[1:]:
                    out = self.alloc(max(t.n, 1) * t.c * 2)
                    self._add_rows(self.b, acc, q, out, t.n, t.c, ME.ACT_NONE, 0, s16=True)
                    acc = out
                t.gsum, t.gsum16 = 0, acc
            else:
                c32 = []
                for p, p16 in t.gc:
                    if not p:                                   # (act16, a channel count the bf16 kernels do not take)
                        p = self.alloc(max(t.n, 1) * t.c * 4)
                        self.b.add(OP_FROM_BF16, p16, p, t.n * t.c)
                    c32.append(p)
                acc = c32[0]
                for p in c32[1:]:
                    out = self.alloc(max(t.n, 1) * t.c * 4)
                    self._add_rows(self.b, acc, p, out, t.n, t.c, ME.ACT_NONE, 0)
                    acc = out
                t.gsum, t.gsum16 = acc, 0
            t.gdone = True
        if want16 and not t.gsum16:
            t.gsum16 = self.alloc(max(t.n, 1) * t.c * 2 * self.kx)
            self._to16(self.b, t.gsum, t.gsum16, t.n, t.c)
        if want32 and not t.gsum:
            t.gsum = self.alloc(max(t.n, 1) * t.c * 4)
            self.b.add(OP_FROM_BF16, t.gsum16, t.gsum, t.n * t.c)
        return t.gsum, t.gsum16

    # The chunk / identity-pair / unit tables come out of me.py's host caches, which are CLEARED when they grow past their
    # limit (every new row count is a new key: a training run with varying batches gets there within tens of steps).  A
    # program holds raw addresses, so it must hold the tensors too -- a cleared cache once handed a compiled program's chunk
    # tables to the allocator while the program was still to run (a GPU memory fault three minutes into a training run).
    def _hold(self, t):
        """Keep a cached table (a tuple of tensors and ints) alive with the program -- once: the same table serves many rows."""
        if id(t) not in self._held:
            self._held.add(id(t))
            self.keep_cached.append(t)
        return t

    def _unit(self, c):
        z, o = self._hold(ME._unit_bn(c, self.dev))
        return z.data_ptr(), o.data_ptr()

    def _chunks(self, n, c):
        return self._hold(ME._bn_chunks((0, n), self.dev, c))

    def _ident(self, n, seglen):
        return self._hold(ME._identity_pairs(n, seglen, self.dev))

    def _add_rows(self, prog, a, b, y, n, c, act, y16, s16=False):
        """y = act(a + b) (b may be 0): cg3d_bn_apply with the identity normalisation, as me.AddReluFunction does.
        s16: a, b, y are bf16 rows (act16)."""
        z, o = self._unit(c)
        ch = self._chunks(n, c)
        prog.add(OP_BN_APPLY, a, b, ch[4].data_ptr(), ch[5], c, z, o, _fbits(0.0), o, z, act | (S16 if s16 else 0), y, y16)

    # ---------------------------------------------------------------- weights of the step's arena
    def _planned(self, w3, frag, need_plain):
        P = ME._WeightPlan
        ME._late_needed(w3)
        e = P.singles.get((w3.data_ptr(), ME._wkind(frag)))
        if e is None or e[0].shape != w3.shape or (need_plain and not e[1]) or e[3] is None or P.dirty or P.gen != self.gen:
            ME._planned_single(w3, need_plain, frag)          # records it: converted from the next forward on
            raise NotReady("weight not in the step's arena yet")
        self.uses_arena = True
        return e[3].data_ptr(), (e[4].data_ptr() if e[4] is not None else 0)

    # ---------------------------------------------------------------- convolution (me.SparseConvFunction)
    def conv(self, x, weight, kmap, K, cin, cout):
        lib = self.lib
        w3 = weight.view(K, cin, cout)
        pin, pout, _, P = kmap.pairs(None)
        use16 = self.bf16 and ME._use_bf16(cin)
        tile_f = ME._use_tile(kmap, K, cin, cout, kmap.n_out, None)
        tile_b = x.need and ME._use_tile(kmap, K, cout, cin, kmap.n_in, None)
        implicit_f = ME.SparseConvFunction._implicit(kmap, P, cin, cout, kmap.n_out, None)
        # operands: forward = the transposed copy (fragment order for the tile kernel), data gradient = the plain copy
        wt = wp = 0
        if tile_f:
            wt, wpf = self._planned(w3, True, tile_b)
            if tile_b:
                wp = wpf
        elif ME._use_bf16(cin):
            wt, _ = self._planned(w3, False, False)
        if tile_b and not tile_f:
            _, wp = self._planned(w3, True, True)
        if x.need and not tile_b and ME._use_bf16(cout):
            _, wp = self._planned(w3, False, True)                # (the per-layer path converts this one on the spot when missing)
        xg = self.rows16(x) if use16 else self.f32(x)
        rows16 = bool(use16)
        n_in, n_out = kmap.n_in, kmap.n_out
        prog = self.f
        if tile_f:
            plan = kmap.tile_plan(False)
            y = self.out(n_out, cout)
            if ME.WANT_BN_STATS and ME.FUSED_BN_STATS and cout <= 512 and plan.ntile > 0:
                y.stats = self.alloc(ME.BN_SLOTS * 2 * cout * 4, R_ZF)
            self._tile_row(prog, xg, wt, plan, y.p or y.p16, n_in, cin, cout, False, y.stats, P, out16=not y.p)
        elif implicit_f:
            y = self.new(n_out, cout)
            self._prof(prog, "implicit_bf16" + ME._ksuffix(), K, cin, cout, P, n_in, n_out, 2.0 if rows16 else 4.0, 4.0 * K * n_out)
            prog.add(OP_SPCONV_FWD, xg, wt, kmap.nbr.data_ptr(), 0, y.p, n_in, n_out, K, cin * (self.kx if rows16 else 1), cout, 2 if rows16 else 1)
        else:
            seg, nseg = kmap.segments(ME._seg_len_fwd(), None)
            y = T(self.alloc(max(n_out, 1) * cout * 4, R_ZF), n_out, cout)          # atomic scatter into zeros
            prec = 1 if ME._use_bf16(cin) else 0
            if prec and self.kx == 3 and not rows16:
                raise NotReady("split operands without row copies")
            wptr = wt if prec else w3.data_ptr()
            self._prof(prog, ("pairs_bf16" + ME._ksuffix()) if prec else "pairs", K, cin, cout, P, n_in, n_out, 2.0 if rows16 else 4.0, 8.0 * P,
                       wb=2.0 if prec else 4.0, nseg=nseg)
            prog.add(OP_PAIRS_FWD, xg, wptr, pin.data_ptr(), pout.data_ptr(), seg.data_ptr(), nseg, 0, y.p, n_out,
                     cin * (self.kx if prec else 1), cout, 2 if rows16 else prec, 1)
        self.tape.append(lambda: self._conv_bwd(x, y, weight, w3, kmap, K, cin, cout, P, wp, tile_b))
        return y

    def _prof(self, prog, kind, K, cin, cout, P, n_in, n_out, xb, map_bytes, wb=2.0, nseg=0, groups=1, yb=4.0):
        """SURVEY 8(d) work of a conv launch: 2 P cin cout flops; every tensor once (yb: bytes per stored output element)."""
        kx = 3.0 if kind.endswith("x3") else 1.0          # split operands: rows and weights three times as long
        wbytes = wb * kx * groups * K * cin * cout
        xb = xb * kx
        prog.prof.append((len(prog.rows), 2.0 * P * cin * cout, xb * n_in * cin + yb * n_out * cout + wbytes + map_bytes,
                          (kind, K, cin, cout, P, n_out, nseg), xb * P * cin + yb * n_out * cout + wbytes + map_bytes))

    def _tile_row(self, prog, x16, wf, plan, y, n_in, cin, cout, wrev, stats, P, ksplit=1, groups=1, out16=False):
        def p(t):
            return t.data_ptr() if t is not None else 0
        self._prof(prog, "tile_bf16" + ME._ksuffix(), plan.K, cin, cout, P, n_in, plan.n_out, 2.0, 2.0 * plan.K * plan.n_out, groups=groups,
                   yb=2.0 if out16 else 4.0)
        prog.add(OP_TILE_FWD, x16, wf, p(plan.slots), p(plan.live), p(plan.pass_tab), p(plan.npass), p(plan.ulist), plan.maxpass,
                 plan.ucap, p(plan.tiles), plan.ntile, p(plan.order), 0, y, n_in, plan.n_out, plan.K, cin * self.kx, cout, ksplit,
                 (1 if wrev else 0) | (TILE_OUT16 if out16 else 0), stats)

    def _conv_bwd(self, x, y, weight, w3, kmap, K, cin, cout, P, wp, tile_b):
        use16 = self.bf16 and ME._use_bf16(cout)
        wprec = ME._wgrad_prec(cin, cout, bool(x.p16) and self.bf16)
        # (fp32 rows of dY: the fp32-operand kernels -- the stem's data / weight gradient -- read them)
        g = self.grad(y, want16=use16, want32=(not use16) or wprec < 2)
        if g is None:
            return
        dy, dy16 = g
        prog = self.b
        pin, pout = kmap.pairs(None)[:2]
        dyg = dy16 if use16 else dy
        dx16 = 0
        if x.need:
            if tile_b:
                plan = kmap.tile_plan(True)
                if self.act16:
                    dx, dx16 = 0, self.alloc(max(kmap.n_in, 1) * cin * 2)
                else:
                    dx = self.alloc(max(kmap.n_in, 1) * cin * 4)
                self._tile_row(prog, dyg, wp, plan, dx or dx16, kmap.n_out, cout, cin, kmap.symmetric, 0, P, out16=not dx)
            elif ME.SparseConvFunction._implicit(kmap, P, cout, cin, kmap.n_in, None):
                dx = self.alloc(max(kmap.n_in, 1) * cin * 4)
                if not wp:
                    raise NotReady("plain bf16 copy missing")
                self._prof(prog, "implicit_bf16" + ME._ksuffix(), K, cout, cin, P, kmap.n_out, kmap.n_in, 2.0 if use16 else 4.0, 4.0 * K * kmap.n_in)
                prog.add(OP_SPCONV_FWD, dyg, wp, kmap.nbrT.data_ptr(), 0, dx, kmap.n_out, kmap.n_in, K, cout * (self.kx if use16 else 1), cin, 2 if use16 else 1)
            else:
                seg, nseg = kmap.segments(ME._seg_len_fwd(), None)
                dx = self.alloc(max(kmap.n_in, 1) * cin * 4, R_ZB)
                if ME._use_bf16(cout):
                    if not wp:
                        raise NotReady("plain bf16 copy missing")
                    if self.kx == 3 and not use16:
                        raise NotReady("split operands without row copies")
                    self._prof(prog, "pairs_bf16" + ME._ksuffix(), K, cout, cin, P, kmap.n_out, kmap.n_in, 2.0 if use16 else 4.0, 8.0 * P, nseg=nseg)
                    prog.add(OP_PAIRS_FWD, dyg, wp, pout.data_ptr(), pin.data_ptr(), seg.data_ptr(), nseg, 0, dx, kmap.n_in, cout * self.kx, cin,
                             2 if use16 else 1, 1)
                else:
                    # fp32 operands: W^T as its own tensor, formed when the pass runs (the weights may change until then)
                    self._prof(prog, "pairs", K, cout, cin, P, kmap.n_out, kmap.n_in, 4.0, 8.0 * P, wb=4.0, nseg=nseg)
                    self.late.append((prog, len(prog.rows), 2, lambda w=w3: w.detach().transpose(1, 2).contiguous()))
                    prog.add(OP_PAIRS_FWD, dy, 0, pout.data_ptr(), pin.data_ptr(), seg.data_ptr(), nseg, 0, dx, kmap.n_in, cout, cin, 0, 1)
            self.gadd(x, dx, dx16)
        # weight gradient
        xw, dyw = x.p, dy
        if wprec >= 2:
            xw, dyw = x.p16, (dy16 if dy16 else self.grad(y, want16=True)[1])
        elif not xw:
            raise NotReady("fp32 rows of a bf16-stored input are not kept for the backward pass")
        seg, nseg = (kmap.wgrad_segments if wprec else kmap.segments)(ME._wgrad_seg_len(P, cin, cout, 1 if wprec else 0, K), None)
        eb = 2.0 if wprec >= 2 else 4.0
        if ME.KernelProfile.wgrad:
            prog.prof.append((len(prog.rows), 2.0 * P * cin * cout, eb * (kmap.n_in * cin + kmap.n_out * cout) + 4.0 * K * cin * cout + 8.0 * P,
                              ("wgrad_bf16x3" if wprec == 3 else "wgrad_bf16" if wprec == 2 else ("wgrad_bf16_fp32rows" if wprec else "wgrad"), K, cin, cout, P, kmap.n_out, nseg),
                              eb * P * (cin + cout) + 4.0 * K * cin * cout))
        self.wgrad_row(OP_PAIRS_WGRAD, xw, dyw, pin.data_ptr(), pout.data_ptr(), seg.data_ptr(), nseg, self.pgrad(weight), K, cin, cout,
                       wprec | WGRAD_ACC)

    # ---------------------------------------------------------------- grouped layers (the class branches)
    def hold(self, t):
        self.keep.append(t)
        return t

    def pgrad_group(self, params):
        """Address (in R_PG) of one contiguous block holding the gradients of G same-sized parameters, in order."""
        off = self._pidx.get(id(params[0]))
        if off is None:
            n = params[0].numel()
            off = self.alloc(len(params) * n * 4, R_PG)
            for g, prm in enumerate(params):
                self._pidx[id(prm)] = off + g * n * 4
                self.params.append((prm, off - R_PG + g * n * 4))
        return off

    def gconv(self, x, weights, kmap, row_bounds, closed):
        """me.GroupedConvFunction: G contiguous row groups with their own weights (one parameter per group)."""
        GC = ME.GroupedConvFunction
        w3s = [w.view(-1, w.shape[-2], w.shape[-1]) for w in weights]
        G, (K, cin, cout) = len(w3s), w3s[0].shape
        if not (self.bf16 and ME._use_bf16(cin) and ME._use_bf16(cout)):
            raise NotReady("grouped convolution outside the bf16 mode")
        pin, pout, _, P = kmap.pairs(row_bounds)
        xg = self.rows16(x)
        lds = GC._lds_tile(kmap, K, cin, cout, closed)
        wt = self.hold(ME._prep_bf16_group(w3s, True, lds)).data_ptr()
        wp = self.hold(ME._prep_bf16_group(w3s, False, lds)).data_ptr() if x.need else 0
        n_in, n_out = kmap.n_in, kmap.n_out
        if lds:
            plan = kmap.tile_plan(False, row_bounds)
            y = self.new(n_out, cout)
            self._tile_row(self.f, xg, wt, plan, y.p, n_in, cin, cout, False, 0, P, GC._ksplit(plan, K), G)
        elif GC._tiled(kmap, P, K, closed):
            tl, ntl = kmap.tiles(row_bounds)
            y = self.new(n_out, cout)
            self.f.add(OP_SPCONV_FWD_TILED, xg, wt, kmap.nbr.data_ptr(), tl.data_ptr(), ntl, 0, y.p, n_in, n_out, K, cin * self.kx, cout, 2)
        else:
            seg, nseg = kmap.segments(ME._seg_len_fwd(), row_bounds)
            y = T(self.alloc(max(n_out, 1) * cout * 4, R_ZF), n_out, cout)          # atomic scatter into zeros
            self.f.add(OP_PAIRS_FWD, xg, wt, pin.data_ptr(), pout.data_ptr(), seg.data_ptr(), nseg, 0, y.p, n_out, cin * self.kx, cout, 2, 1)
        self.tape.append(lambda: self._gconv_bwd(x, y, weights, kmap, row_bounds, closed, G, K, cin, cout, P, wp, lds))
        return y

    def _gconv_bwd(self, x, y, weights, kmap, rb, closed, G, K, cin, cout, P, wp, lds):
        g = self.grad(y, want16=True)
        if g is None:
            return
        GC = ME.GroupedConvFunction
        dy16 = g[1]
        pin, pout = kmap.pairs(rb)[:2]
        if x.need:
            if lds:
                plan = kmap.tile_plan(True, rb)
                dx = self.alloc(max(kmap.n_in, 1) * cin * 4)
                self._tile_row(self.b, dy16, wp, plan, dx, kmap.n_out, cout, cin, kmap.symmetric, 0, P, GC._ksplit(plan, K), G)
            elif GC._tiled(kmap, P, K, closed):
                tl, ntl = kmap.tiles(rb)
                dx = self.alloc(max(kmap.n_in, 1) * cin * 4)
                self.b.add(OP_SPCONV_FWD_TILED, dy16, wp, kmap.nbrT.data_ptr(), tl.data_ptr(), ntl, 0, dx, kmap.n_out, kmap.n_in, K,
                           cout * self.kx, cin, 2)
            else:
                seg, nseg = kmap.segments(ME._seg_len_fwd(), rb)
                dx = self.alloc(max(kmap.n_in, 1) * cin * 4, R_ZB)
                self.b.add(OP_PAIRS_FWD, dy16, wp, pout.data_ptr(), pin.data_ptr(), seg.data_ptr(), nseg, 0, dx, kmap.n_in, cout * self.kx, cin, 2, 1)
            self.gadd(x, dx)
        seg, nseg = kmap.segments(ME._wgrad_seg_len(P, cin, cout, 1, G * K), rb)
        self.wgrad_row(OP_PAIRS_WGRAD, self.rows16(x) if not x.p16 else x.p16, dy16, pin.data_ptr(), pout.data_ptr(), seg.data_ptr(), nseg,
                       self.pgrad_group(weights), G * K, cin, cout, (3 if self.kx == 3 else 2) | WGRAD_ACC)

    def gbn_act(self, x, bns, bounds, act):
        """me.FusedBNActFunction over G row groups; the G modules' parameters and running statistics as [G, C] arrays (me.BNStack)."""
        n, c, G = x.n, x.c, len(bns)
        b0 = bns[0]
        if not (b0.training and ME.GROUPED_BN_STACK and ME._stackable(bns) and c % 4 == 0) or ME._sync_group_of(b0) is not None:
            raise NotReady("grouped BatchNorm form without a program counterpart")
        st = self.hold(ME.BNStack.of(bns))
        red, nred, _, group_n, app, napp, _ = self._hold(ME._bn_chunks(tuple(bounds), self.dev, c))
        sums = self.alloc(ME.BN_SLOTS * 2 * G * c * 4, R_ZF)
        self.f.add(OP_BN_SUMS, x.p, red.data_ptr(), nred, G, c, sums)
        mv = self.alloc(2 * G * c * 4, R_ZF)
        mean, var = mv, mv + G * c * 4
        y = self.new(n, c)
        if self.want16(c):
            y.p16 = self.alloc(max(n, 1) * c * 2)
        self.f.add(OP_BN_APPLY_SUMS, x.p, 0, app.data_ptr(), napp, G, c, sums, group_n.data_ptr(), _fbits(b0.eps),
                   st.weight.data_ptr(), st.bias.data_ptr(), act, y.p, y.p16, mean, var, st.running_mean.data_ptr(),
                   st.running_var.data_ptr(), st.num_batches_tracked.data_ptr(), _fbits(b0.momentum))
        self.tape.append(lambda: self._gbn_bwd(x, y, bns, st, act, mean, var, G, (red, nred, group_n, app, napp)))
        return y

    def _gbn_bwd(self, x, y, bns, st, act, mean, var, G, ch):
        g = self.grad(y)
        if g is None:
            return
        dy = g[0]
        red, nred, group_n, app, napp = ch
        n, c = x.n, x.c
        eps = _fbits(bns[0].eps)
        dsums = self.alloc(ME.BN_SLOTS * 2 * G * c * 4, R_ZB)
        self.b.add(OP_BN_BWD_SUMS, dy, x.p, y.p, red.data_ptr(), nred, G, c, mean, var, eps, act, dsums)
        dx = self.alloc(max(n, 1) * c * 4)
        dx16 = self.alloc(max(n, 1) * c * 2) if (x.need and self.want16(c)) else 0
        self.b.add(OP_BN_BWD_APPLY_SUMS, dy, x.p, y.p, app.data_ptr(), napp, G, c, mean, var, eps, st.weight.data_ptr(), dsums,
                   group_n.data_ptr(), act, 1, dx, dx16, 0, self.pgrad_group([b.bias for b in bns]),
                   self.pgrad_group([b.weight for b in bns]))
        self.gadd(x, dx, dx16)

    # ---------------------------------------------------------------- 1x1x1 convolution (me.LinearFunction)
    def linear(self, x, weight, cin, cout):
        n = x.n
        own = ME.LinearFunction._own(n, cin, cout) and self.lib.device_kernels
        w2 = weight.view(cin, cout)
        if own:
            wt, wp = self._planned(w2.view(1, cin, cout), True, x.need)
            x16 = self.rows16(x)
            units, nchunk = -(-n // 128) * (cout // (128 if cout % 128 == 0 else 64)), cin * self.kx // 64
            ksplit = min(nchunk, 64 // max(units, 1)) if (units <= 16 and nchunk >= 4) else 1
            # (a split contraction meets in fp32 partial products: its output stays fp32 rows)
            y = self.out(n, cout) if (ksplit == 1 and n >= ACT16_MIN_ROWS) else self.new(n, cout)
            part = 0
            if ksplit > 1:
                part = self.alloc(ksplit * max(n, 1) * cout * 4)
            elif ME.WANT_BN_STATS and ME.FUSED_BN_STATS and cout <= 1024:
                y.stats = self.alloc(ME.BN_SLOTS * 2 * cout * 4, R_ZF)
            self.f.add(OP_LINEAR_FWD, x16, wt, 0, y.p or y.p16, n, cin * self.kx, cout, max(ksplit, 1) | (0 if y.p else LIN_OUT16), y.stats, part)
        elif self.bf16 and self.kx == 1 and ME._use_bf16(cin) and ME.LinearFunction._skinny(n, cin, cout):
            # many rows x few output channels in the bench precision (the vote offsets: 64 -> 3): me.LinearFunction._rows_gemm,
            # i.e. the pair kernel on the identity map with fp32 rows rounded on the fly and the bf16 copy of the weights
            wt, _ = self._planned(w2.view(1, cin, cout), False, False)
            ar, seg, nseg = self._ident(n, 128)
            y = T(self.alloc(max(n, 1) * cout * 4, R_ZF), n, cout)
            self.f.add(OP_PAIRS_FWD, self.f32(x), wt, ar.data_ptr(), ar.data_ptr(), seg.data_ptr(), nseg, 0, y.p, n, cin, cout, 1, 1)
            wp = 0
        else:
            # generic form (fp32 parity mode / the oracle): the pair kernel on the identity map, me.LinearFunction._rows_gemm
            ar, seg, nseg = self._ident(n, 128 if self.lib.is_device else (1 << 30))
            y = T(self.alloc(max(n, 1) * cout * 4, R_ZF), n, cout)
            self.f.add(OP_PAIRS_FWD, self.f32(x), w2.data_ptr(), ar.data_ptr(), ar.data_ptr(), seg.data_ptr(), nseg, 0, y.p, n, cin, cout, 0, 1)
            wp = 0
        self.tape.append(lambda: self._linear_bwd(x, y, weight, w2, cin, cout, own, wp))
        return y

    def _linear_bwd(self, x, y, weight, w2, cin, cout, own, wp):
        wprec = ME._wgrad_prec(cin, cout, bool(own and x.p16))
        g = self.grad(y, want16=own, want32=(not own) or wprec < 2)
        if g is None:
            return
        dy, dy16 = g
        n, prog = x.n, self.b
        dx16 = 0
        if x.need:
            if own:
                units, nchunk = -(-n // 128) * (cin // (128 if cin % 128 == 0 else 64)), cout * self.kx // 64
                ksplit = min(nchunk, 64 // max(units, 1)) if (units <= 16 and nchunk >= 4) else 1
                if self.act16 and ksplit == 1 and n >= ACT16_MIN_ROWS:
                    dx, dx16 = 0, self.alloc(max(n, 1) * cin * 2)
                else:
                    dx = self.alloc(max(n, 1) * cin * 4)
                part = self.alloc(ksplit * max(n, 1) * cin * 4) if ksplit > 1 else 0
                prog.add(OP_LINEAR_FWD, dy16, wp, 0, dx or dx16, n, cout * self.kx, cin, max(ksplit, 1) | (0 if dx else LIN_OUT16), 0, part)
            else:
                ar, seg, nseg = self._ident(n, 128 if self.lib.is_device else (1 << 30))
                dx = self.alloc(max(n, 1) * cin * 4, R_ZB)
                self.late.append((prog, len(prog.rows), 2, lambda w=w2: w.detach().t().contiguous()))
                prog.add(OP_PAIRS_FWD, dy, 0, ar.data_ptr(), ar.data_ptr(), seg.data_ptr(), nseg, 0, dx, n, cout, cin, 0, 1)
            self.gadd(x, dx, dx16)
        xc, dyc = x.p, dy
        if wprec >= 2:
            xc, dyc = x.p16, dy16
        elif not xc:
            raise NotReady("fp32 rows of a bf16-stored input are not kept for the backward pass")
        ar, seg, nseg = self._ident(n, ME._wgrad_seg_len(n, cin, cout, 1 if wprec else 0, 1))
        self.wgrad_row(OP_PAIRS_WGRAD, xc, dyc, ar.data_ptr(), ar.data_ptr(), seg.data_ptr(), nseg, self.pgrad(weight), 1, cin, cout,
                       wprec | WGRAD_ACC)

    # ---------------------------------------------------------------- BatchNorm (+ residual) (+ activation) (me.FusedBNActFunction)
    def bn_act(self, x, bn, act, res=None):
        n, c = x.n, x.c
        if not (bn.training and bn.track_running_stats and bn.momentum is not None and c % 4 == 0) or ME._sync_group_of(bn) is not None:
            raise NotReady("BatchNorm form without a program counterpart (evaluation statistics, --sync_bn)")
        red, nred, _, group_n, app, napp, _ = self._chunks(n, c)
        prog = self.f
        s16 = self.s16ok(c, n)
        sums = x.stats
        if not sums:
            # (the statistics pass reads fp32 rows: a convolution that left none also left its statistics; what remains are
            # the few small tensors of DAPPM that a BatchNorm reads straight from a join)
            sums = self.alloc(ME.BN_SLOTS * 2 * c * 4, R_ZF)
            prog.add(OP_BN_SUMS, self.f32(x), red.data_ptr(), nred, 1, c, sums)
        mv = self.alloc(2 * c * 4, R_ZF)
        mean, var = mv, mv + c * 4
        gamma, beta = bn.weight, bn.bias
        if s16:
            y = self.new16(n, c)
            prog.add(OP_BN_APPLY_SUMS, self.rows16(x), self.rows16(res) if res is not None else 0, app.data_ptr(), napp, 1, c, sums,
                     group_n.data_ptr(), _fbits(bn.eps), gamma.data_ptr(), beta.data_ptr(), act | S16, y.p16, 0, mean, var,
                     bn.running_mean.data_ptr(), bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(), _fbits(bn.momentum))
            self.tape.append(lambda: self._bn_bwd16(x, y, res, bn, act, mean, var, (red, nred, group_n, app, napp)))
            return y
        y = self.new(n, c)
        if self.want16(c):
            y.p16 = self.alloc(max(n, 1) * c * 2)
        prog.add(OP_BN_APPLY_SUMS, self.f32(x), self.f32(res) if res is not None else 0, app.data_ptr(), napp, 1, c, sums, group_n.data_ptr(),
                 _fbits(bn.eps), gamma.data_ptr(), beta.data_ptr(), act, y.p, y.p16, mean, var, bn.running_mean.data_ptr(),
                 bn.running_var.data_ptr(), bn.num_batches_tracked.data_ptr(), _fbits(bn.momentum))
        self.tape.append(lambda: self._bn_bwd(x, y, res, bn, act, mean, var, (red, nred, group_n, app, napp)))
        return y

    def _bn_bwd16(self, x, y, res, bn, act, mean, var, ch):
        """bf16 storage: dY, X, Y in, dX (and dResidual) out as bf16 rows; statistics and parameter gradients fp32."""
        g = self.grad(y, want16=True)
        if g is None:
            return
        dy16 = g[1]
        red, nred, group_n, app, napp = ch
        n, c, prog = x.n, x.c, self.b
        dsums = self.alloc(ME.BN_SLOTS * 2 * c * 4, R_ZB)
        eps = _fbits(bn.eps)
        prog.add(OP_BN_BWD_SUMS, dy16, x.p16, y.p16, red.data_ptr(), nred, 1, c, mean, var, eps, act | S16, dsums)
        dx16 = self.alloc(max(n, 1) * c * 2)               # (the kernel always writes dX)
        dres16 = self.alloc(max(n, 1) * c * 2) if (res is not None and res.need) else 0
        prog.add(OP_BN_BWD_APPLY_SUMS, dy16, x.p16, y.p16, app.data_ptr(), napp, 1, c, mean, var, eps, bn.weight.data_ptr(), dsums,
                 group_n.data_ptr(), act | S16, 1, dx16, 0, dres16, self.pgrad(bn.bias), self.pgrad(bn.weight))
        self.gadd(x, 0, dx16)
        if dres16:
            self.gadd(res, 0, dres16)

    def _bn_bwd(self, x, y, res, bn, act, mean, var, ch):
        g = self.grad(y, want32=True)
        if g is None:
            return
        dy = g[0]
        red, nred, group_n, app, napp = ch
        n, c, prog = x.n, x.c, self.b
        dsums = self.alloc(ME.BN_SLOTS * 2 * c * 4, R_ZB)
        eps = _fbits(bn.eps)
        prog.add(OP_BN_BWD_SUMS, dy, x.p, y.p, red.data_ptr(), nred, 1, c, mean, var, eps, act, dsums)
        dx = self.alloc(max(n, 1) * c * 4) if x.need else 0
        dx16 = self.alloc(max(n, 1) * c * 2) if (x.need and self.want16(c)) else 0
        dres = self.alloc(max(n, 1) * c * 4) if (res is not None and res.need) else 0
        if not dx:
            dx = self.alloc(max(n, 1) * c * 4)          # (the kernel always writes dX)
        prog.add(OP_BN_BWD_APPLY_SUMS, dy, x.p, y.p, app.data_ptr(), napp, 1, c, mean, var, eps, bn.weight.data_ptr(), dsums,
                 group_n.data_ptr(), act, 1, dx, dx16, dres, self.pgrad(bn.bias), self.pgrad(bn.weight))
        self.gadd(x, dx, dx16)
        if dres:
            self.gadd(res, dres)

    # ---------------------------------------------------------------- relu(a [+ b]) / a + b
    def add_act(self, a, b, act):
        need = a.need or (b is not None and b.need)
        if self.s16ok(a.c, a.n):
            y = self.new16(a.n, a.c, need=need)
            self._add_rows(self.f, self.rows16(a), self.rows16(b) if b is not None else 0, y.p16, a.n, a.c, act, 0, s16=True)
            self.tape.append(lambda: self._add_act_bwd(a, b, y, act))
            return y
        y = self.new(a.n, a.c, need=need)
        if act == ME.ACT_RELU and self.want16(a.c):
            y.p16 = self.alloc(max(a.n, 1) * a.c * 2)
        self._add_rows(self.f, self.f32(a), self.f32(b) if b is not None else 0, y.p, a.n, a.c, act, y.p16)
        self.tape.append(lambda: self._add_act_bwd(a, b, y, act))
        return y

    def _add_act_bwd(self, a, b, y, act):
        s16 = not y.p                                    # the join was made on bf16 rows
        g = self.grad(y, want16=s16, want32=not s16)
        if g is None:
            return
        dy = g[0]
        if act == ME.ACT_NONE:
            self.gadd(a, dy, g[1])                      # the same rows are the gradient of both addends
            if b is not None:
                self.gadd(b, dy, g[1])
            return
        # relu: dz = dy where y > 0 -- cg3d_bn_bwd_apply with the identity normalisation (use_batch_stats = 0: dx = dz)
        n, c = a.n, a.c
        z, o = self._unit(c)
        ch = self._chunks(n, c)
        if s16:
            dz16 = self.alloc(max(n, 1) * c * 2)
            self.b.add(OP_BN_BWD_APPLY, g[1], y.p16, y.p16, ch[4].data_ptr(), ch[5], c, z, o, _fbits(0.0), o, z, z, o, act | S16, 0, dz16, 0, 0)
            self.gadd(a, 0, dz16)
            if b is not None:
                self.gadd(b, 0, dz16)
            return
        dz = self.alloc(max(n, 1) * c * 4)
        self.b.add(OP_BN_BWD_APPLY, dy, y.p, y.p, ch[4].data_ptr(), ch[5], c, z, o, _fbits(0.0), o, z, z, o, act, 0, dz, 0, 0)
        self.gadd(a, dz)
        if b is not None:
            self.gadd(b, dz)

    # ---------------------------------------------------------------- interpolation (SparseTensor.features_at_coordinates)
    def interp(self, x, idx, w, nq):
        y = self.new(nq, x.c, need=x.need)
        self.f.add(OP_INTERP_FWD, self.f32(x), idx.data_ptr(), w.data_ptr(), y.p, nq, x.c)
        self.tape.append(lambda: self._interp_bwd(x, y, idx, w, nq))
        return y

    def _interp_bwd(self, x, y, idx, w, nq):
        g = self.grad(y, want32=True)
        if g is None or not x.need:
            return
        df = self.alloc(max(x.n, 1) * x.c * 4, R_ZB)
        self.b.add(OP_INTERP_BWD, g[0], idx.data_ptr(), w.data_ptr(), df, nq, x.c)
        self.gadd(x, df)

    # ---------------------------------------------------------------- average pooling (me.ScatterMeanFunction)
    def scatter_mean(self, x, smap, n_out):
        J, n_in = smap.shape
        y = self.new(n_out, x.c, need=x.need)
        cnt = self.alloc(max(n_out, 1) * 4)
        # (the call zero-fills `out` and `cnt` itself)
        self.f.add(OP_SCATTER_MEAN_FWD, self.f32(x), smap.data_ptr(), J, y.p, cnt, n_in, n_out, x.c)
        self.tape.append(lambda: self._scatter_mean_bwd(x, y, smap, cnt, n_out))
        return y

    def _scatter_mean_bwd(self, x, y, smap, cnt, n_out):
        g = self.grad(y, want32=True)
        if g is None or not x.need:
            return
        J, n_in = smap.shape
        df = self.alloc(max(n_in, 1) * x.c * 4)
        self.b.add(OP_SCATTER_MEAN_BWD, g[0], cnt, smap.data_ptr(), J, df, n_in, n_out, x.c)
        self.gadd(x, df)

    # ---------------------------------------------------------------- channel concatenation
    def cat(self, ts):
        n, c = ts[0].n, sum(t.c for t in ts)
        s16 = self.act16 and n >= ACT16_MIN_ROWS and all(t.c % 8 == 0 for t in ts)
        w = 2 if s16 else 4                                # bytes per element of the rows that are joined
        y = self.new16(n, c, need=any(t.need for t in ts)) if s16 else self.new(n, c, need=any(t.need for t in ts))
        o = 0
        for t in ts:
            self.f.add(OP_COPY2D, (y.p16 if s16 else y.p) + o * w, c * w, self.rows16(t) if s16 else self.f32(t), t.c * w, t.c * w, n)
            o += t.c
        self.tape.append(lambda: self._cat_bwd(ts, y))
        return y

    def _cat_bwd(self, ts, y):
        s16 = not y.p
        g = self.grad(y, want16=s16, want32=not s16)
        if g is None:
            return
        o, c, w = 0, y.c, (2 if s16 else 4)
        src = g[1] if s16 else g[0]
        for t in ts:
            if t.need:
                d = self.alloc(max(t.n, 1) * t.c * w)
                self.b.add(OP_COPY2D, d, t.c * w, src + o * w, c * w, t.c * w, t.n)
                self.gadd(t, 0 if s16 else d, d if s16 else 0)
            o += t.c

    # ---------------------------------------------------------------- backward emission
    def mark(self, name):
        """A named point of the forward pass: when the backward pass has consumed everything recorded AFTER it (i.e. the
        gradients of every layer after this point are complete), `marks[name]` = the backward row reached."""
        self.mark_at[len(self.tape)] = name

    def emit_backward(self, out):
        outs = out if isinstance(out, (list, tuple)) else [out]
        for o, region in zip(outs, (R_DOUT, R_DOUT2)):
            o.gc = [(region, 0)]
        for i in range(len(self.tape) - 1, -1, -1):
            lane, fn = self.tape[i]
            self.set_lane(lane)                     # a layer's backward rows run on the lane of its forward rows
            fn()
            name = self.mark_at.get(i)
            if name is not None:
                self.marks[name] = len(self.b.rows)
        self.set_lane(0)
        self.tape = None


# ------------------------------------------------------------------------------------------------ module walkers
class _X:
    """A sparse tensor inside a program: features + the coordinate map they live on."""
    __slots__ = ("t", "key")

    def __init__(self, t, key):
        self.t, self.key = t, key


class Emitter:
    """Walks the BiResNet module tree (cagroup3d_amd/pcdet/models/backbones_3d/biresnet.py) like its forward methods do."""

    def __init__(self, builder, mgr):
        self.b, self.mgr = builder, mgr
        self._interp = {}
        self._coordsf = {}

    # -- layers
    def conv(self, m, x):
        mgr = self.mgr
        assert m.bias is None, "the backbone's convolutions carry no bias"
        if isinstance(m, ME.MinkowskiConvolutionTranspose):
            in_ts = mgr.get(x.key).tensor_stride
            out_ts = in_ts // m.stride
            cands = [k for k in mgr._maps if k.tensor_stride == out_ts and mgr._strided.get((k, m.stride)) == x.key]
            assert cands, "transposed convolution needs the finer map it was strided from"
            out_key = cands[0]
            km = mgr.kernel_map(x.key, out_key, m.kernel_size, m.dilation, True)
            return _X(self.b.conv(x.t, m.kernel, km, m.kernel_volume, m.in_channels, m.out_channels), out_key)
        out_key = mgr.stride(x.key, m.stride) if m.stride > 1 else x.key
        if m.kernel_volume == 1 and m.stride == 1:
            return _X(self.b.linear(x.t, m.kernel, m.in_channels, m.out_channels), out_key)
        km = mgr.kernel_map(x.key, out_key, m.kernel_size, m.dilation, False)
        return _X(self.b.conv(x.t, m.kernel, km, m.kernel_volume, m.in_channels, m.out_channels), out_key)

    def bn(self, m, x, act=ME.ACT_NONE, residual=None):
        return _X(self.b.bn_act(x.t, m.bn, act, residual.t if residual is not None else None), x.key)

    def relu(self, x):
        return _X(self.b.add_act(x.t, None, ME.ACT_RELU), x.key)

    def add_relu(self, a, b):
        assert a.key == b.key
        return _X(self.b.add_act(a.t, b.t, ME.ACT_RELU), a.key)

    def add(self, a, b):
        assert a.key == b.key
        return _X(self.b.add_act(a.t, b.t, ME.ACT_NONE), a.key)

    def pool(self, m, x):
        pmap, out_key = m.pool_map(self.mgr, x.key)
        return _X(self.b.scatter_mean(x.t, pmap, self.mgr.get(out_key).n), out_key)

    def at_coordinates(self, x, dst_key):
        """x interpolated at the voxel positions of the map `dst_key` (features_at_coordinates(dst.C.float()))."""
        ck = (x.key, dst_key)
        hit = self._interp.get(ck)
        if hit is None:
            dst = self.mgr.get(dst_key)
            q = self._coordsf.get(dst_key)
            if q is None:
                q = self._coordsf[dst_key] = dst.coords.float().contiguous()
            hit = self._interp[ck] = ME.interp_tables(self.mgr.get(x.key), q)
            self.b.keep.append(hit)
        idx, w = hit
        return _X(self.b.interp(x.t, idx, w, idx.shape[0]), dst_key)

    def seq(self, mods, x):
        """me.Sequential.forward: BatchNorm followed by ReLU is one fused launch pair."""
        mods = list(mods)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, ME.MinkowskiBatchNorm) and isinstance(nxt, (ME.MinkowskiReLU, ME.MinkowskiELU)):
                x = self.bn(m, x, ME.ACT_RELU if isinstance(nxt, ME.MinkowskiReLU) else ME.ACT_ELU)
                i += 2
                continue
            x = self.module(m, x)
            i += 1
        return x

    def module(self, m, x):
        from .pcdet.models.backbones_3d import biresnet as B
        if isinstance(m, (ME.MinkowskiConvolution, ME.MinkowskiConvolutionTranspose)):
            return self.conv(m, x)
        if isinstance(m, ME.MinkowskiBatchNorm):
            return self.bn(m, x)
        if isinstance(m, ME.MinkowskiReLU):
            return self.relu(x)
        if isinstance(m, ME.MinkowskiAvgPooling):
            return self.pool(m, x)
        if isinstance(m, torch.nn.Sequential):
            return self.seq(m, x)
        if isinstance(m, B.BasicBlock):
            out = self.conv(m.conv2, self.bn(m.norm1, self.conv(m.conv1, x), ME.ACT_RELU))
            res = x if m.downsample is None else self.module(m.downsample, x)
            return self.bn(m.norm2, out, ME.ACT_NONE if m.no_relu else ME.ACT_RELU, res)
        if isinstance(m, B.Bottleneck):
            out = self.bn(m.norm1, self.conv(m.conv1, x), ME.ACT_RELU)
            out = self.bn(m.norm2, self.conv(m.conv2, out), ME.ACT_RELU)
            res = x if m.downsample is None else self.module(m.downsample, x)
            return self.bn(m.norm3, self.conv(m.conv3, out), ME.ACT_NONE if m.no_relu else ME.ACT_RELU, res)
        if isinstance(m, B.DAPPM):
            # ~75 launches of a few microseconds on 150-1 300 voxels, one behind the other in the module's order -- and the
            # whole device waits for them (the stride-4 chain needs their sum).  Only the `process` convolutions form a chain:
            # the four pooled branches (pool, BN, ReLU, 1x1x1, interpolation: 9 launches each) and the shortcut depend on x
            # alone and go to the lanes of DAPPM_LANES, round robin.
            here = self.b.lane
            side = [l for l in DAPPM_LANES if l != here] if (LANES and here) else []
            ups = []
            for i, scale in enumerate((m.scale1, m.scale2, m.scale3, m.scale4)):
                if side:
                    self.b.set_lane(side[i % len(side)])
                ups.append(self.at_coordinates(self.module(scale, x), x.key))
            if side:
                self.b.set_lane(side[len(ups) % len(side)])
            short = self.module(m.shortcut, x)
            self.b.set_lane(here)
            feats = [self.module(m.scale0, x)]
            for up, process in zip(ups, (m.process1, m.process2, m.process3, m.process4)):
                feats.append(self.module(process, self.add(up, feats[-1])))
            cat = _X(self.b.cat([f.t for f in feats]), x.key)
            return self.add(self.module(m.compression, cat), short)
        raise NotReady("no program form for %s" % type(m).__name__)

    def biresnet(self, net, x, mid_mark=False):
        """BiResNet.forward (cagroup3d_amd/pcdet/models/backbones_3d/biresnet.py; reference biresnet.py:358-406)."""
        x = self.module(net.conv1, x)
        l1 = self.module(net.layer1, x)
        l2 = self.module(net.layer2, self.relu(l1))
        if mid_mark:
            self.b.mark("mid")           # every deeper layer's gradient is complete when the backward pass is back here
        r2 = self.relu(l2)
        # From here to the last join the net is two chains that meet three times: the stride-8 / 16 / 32 layers (lane 1: 23 k, 5 k
        # and 1 k voxels for 4 x S50k -- launches of 40-360 workgroup units and DAPPM's ~60 launches of a few microseconds) and
        # the stride-4 layers (lane 0: 82 k voxels, 640 units per launch).  Every cross term sits on the lane of its CONSUMER's
        # chain: down3 / down4 read the stride-4 rows and feed the coarse chain (lane 1), the interpolations read the coarse
        # rows and feed the stride-4 chain (lane 0).
        lane = self.b.set_lane
        lane(1)
        l3 = self.module(net.layer3, r2)
        r3 = self.relu(l3)
        c3 = self.module(net.compression3, r3)
        lane(0)
        hi = self.module(net.layer3_, r2)
        rh = self.relu(hi)
        lane(1)
        lo_r = self.add_relu(l3, self.module(net.down3, rh))
        l4 = self.module(net.layer4, lo_r)
        r4 = self.relu(l4)
        c4 = self.module(net.compression4, r4)
        lane(0)
        hi_r = self.add_relu(hi, self.at_coordinates(c3, hi.key))
        hi = self.module(net.layer4_, hi_r)
        rh = self.relu(hi)
        lane(1)
        lo_r = self.add_relu(l4, self.module(net.down4, rh))
        sp = self.module(net.spp, self.module(net.layer5, lo_r))
        lane(0)
        hi_r = self.add_relu(hi, self.at_coordinates(c4, hi.key))
        hi = self.module(net.layer5_, hi_r)
        hi = self.add(hi, self.at_coordinates(sp, hi.key))
        return self.module(net.out, hi)


# ------------------------------------------------------------------------------------------------ lanes
def _schedule(prog, starts, cuts=()):
    """Event edges between the lanes of `prog` (rows still with region-relative addresses), in place.

    Rows of one lane run in table order on their queue; a row needs an edge from ANOTHER lane when it reads a block that lane
    wrote last, or writes a block that lane read or wrote (ROLES).  A block is what one `Builder.alloc` returned (`starts`) or,
    outside the regions (module buffers such as running statistics), an address.  An edge is an EVENT_RECORD right behind the
    producing row on its lane and an EVENT_WAIT in front of the consuming row on its lane; a lane that already waited for a
    later row of the other one needs none.  Every part of the table (cuts: row indices where the caller splits it to run a
    callback between two halves of a backward pass) has every other lane wait for lane 0's position at the start of the part
    before its first row, and ends with lane 0 waiting for the end of every other lane: what precedes the table on the stream
    (the zero-fill of its regions) and what follows it (the framework's reads of the outputs) see one queue.

    Events are numbered slots (row = opcode, slot, 1): `_bind_events` puts the handles in when the pass is run.
    Returns (old -> new row index, cut -> new row index at which the table is split there: behind the join of the part that
    ends at the cut, in front of what the next part starts with)."""
    import bisect
    rows, lanes = prog.rows, prog.lanes
    n = len(rows)
    if not any(lanes):
        return list(range(n)), {c: c for c in cuts}

    def block(addr):
        tag = addr >> TAG
        if not tag:
            return addr
        lst = starts.get(tag << TAG)
        off = addr - (tag << TAG)
        if not lst:
            return (tag, 0)
        return (tag, lst[bisect.bisect_right(lst, off) - 1])

    reads, writes = [None] * n, [None] * n
    for i in range(n):
        row = rows[i]
        rd, wr = ROLES[int(row[0]) & OPCODE_MASK]
        reads[i] = [block(row[c]) for c in rd if c < len(row) and row[c]]
        writes[i] = [block(row[c]) for c in wr if c < len(row) and row[c]]
    NL = max(lanes) + 1
    assert NL <= MAX_LANES
    before = [[] for _ in range(n + 1)]          # rows put in front of old row i
    after = [[] for _ in range(n)]               # rows put right behind old row i
    tail = [[] for _ in range(n + 1)]            # the join of the part that ends in front of old row i (index n: end of table)
    nev = 0
    bounds = [0] + sorted(set(c for c in cuts if 0 < c < n)) + [n]
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        lastw, lastr = {}, {}                    # block -> per lane: sequence number of its last writer / reader there
        cnt = [0] * NL                           # rows seen per lane
        where = [[None] for _ in range(NL)]      # lane, sequence number -> old row index
        seen = [[0] * NL for _ in range(NL)]     # seen[L][M]: lane L has waited for row number seen[L][M] of lane M
        forked = [True] + [False] * (NL - 1)
        fork_ev = None
        for i in range(lo, hi):
            L = lanes[i]
            need = [0] * NL
            blocks_r, blocks_w = reads[i], writes[i]
            for bk in blocks_r:
                w = lastw.get(bk)
                if w is not None:
                    for M in range(NL):
                        if w[M] > need[M]:
                            need[M] = w[M]
            for bk in blocks_w:
                for t in (lastw.get(bk), lastr.get(bk)):
                    if t is not None:
                        for M in range(NL):
                            if t[M] > need[M]:
                                need[M] = t[M]
            if not forked[L]:
                # the lane's first row of this part: behind everything lane 0 was given before the part (and the zero-fill)
                if fork_ev is None:
                    fork_ev = nev
                    nev += 1
                    before[lo].insert(0, ((OP_EVENT_RECORD, fork_ev, 1), 0))
                before[i].append(((OP_EVENT_WAIT, fork_ev, 1), L))
                forked[L] = True
            for M in range(NL):
                if M != L and need[M] > seen[L][M]:
                    after[where[M][need[M]]].append(((OP_EVENT_RECORD, nev, 1), M))
                    before[i].append(((OP_EVENT_WAIT, nev, 1), L))
                    nev += 1
                    seen[L][M] = need[M]
            cnt[L] += 1
            where[L].append(i)
            q = cnt[L]
            for bk in blocks_r:
                lastr.setdefault(bk, [0] * NL)[L] = q
            for bk in blocks_w:
                lastw.setdefault(bk, [0] * NL)[L] = q
                lastr.setdefault(bk, [0] * NL)[L] = q
        for M in range(1, NL):                   # the join: lane 0 behind the last row of every other lane
            if cnt[M] > seen[0][M]:
                after[where[M][cnt[M]]].append(((OP_EVENT_RECORD, nev, 1), M))
                tail[hi].append(((OP_EVENT_WAIT, nev, 1), 0))
                nev += 1
    # Issue order.  The table is also the order in which the HOST hands the rows to the queues, one foreign launch call after
    # the other (~5 us each): in emission order the ~140 launches of DAPPM (lane 1) would all be handed over before the
    # handful of stride-4 launches that can run beside them (lane 0) -- by then the device is through with DAPPM and nothing
    # overlaps.  Inside a part the lanes' row sequences are therefore merged round robin, one row of every lane that can
    # advance in turn (a WAIT advances once its RECORD has been placed: hipStreamWaitEvent refers to the record call before it
    # on the host); every lane keeps its own order, so the table is still a valid sequential order (the oracle runs it as one).
    out_rows, out_lanes, index, cutmap = [], [], [None] * n, {}
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        for r, l in tail[lo]:                    # (the join of the part that ended here)
            out_rows.append(r)
            out_lanes.append(l)
        cutmap[lo] = len(out_rows)
        queues = [[] for _ in range(NL)]
        for i in range(lo, hi):
            for r, l in before[i]:
                queues[l].append((r, None))
            queues[lanes[i]].append((rows[i], i))
            for r, l in after[i]:
                queues[l].append((r, None))
        pos, placed = [0] * NL, set()
        left = sum(len(q) for q in queues)
        while left:
            moved = False
            for l in range(NL):
                if pos[l] >= len(queues[l]):
                    continue
                r, orig = queues[l][pos[l]]
                op = int(r[0]) & OPCODE_MASK
                if orig is None and op == OP_EVENT_WAIT and r[1] not in placed:
                    continue
                if orig is None and op == OP_EVENT_RECORD:
                    placed.add(r[1])
                if orig is not None:
                    index[orig] = len(out_rows)
                out_rows.append(r)
                out_lanes.append(l)
                pos[l] += 1
                left -= 1
                moved = True
            assert moved, "lanes deadlocked: a wait precedes its record on every lane"
    for r, l in tail[n]:
        out_rows.append(r)
        out_lanes.append(l)
    cutmap[n] = len(out_rows)
    prog.rows, prog.lanes, prog.nevents = out_rows, out_lanes, nev
    prog.prof = [(index[r],) + tuple(rest) for (r, *rest) in prog.prof]
    return index, {c: cutmap[c] for c in cuts}


# The derivation in the library (cg3d_program_schedule, include/cagroup3d_program.h): the same algorithm as `_schedule` above --
# which stays as its specification (CG3D_SCHED_NATIVE=0 runs it; tests/test_engine_lanes.py compares the two row for row) --
# in ~50 us and outside the interpreter lock.  In Python it cost the thread that compiles the next batch's program 3-4 ms per
# step, and with the step bound by what the two host threads get done under one lock that was most of what the lanes had won.
SCHED_NATIVE = os.environ.get("CG3D_SCHED_NATIVE", "1") != "0"


def _schedule_native(lib, prog, starts, cuts=()):
    """`_schedule` by the library; leaves the scheduled table in `prog.scheduled`."""
    import ctypes
    n = len(prog.rows)
    if not any(prog.lanes):
        return list(range(n)), {c: c for c in cuts}
    P = np.ascontiguousarray(prog.table())
    nreg = 16                                               # CG3D_PROG_REGIONS
    first = np.zeros(nreg + 1, dtype=np.int64)
    parts = []
    for r in range(nreg):
        lst = starts.get(r << TAG) or ()
        first[r + 1] = first[r] + len(lst)
        if lst:
            parts.append(np.asarray(lst, dtype=np.int64))
    st = np.concatenate(parts) if parts else np.zeros(1, dtype=np.int64)
    cs = np.asarray(sorted(cuts), dtype=np.int64)
    NL = max(prog.lanes) + 1
    cap = n * (1 + 2 * NL) + 16 * (len(cs) + 2)
    out = np.empty((cap, STRIDE), dtype=np.int64)
    index = np.empty(max(n, 1), dtype=np.int64)
    cidx = np.empty(max(len(cs), 1), dtype=np.int64)
    n_out, n_ev = ctypes.c_int64(0), ctypes.c_int64(0)
    rc = lib.raw("cg3d_program_schedule")(P.ctypes.data, n, st.ctypes.data, first.ctypes.data, cs.ctypes.data if len(cs) else None, len(cs),
                                          out.ctypes.data, cap, index.ctypes.data, cidx.ctypes.data,
                                          ctypes.cast(ctypes.pointer(n_out), ctypes.c_void_p), ctypes.cast(ctypes.pointer(n_ev), ctypes.c_void_p))
    if rc != 0:
        raise _lib.CG3DError("cg3d_program_schedule failed with status %d" % rc)
    prog.scheduled = out[:n_out.value].copy()
    prog.nevents = int(n_ev.value)
    idx = index[:n].tolist()
    prog.prof = [(idx[r],) + tuple(rest) for (r, *rest) in prog.prof]
    return idx, {int(c): int(i) for c, i in zip(cs.tolist(), cidx.tolist())}


_EVENT_POOL = __import__("threading").local()


def _event_pool(lib, nevents):
    """(handles, int64 array of them) of at least `nevents` ordering events of the calling thread for (`lib`, current device): an
    event belongs to the device that was current when it was created, so a thread that issues passes on two devices keeps one
    pool per device."""
    dev = torch.cuda.current_device() if lib.is_device else -1
    pools = getattr(_EVENT_POOL, "pools", None)
    if pools is None or _EVENT_POOL.lib is not lib:
        pools, _EVENT_POOL.lib = {}, lib
        _EVENT_POOL.pools = pools
    ent = pools.get(dev)
    if ent is None:
        ent = pools[dev] = [[], None]
    pool = ent[0]
    grown = False
    while len(pool) < nevents:
        h = ctypes_i64()
        lib.call("cg3d_event_create_sync", ctypes_ref(h))
        pool.append(h.value)
        grown = True
    if ent[1] is None or grown:
        ent[1] = np.asarray(pool, dtype=np.int64)
    return pool, ent[1]


def _bind_events(P, lib, nevents):
    """Handles of ordering events (cg3d_event_create_sync, created once per thread and reused by every pass: a wait refers to
    the record issued before it, and one thread issues one pass at a time) in place of the slot numbers of `_schedule`."""
    if not nevents:
        return P
    pool = _event_pool(lib, nevents)[0]
    op = P[:, 0] & OPCODE_MASK
    m = ((op == OP_EVENT_RECORD) | (op == OP_EVENT_WAIT)) & (P[:, 2] == 1)
    if m.any():
        P[m, 1] = np.asarray(pool, dtype=np.int64)[P[m, 1]]
        P[m, 2] = 0
    return P


_SIDE_STREAMS = {}


def _lane_streams(lib):
    """(ctypes array, count) of the queues of lanes 0 .. MAX_LANES - 1: torch's current stream and this process's side streams of
    the device (CG3D_LANE_PRIORITY: comma-separated stream priorities of lanes 1, 2, ...; default 0).  Only the lanes in use get
    a stream of their own (the device has four hardware queues by default, and every stream created takes a turn on them: the
    main stream, the dry run's stream and three lanes already share); the others fall back to lane 0's."""
    import ctypes
    main = lib.stream()
    if not lib.is_device:
        return (ctypes.c_void_p * MAX_LANES)(*([None] * MAX_LANES)), MAX_LANES
    dev = torch.cuda.current_device()
    side = _SIDE_STREAMS.get(dev)
    if side is None:
        used = max([1, WGRAD_LANE] + DAPPM_LANES)
        prio = [int(x) for x in os.environ.get("CG3D_LANE_PRIORITY", "0").split(",")]
        side = _SIDE_STREAMS[dev] = [torch.cuda.Stream(device=dev, priority=prio[min(i, len(prio) - 1)]) for i in range(used)]
    return (ctypes.c_void_p * MAX_LANES)(main, *[x.cuda_stream for x in side], *([main] * (MAX_LANES - 1 - len(side)))), MAX_LANES


# ------------------------------------------------------------------------------------------------ a compiled pass
class Compiled:
    """Forward + backward tables of one batch, the tensors they point into, and the region sizes."""

    def __init__(self, b, out, out_key, n_in, c_in, mgr=None):
        self.mgr = mgr                   # the coordinate manager owns the maps / plans / pair lists the rows point into
        # event edges between the lanes (no-ops for a one-lane pass); row indices recorded during emission follow
        if SCHED_NATIVE:
            fidx, _ = _schedule_native(b.lib, b.f, b.starts)
            bidx, bcuts = _schedule_native(b.lib, b.b, b.starts, tuple(b.marks.values()))
        else:
            fidx, _ = _schedule(b.f, b.starts)
            bidx, bcuts = _schedule(b.b, b.starts, tuple(b.marks.values()))
        self.lanes = bool(b.f.nevents or b.b.nevents)
        self.nevents = max(b.f.nevents, b.b.nevents)
        self.fwd, self.bwd = b.f.table(), b.b.table()
        self.fprof, self.bprof = b.f.prof, b.b.prof
        self.size = dict(b.size)
        self.keep, self.params, self.late = b.keep, b.params, b.late
        self.marks = {name: bcuts[row] for name, row in b.marks.items()}
        self.keep_cached = b.keep_cached
        self.late_f = [(fidx[r], c, fn) for (pg, r, c, fn) in b.late if pg is b.f]
        self.late_b = [(bidx[r], c, fn) for (pg, r, c, fn) in b.late if pg is b.b]
        self.out = (out.p, out.n, out.c, out.p16)
        self.out_key = out_key
        self.n_in, self.c_in = n_in, c_in
        self.gen = b.gen
        self.lib = b.lib
        self.dev = b.dev
        self.stream_device = None
        self.uses_arena = b.uses_arena

    def usable(self):
        P = ME._WeightPlan
        return (not P.dirty) and P.gen == self.gen and _lib.get() is self.lib

    def check_weights_live(self):
        """The rows address the bf16 weight copies of the step's arena, which `me.prepare_weights()` refreshes at the start of a
        detector forward and `me.finish_weights()` declares stale at its end (an optimizer step may follow).  A pass issued
        outside such a forward -- BiResNet.forward called directly after optimizer.step(), a profiling tool -- would multiply by
        the PREVIOUS step's weights while the per-layer path converts on the spot: refuse, the caller takes that path."""
        if self.uses_arena and not ME._WeightPlan.live:
            raise NotReady("the step's weight arena is not live (no detector forward in progress)")


def compile_backbone(net, sp, mid_mark=False):
    """Program pair for `net` (BiResNet, training mode) on the sparse tensor `sp` (only its coordinate side and shape are
    read).  Builds -- through the coordinate manager, with its host reads -- every map, plan and table the pass needs.
    Raises NotReady when something is missing (first steps: weights not yet in the step's arena)."""
    lib = _lib.get()
    b = Builder(lib, sp.F.device if sp.F is not None else sp.C.device, ME._WeightPlan.gen)
    b.act16 = bool(ACT_BF16 and b.bf16 and b.kx == 1)
    n, c = sp.F.shape
    x = _X(T(R_IN, n, c, need=False), sp.coordinate_map_key)
    out = Emitter(b, sp.coordinate_manager).biresnet(net, x, mid_mark)
    b.f32(out.t)                 # the pass hands fp32 rows to the head (and, next to them, the bf16 rows it computed them from)
    b.emit_backward(out.t)
    return Compiled(b, out.t, out.key, n, c, sp.coordinate_manager)


def _resolve(table, bases):
    P = table.copy()
    tag = P >> TAG
    for r, base in bases.items():
        m = tag == (r >> TAG)
        if m.any():
            P[m] += base - r
    return P


def _with_events(P, prof, lib):
    """Rows of `P` with an event pair around every profiled row; returns (table, records for me.KernelProfile)."""
    if not prof:
        return P, []
    out, recs, last = [], [], 0
    # (ascending rows: the issue order of a table with lanes is not the emission order the entries were recorded in)
    for (row, flops, nbytes, meta, pbytes) in sorted(prof, key=lambda t: t[0]):
        ev = _Events(lib)
        out.append(P[last:row])
        e0 = np.zeros((1, STRIDE), dtype=np.int64)
        e0[0, 0], e0[0, 1] = OP_EVENT_RECORD | (int(P[row, 0]) & ~OPCODE_MASK), ev.h[0]      # on the lane of the row it times
        e1 = e0.copy()
        e1[0, 1] = ev.h[1]
        out += [e0, P[row:row + 1], e1]
        last = row + 1
        recs.append((_EvStart(ev), None, flops, nbytes, meta, pbytes))
    out.append(P[last:])
    return np.concatenate(out), recs


# Whether the lanes pay is decided by the device's queues, not by this file: streams are mapped onto a few hardware queues in
# creation order, and a process with other streams around (a communication library's, a framework's) -- or an explicit
# GPU_MAX_HW_QUEUES, or stream priorities -- can end up with two lanes on one queue: measured 29 ms per step instead of 23.5.
# The first backbone passes of a process are therefore timed alternately on their lanes and on the one stream (events on the
# current stream around the table: it ends with the join), and the lanes stay only if they are not slower.
AUTOTUNE = os.environ.get("CG3D_LANES_AUTOTUNE", "1") != "0"


class _LaneTuner:
    skip, samples, decided, verdict = 1, [], False, None

    @classmethod
    def begin(cls, comp, lib):
        """None (no measurement: run as LANES_RUN says) or (lanes_run, start event) for this pass."""
        if cls.decided or not (AUTOTUNE and LANES_RUN and comp.lanes and lib.is_device) or ME.KernelProfile.enabled:
            return None
        if cls.skip > 0:
            cls.skip -= 1
            return None
        if len(cls.samples) >= 4:
            # (per input row: the batches of a training run differ in size; the verdict is reported for the median batch.  Two
            # passes each way and the faster of the two: a pass can only be disturbed towards slower, and the decision should
            # fall inside a handful of warm-up steps)
            for _, _, e1, _ in cls.samples:
                e1.synchronize()             # (a one-off in warm-up: nothing guarantees that a host read followed the sampled passes)
            rows = sorted(n for _, _, _, n in cls.samples)[len(cls.samples) // 2]
            on = min(e0.elapsed_time(e1) / n for m, e0, e1, n in cls.samples if m)
            off = min(e0.elapsed_time(e1) / n for m, e0, e1, n in cls.samples if not m)
            cls.verdict = (on * rows, off * rows)
            cls.decided = True
            if cls.verdict[0] > 1.05 * cls.verdict[1]:
                import sys
                globals()["LANES_RUN"] = False
                sys.stderr.write("cagroup3d_amd.engine: lanes off -- the backbone's forward table took %.2f ms on its lanes against %.2f ms on one "
                                 "stream (queues shared with other streams of this process?)\n" % cls.verdict)
            return None
        mode = len(cls.samples) % 2 == 0
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return mode, e0, max(int(comp.n_in), 1)

    @classmethod
    def end(cls, tok):
        if tok is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            cls.samples.append((tok[0], tok[1], e1, tok[2]))


def _run(lib, P, nrows=None, comp=None, lanes_run=None):
    """comp: the compiled pass the rows come from (its ordering events are bound here; a pass with lanes goes to the lanes entry)."""
    P = np.ascontiguousarray(P)
    n = P.shape[0] if nrows is None else nrows
    if n == 0:
        return
    import ctypes
    fail = ctypes.c_int64(-1)
    if comp is not None and comp.lanes:
        P = _bind_events(P, lib, comp.nevents)
        streams, ns = _lane_streams(lib)
        rc = lib.raw("cg3d_run_program_lanes")(P.ctypes.data, n, ctypes.cast(streams, ctypes.c_void_p), ns if (LANES_RUN if lanes_run is None else lanes_run) else 1,
                                               ctypes.cast(ctypes.pointer(fail), ctypes.c_void_p))
    else:
        rc = lib.raw("cg3d_run_program")(P.ctypes.data, n, lib.stream(), ctypes.cast(ctypes.pointer(fail), ctypes.c_void_p))
    if rc != 0:
        if lib.is_device:
            torch.cuda.synchronize()         # (as in _run_bound: side lanes may still be running rows of this pass)
        raise _lib.CG3DError("cg3d_run_program: row %d (opcode %d) failed with status %d" % (fail.value, (int(P[fail.value, 0]) & OPCODE_MASK) if 0 <= fail.value < n else -1, rc))


# A pass without run-time operands and without timing events is issued from its compiled table as it is: the library applies the
# region bases and the event handles row by row (cg3d_run_program_bound).  The numpy form -- copy, shift, five masked adds, the
# event scan, a concatenation with the zero-fill row -- cost the issuing thread ~0.15 ms per pass, seven passes per step, four of
# them inside the host-bound half.
BOUND = os.environ.get("CG3D_PROGRAM_BOUND", "1") != "0"


def _run_bound(lib, comp, table, bases, zero_ptr, zero_bytes, lo=0, hi=None, lanes_run=None):
    """Rows [lo, hi) of the compiled (unresolved) `table` with `bases` = {region tag: address}; the zero-fill first (0 bytes: none)."""
    import ctypes
    hi = table.shape[0] if hi is None else hi
    b = np.zeros(16, dtype=np.int64)
    for r, a in bases.items():
        b[r >> TAG] = a
    ev_ptr, nev = None, 0
    if comp.nevents:
        arr = _event_pool(lib, comp.nevents)[1]
        ev_ptr, nev = arr.ctypes.data, arr.shape[0]
    streams, ns = _lane_streams(lib)
    use = ns if (comp.lanes and (LANES_RUN if lanes_run is None else lanes_run)) else 1
    fail = ctypes.c_int64(-1)
    rc = lib.raw("cg3d_run_program_bound")(table.ctypes.data + lo * STRIDE * 8, hi - lo, b.ctypes.data, ev_ptr, nev, zero_ptr or None, int(zero_bytes),
                                           ctypes.cast(streams, ctypes.c_void_p), use, ctypes.cast(ctypes.pointer(fail), ctypes.c_void_p))
    if rc != 0:
        f = fail.value
        if lib.is_device:
            torch.cuda.synchronize()         # rows issued before the failing one may still run on side lanes: the caller frees the arena next
        raise _lib.CG3DError("cg3d_run_program_bound: row %d (opcode %d) failed with status %d"
                             % (lo + f, (int(table[lo + f, 0]) & OPCODE_MASK) if 0 <= f < hi - lo else -1, rc))


# ------------------------------------------------------------------------------------------------ parameter gradients
# A backward table writes every parameter gradient of its pass into ONE zero-filled buffer.  Handing 200+ parameters a fresh
# slice each (slice + view + attribute) cost the issuing thread ~1 ms per pass -- inside the stretch of the step where the
# device waits for the host.  The buffer and its per-parameter views are therefore kept per layout and reused while that is
# safe: every parameter's .grad must be None when the pass starts (the training loop's zero_grad(set_to_none=True)), else
# -- gradient accumulation over several backward passes, a second backward through a retained graph -- the pass gets a
# fresh buffer and adds, as before.  The contract that follows from the reuse: a .grad handed out by one pass is a view of
# the pooled buffer, and the NEXT pass zero-fills and rewrites that buffer -- whoever keeps a gradient beyond
# zero_grad(set_to_none=True) (gradient logging, EMA of gradients) must clone it, or run with CG3D_PG_REUSE=0.
PG_REUSE = os.environ.get("CG3D_PG_REUSE", "1") != "0"
_PG_POOL = collections.OrderedDict()


def _pg_views(comp, device):
    """(buffer, views aligned with comp.params, fresh): a zero-filled buffer for the pass's parameter gradients."""
    n = max(comp.size[R_PG] // 4, 1)
    params = comp.params
    if PG_REUSE and params and all(prm.grad is None for prm, _ in params):
        key = (n, str(device), tuple((id(prm), off) for prm, off in params))
        hit = _PG_POOL.get(key)
        if hit is not None and all(a is b[0] for a, b in zip(hit[2], params)):
            hit[0].zero_()
            return hit[0], hit[1], False
        pg = torch.zeros(n, dtype=torch.float32, device=device)
        views = [pg[off // 4: off // 4 + prm.numel()].view_as(prm) for prm, off in params]
        _PG_POOL[key] = (pg, views, [prm for prm, _ in params])
        while len(_PG_POOL) > 6:
            _PG_POOL.popitem(last=False)
        return pg, views, False
    pg = torch.zeros(n, dtype=torch.float32, device=device)
    return pg, [pg[off // 4: off // 4 + prm.numel()].view_as(prm) for prm, off in params], True


def _arena_alive(ctx):
    """A program node's backward runs ONCE: it releases the arena its rows point into (activations, statistics, scratch: the
    largest allocation of the step).  These nodes save nothing through save_for_backward, so autograd itself would not object
    to a second backward (retain_graph=True, or autograd.grad followed by backward) -- which would re-run the whole table on
    freed or reused memory and write parameter gradients from it."""
    if ctx.arena is None:
        raise RuntimeError("engine program: backward already ran for this node and its arena has been released "
                           "(a launch program's backward pass can run once; recompute the forward for another one)")


class BackboneFunction(torch.autograd.Function):
    """The whole backbone as ONE autograd node: forward = the forward table, backward = the backward table; the parameters'
    gradients are written to one fresh zero-filled buffer and handed to the parameters here (`p.grad = view`, added to an
    existing gradient), so no AccumulateGrad node runs per parameter.

    hooks: None or {mark name: (callable, ids of the parameters whose gradients must be visible to it)} -- the callable runs
    between the two parts of the backward table cut at that mark (the data-parallel exchange sends its mid bucket there)."""

    @staticmethod
    def forward(ctx, feats, anchor, comp, hooks, io):
        lib = comp.lib
        feats = feats.contiguous()
        assert feats.shape == (comp.n_in, comp.c_in) and feats.dtype == torch.float32
        sz = comp.size
        zf, zb, act = sz[R_ZF], sz[R_ZB], sz[R_ACT]
        arena = torch.empty(zf + zb + act + ALIGN, dtype=torch.uint8, device=feats.device)
        base = (arena.data_ptr() + ALIGN - 1) & ~(ALIGN - 1)
        bases = {R_ZF: base, R_ZB: base + zf, R_ACT: base + zf + zb, R_IN: feats.data_ptr()}
        keep = []
        prof = bool(ME.KernelProfile.enabled and lib.is_device)
        tok = _LaneTuner.begin(comp, lib)
        if BOUND and not comp.late_f and not prof:
            _run_bound(lib, comp, comp.fwd, bases, base, zf, lanes_run=tok[0] if tok else None)
        else:
            P = _resolve(comp.fwd, bases)
            for r, c, fn in comp.late_f:
                t = fn()
                keep.append(t)
                P[r, c] = t.data_ptr()
            head = np.zeros((1, STRIDE), dtype=np.int64)
            head[0, :4] = (OP_MEMSET, base, 0, zf)
            if prof:
                P, recs = _with_events(P, comp.fprof, lib)
                ME.KernelProfile.records.extend(recs)
            _run(lib, np.concatenate([head, P]), comp=comp, lanes_run=tok[0] if tok else None)
        _LaneTuner.end(tok)
        ctx.comp, ctx.arena, ctx.bases, ctx.feats, ctx.keep, ctx.hooks, ctx.prof = comp, arena, bases, feats, keep, hooks, prof
        p, n, c, p16 = comp.out
        shift = bases[R_ACT] - arena.data_ptr()
        off = p - R_ACT + shift
        y = arena[off:off + n * c * 4].view(torch.float32).view(n, c)
        if p16 and io is not None:
            o16 = p16 - R_ACT + shift
            io["y16"] = arena[o16:o16 + n * c * 2].view(torch.int16).view(n, c)
        return y

    @staticmethod
    def backward(ctx, dy):
        _arena_alive(ctx)
        comp, lib, hooks = ctx.comp, ctx.comp.lib, ctx.hooks or {}
        dy = dy.contiguous()
        bases = dict(ctx.bases)
        pg, views, _ = _pg_views(comp, dy.device)
        bases[R_PG], bases[R_DOUT] = pg.data_ptr(), dy.data_ptr()
        fast = BOUND and not comp.late_b and not ctx.prof
        keep = []
        cuts = sorted((row, name) for name, row in comp.marks.items() if name in hooks)
        parts, last = [], 0
        if fast:
            for row, name in cuts:
                parts.append(((last, row), None, name))
                last = row
            parts.append(((last, comp.bwd.shape[0]), None, None))
        else:
            P = _resolve(comp.bwd, bases)
            for r, c, fn in comp.late_b:
                t = fn()
                keep.append(t)
                P[r, c] = t.data_ptr()
            head = np.zeros((1, STRIDE), dtype=np.int64)
            head[0, :4] = (OP_MEMSET, bases[R_ZB], 0, comp.size[R_ZB])
            for row, name in cuts:
                parts.append((P[last:row], comp_prof_slice(comp.bprof, last, row), name))
                last = row
            parts.append((P[last:], comp_prof_slice(comp.bprof, last, P.shape[0]), None))
        given = set()

        def give(ids):
            # hand the gradients written so far to their parameters (ids None: all that are left)
            with torch.no_grad():
                for (prm, _), g in zip(comp.params, views):
                    k = id(prm)
                    if k in given or (ids is not None and k not in ids):
                        continue
                    if prm.requires_grad:          # (a frozen layer's gradient is computed by the table but not handed out)
                        prm.grad = g if prm.grad is None else prm.grad + g
                    given.add(k)
        first = True
        for rows, prof, name in parts:
            if fast:
                _run_bound(lib, comp, comp.bwd, bases, bases[R_ZB] if first else 0, comp.size[R_ZB] if first else 0, rows[0], rows[1])
            else:
                if ctx.prof:
                    rows, recs = _with_events(rows, prof, lib)
                    ME.KernelProfile.records.extend(recs)
                _run(lib, np.concatenate([head, rows]) if first else rows, comp=comp)
            first = False
            if name is not None:
                fn, ids = hooks[name]
                give(ids)
                fn()
        give(None)
        ctx.arena = ctx.keep = ctx.feats = None
        return None, None, None, None, None


# ------------------------------------------------------------------------------------------------ the class branches
CLASS_PROGRAM = os.environ.get("CG3D_CLASS_PROGRAM", "1") != "0"
CLASS_STATS = {"program_passes": 0, "not_ready": 0}


def class_branches_applicable(head):
    """Training step of the batched dense head in the bench precision on the device library (CG3D_ENGINE_ANY: tests)."""
    if not (ENABLED and CLASS_PROGRAM and head.training and torch.is_grad_enabled()) or ME.coords_only():
        return False
    if not (ME._prec() in (1, 3) and ME.BF16_ROWS and ME.GROUPED_BN_STACK):
        return False
    return _lib.get().device_kernels or os.environ.get("CG3D_ENGINE_ANY") == "1"


def compile_class_branches(head, nf, nc, c, km9, km5, km_up, ident, fine_bounds, coarse_bounds, device):
    """Program pair for the feature side of the class branches (reference cagroup_head.py:227-282, all classes at once as in
    CAGroup3DHead._class_branches_batched): out conv 9^3 + BN + ELU on the fine map, expand conv 5^3 + BN + ELU on the coarse
    map, generative transposed conv + BN + ELU back onto the fine voxels, concatenation, fuse conv + BN + ELU.  The maps are
    the ones the caller built; compiling adds their pair lists / tile plans (with their host reads) as the per-layer path
    does.  Inputs: R_IN = features on the fine map [nf, c], R_IN2 = features on the coarse map [nc, c]."""
    lib = _lib.get()
    b = Builder(lib, device, ME._WeightPlan.gen)
    xf, xc = T(R_IN, nf, c, need=True), T(R_IN2, nc, c, need=True)
    elu = ME.ACT_ELU
    L = head.__dict__.get("_cb_layers")
    m0 = head.cls_individual_out[0]
    if L is None or L[0][0][0] is not m0[0].kernel or L[0][1][0] is not m0[1].bn:      # (a replaced module, e.g. convert_sync_batchnorm)
        # (216 module-tree look-ups per step otherwise; the lists hold the modules' own Parameter / BatchNorm1d objects)
        L = head.__dict__["_cb_layers"] = tuple(
            ([m[0].kernel for m in mods], [(m[1][0] if up else m[1]).bn for m in mods])
            for mods, up in ((head.cls_individual_out, False), (head.cls_individual_expand_out, False),
                             (head.cls_individual_up, True), (head.cls_individual_fuse, False)))
    ME.pairs_many([(km9, fine_bounds), (km5, coarse_bounds), (km_up, fine_bounds), (ident, fine_bounds)])    # one host read for the four
    # two independent branches until the concatenation: the 9^3 convolution on the fine map streams its 18 classes' weights
    # (HBM-bound, lane 0); the 5^3 convolution on the coarse map and the transposed convolution back are pair-kernel launches
    # bound by their gather latency (lane 1)
    b.set_lane(1)
    e = b.gbn_act(b.gconv(xc, L[1][0], km5, coarse_bounds, True), L[1][1], coarse_bounds, elu)
    u = b.gbn_act(b.gconv(e, L[2][0], km_up, fine_bounds, False), L[2][1], fine_bounds, elu)
    b.set_lane(0)
    a = b.gbn_act(b.gconv(xf, L[0][0], km9, fine_bounds, True), L[0][1], fine_bounds, elu)
    f = b.gbn_act(b.gconv(b.cat([u, a]), L[3][0], ident, fine_bounds, False), L[3][1], fine_bounds, elu)
    b.emit_backward(f)
    gin = (b.grad(xf), b.grad(xc))              # (may add the rows that sum several contributions: before the tables are cut)
    comp = Compiled(b, f, None, nf, c)
    comp.n_in2, comp.gin = nc, gin
    # the rows address the maps' tables, pair lists, segment tables and tile plans (cached inside the KernelMap objects): the
    # program owns them until its backward pass has run -- the caller's sparse tensors and managers die with its frame
    comp.maps = (km9, km5, km_up, ident)
    return comp


def _arena_view(arena, bases, addr, n, c, dtype=torch.float32, width=4):
    tag = (addr >> TAG) << TAG
    off = bases[tag] + (addr - tag) - arena.data_ptr()
    return arena[off:off + n * c * width].view(dtype).view(n, c)


class ClassBranchFunction(torch.autograd.Function):
    """The feature side of all class branches as ONE autograd node (two inputs, one output); the parameters' gradients are
    slices of one zero-filled buffer handed to the parameters in backward, as in BackboneFunction."""

    @staticmethod
    def forward(ctx, xf, xc, comp):
        lib = comp.lib
        xf, xc = xf.contiguous(), xc.contiguous()
        assert xf.shape == (comp.n_in, comp.c_in) and xc.shape == (comp.n_in2, comp.c_in) and xf.dtype == xc.dtype == torch.float32
        sz = comp.size
        zf, zb, act = sz[R_ZF], sz[R_ZB], sz[R_ACT]
        arena = torch.empty(zf + zb + act + ALIGN, dtype=torch.uint8, device=xf.device)
        base = (arena.data_ptr() + ALIGN - 1) & ~(ALIGN - 1)
        bases = {R_ZF: base, R_ZB: base + zf, R_ACT: base + zf + zb, R_IN: xf.data_ptr(), R_IN2: xc.data_ptr()}
        prof = bool(ME.KernelProfile.enabled and lib.is_device)
        if BOUND and not comp.late_f and not prof:
            _run_bound(lib, comp, comp.fwd, bases, base, zf)
        else:
            P = _resolve(comp.fwd, bases)
            head = np.zeros((1, STRIDE), dtype=np.int64)
            head[0, :4] = (OP_MEMSET, base, 0, zf)
            if prof:
                P, recs = _with_events(P, comp.fprof, lib)
                ME.KernelProfile.records.extend(recs)
            _run(lib, np.concatenate([head, P]), comp=comp)
        ctx.comp, ctx.arena, ctx.bases, ctx.inputs, ctx.prof = comp, arena, bases, (xf, xc), prof
        p, n, c, p16 = comp.out
        y = _arena_view(arena, bases, p, n, c)
        if p16:
            ME._ROWS16[y.data_ptr()] = (y, _arena_view(arena, bases, p16, n, c, torch.int16, 2))
        return y

    @staticmethod
    def backward(ctx, dy):
        _arena_alive(ctx)
        comp, lib = ctx.comp, ctx.comp.lib
        dy = dy.contiguous()
        bases = dict(ctx.bases)
        pg, views, _ = _pg_views(comp, dy.device)
        bases[R_PG], bases[R_DOUT] = pg.data_ptr(), dy.data_ptr()
        if BOUND and not comp.late_b and not ctx.prof:
            _run_bound(lib, comp, comp.bwd, bases, bases[R_ZB], comp.size[R_ZB])
        else:
            P = _resolve(comp.bwd, bases)
            head = np.zeros((1, STRIDE), dtype=np.int64)
            head[0, :4] = (OP_MEMSET, bases[R_ZB], 0, comp.size[R_ZB])
            if ctx.prof:
                P, recs = _with_events(P, comp.bprof, lib)
                ME.KernelProfile.records.extend(recs)
            _run(lib, np.concatenate([head, P]), comp=comp)
        with torch.no_grad():
            for (prm, _), g in zip(comp.params, views):
                if prm.requires_grad:
                    prm.grad = g if prm.grad is None else prm.grad + g
        grads = []
        for gi, (n, c) in zip(comp.gin, ((comp.n_in, comp.c_in), (comp.n_in2, comp.c_in))):
            grads.append(_arena_view(ctx.arena, bases, gi[0], n, c) if gi is not None else None)
        ctx.arena = ctx.inputs = None
        return grads[0], grads[1], None


def run_class_branches(head, xf, xc, km9, km5, km_up, ident, fine_bounds, coarse_bounds):
    """Features [nf, c] after the fuse BatchNorm + ELU of all class branches.  Raises NotReady when a layer has no program form."""
    try:
        comp = compile_class_branches(head, xf.shape[0], xc.shape[0], xf.shape[1], km9, km5, km_up, ident, fine_bounds,
                                      coarse_bounds, xf.device)
    except NotReady:
        CLASS_STATS["not_ready"] += 1
        raise
    try:
        comp.check_weights_live()
    except NotReady:
        CLASS_STATS["not_ready"] += 1
        raise
    CLASS_STATS["program_passes"] += 1
    return ClassBranchFunction.apply(xf, xc, comp)


# ------------------------------------------------------------------------------------------------ the head's first layers
# Round 4: measured at 153.6 against 153.7 scenes/s on one queue (the seven layers' backward nodes cost the issuing thread
# 0.55 ms, the program's node 0.35 ms, and compiling it inline adds to the start of the step) -- off.  Round 5, with the two
# branches on two lanes (three 1x1x1 layers beside one 3^3 convolution, both inside the device-bound half of the step): per-step
# medians 23.6 / 23.3 / 23.3 ms against 24.1 / 23.8 / 23.6 over 100 pinned steps, alternating -- on (CG3D_HEAD_PROGRAM=0: off).
HEAD_PROGRAM = os.environ.get("CG3D_HEAD_PROGRAM", "1") == "1"
HEAD_STATS = {"program_passes": 0, "not_ready": 0}


def head_pre_applicable(head):
    """Training step in the bench precision on the device library."""
    if not (ENABLED and HEAD_PROGRAM and head.training and torch.is_grad_enabled()) or ME.coords_only():
        return False
    return _lib.get().device_kernels and ME._prec() in (1, 3) and ME.BF16_ROWS


def compile_head_pre(head, sp, has16):
    """Program pair for the two branches the dense head puts on the backbone output before anything data dependent
    (reference cagroup_head.py:150-162, 196-199): the vote-offset block (1x1x1 conv, BN, ELU, twice, then the 1x1x1 conv to
    3 n_vote offsets) and the feature-offset block (3^3 conv, BN, ELU).  One input (R_IN: the backbone output rows; R_IN2:
    their bf16 copy when the backbone program left one), two outputs."""
    lib = _lib.get()
    b = Builder(lib, sp.F.device, ME._WeightPlan.gen)
    n, c = sp.F.shape
    xt = T(R_IN, n, c, need=True)
    if has16:
        xt.p16 = R_IN2
    em = Emitter(b, sp.coordinate_manager)
    x = _X(xt, sp.coordinate_map_key)
    b.set_lane(1)                       # two branches on the same rows: three 1x1x1 layers beside one 3^3 convolution
    off = em.seq(head.offset_block, x)
    b.set_lane(0)
    fo = em.seq(head.feature_offset, x)
    b.emit_backward([off.t, fo.t])
    gin = b.grad(xt)
    comp = Compiled(b, off.t, None, n, c, sp.coordinate_manager)
    comp.out2 = (fo.t.p, fo.t.n, fo.t.c, fo.t.p16)
    comp.gin = gin
    return comp


class HeadPreFunction(torch.autograd.Function):
    """The head's two coordinate-independent branches as ONE autograd node (one input, two outputs)."""

    @staticmethod
    def forward(ctx, x, x16, comp):
        lib = comp.lib
        x = x.contiguous()
        assert x.shape == (comp.n_in, comp.c_in) and x.dtype == torch.float32
        sz = comp.size
        zf, zb, act = sz[R_ZF], sz[R_ZB], sz[R_ACT]
        arena = torch.empty(zf + zb + act + ALIGN, dtype=torch.uint8, device=x.device)
        base = (arena.data_ptr() + ALIGN - 1) & ~(ALIGN - 1)
        bases = {R_ZF: base, R_ZB: base + zf, R_ACT: base + zf + zb, R_IN: x.data_ptr(),
                 R_IN2: x16.data_ptr() if x16 is not None else 0}
        prof = bool(ME.KernelProfile.enabled and lib.is_device)
        if BOUND and not comp.late_f and not prof:
            _run_bound(lib, comp, comp.fwd, bases, base, zf)
        else:
            P = _resolve(comp.fwd, bases)
            head = np.zeros((1, STRIDE), dtype=np.int64)
            head[0, :4] = (OP_MEMSET, base, 0, zf)
            if prof:
                P, recs = _with_events(P, comp.fprof, lib)
                ME.KernelProfile.records.extend(recs)
            _run(lib, np.concatenate([head, P]), comp=comp)
        ctx.comp, ctx.arena, ctx.bases, ctx.inputs, ctx.prof = comp, arena, bases, (x, x16), prof
        outs = []
        for p, n, c, p16 in (comp.out, comp.out2):
            y = _arena_view(arena, bases, p, n, c)
            if p16:
                ME._ROWS16[y.data_ptr()] = (y, _arena_view(arena, bases, p16, n, c, torch.int16, 2))
            outs.append(y)
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, d0, d1):
        _arena_alive(ctx)
        comp, lib = ctx.comp, ctx.comp.lib
        bases = dict(ctx.bases)
        shapes = ((comp.out[1], comp.out[2]), (comp.out2[1], comp.out2[2]))
        d0 = d0.contiguous() if d0 is not None else torch.zeros(shapes[0], dtype=torch.float32, device=ctx.arena.device)
        d1 = d1.contiguous() if d1 is not None else torch.zeros(shapes[1], dtype=torch.float32, device=ctx.arena.device)
        pg, views, _ = _pg_views(comp, d0.device)
        bases[R_PG], bases[R_DOUT], bases[R_DOUT2] = pg.data_ptr(), d0.data_ptr(), d1.data_ptr()
        keep = []
        if BOUND and not comp.late_b and not ctx.prof:
            _run_bound(lib, comp, comp.bwd, bases, bases[R_ZB], comp.size[R_ZB])
        else:
            P = _resolve(comp.bwd, bases)
            for r, c, fn in comp.late_b:
                t = fn()
                keep.append(t)
                P[r, c] = t.data_ptr()
            head = np.zeros((1, STRIDE), dtype=np.int64)
            head[0, :4] = (OP_MEMSET, bases[R_ZB], 0, comp.size[R_ZB])
            if ctx.prof:
                P, recs = _with_events(P, comp.bprof, lib)
                ME.KernelProfile.records.extend(recs)
            _run(lib, np.concatenate([head, P]), comp=comp)
        with torch.no_grad():
            for (prm, _), g in zip(comp.params, views):
                if prm.requires_grad:
                    prm.grad = g if prm.grad is None else prm.grad + g
        dx = _arena_view(ctx.arena, bases, comp.gin[0], comp.n_in, comp.c_in) if comp.gin is not None else None
        ctx.arena = ctx.inputs = None
        return dx, None, None


def run_head_pre(head, sp):
    """(vote offsets [n, 3 n_vote], offset features [n, c n_vote]) of the dense head on the backbone output `sp`.
    Raises NotReady when a layer has no program form (first steps: weights not in the step's arena yet)."""
    x = sp.F
    hit = ME._ROWS16.get(x.data_ptr())
    x16 = hit[1] if (hit is not None and hit[0].shape == x.shape and hit[0].data_ptr() == x.data_ptr() and ME._prec() == 1) else None
    try:
        comp = compile_head_pre(head, sp, x16 is not None)
    except NotReady:
        HEAD_STATS["not_ready"] += 1
        raise
    try:
        comp.check_weights_live()
    except NotReady:
        HEAD_STATS["not_ready"] += 1
        raise
    HEAD_STATS["program_passes"] += 1
    return HeadPreFunction.apply(x, x16, comp)


def comp_prof_slice(prof, lo, hi):
    return [(r - lo, f, b, m, pb) for (r, f, b, m, pb) in prof if lo <= r < hi]


def run_backbone(net, sp, comp=None, hooks=None):
    """Forward pass of `net` on `sp` through its program; returns the output SparseTensor.  `comp`: the pair compiled ahead
    (CAGroup3D.prefetch_coordinates), else compiled here.  Raises NotReady when the program cannot be built yet."""
    try:
        if comp is None or not comp.usable() or (hooks and any(k not in comp.marks for k in hooks)):
            comp = compile_backbone(net, sp, mid_mark=bool(hooks and "mid" in hooks))
            STATS["compiled_inline"] += 1
        else:
            STATS["compiled_ahead"] += 1
        comp.check_weights_live()
    except NotReady:
        STATS["not_ready"] += 1
        raise
    STATS["program_passes"] += 1
    anchor = net.conv1[0].kernel
    io = {}
    y = BackboneFunction.apply(sp.F, anchor, comp, hooks, io)
    if io.get("y16") is not None:
        ME._ROWS16[y.data_ptr()] = (y, io["y16"])       # the head's first convolutions gather from the bf16 copy
    return ME.SparseTensor(features=y, coordinate_map_key=comp.out_key, coordinate_manager=sp.coordinate_manager)


def applicable(net, compiling=False):
    """A training step in the bench precision on the device library.  CG3D_ENGINE_ANY=1 (tests) also admits the fp32 parity
    mode -- on the device library and on the CPU oracle -- where every product runs through the generic pair kernels.
    compiling: asked by the dry run that compiles the program ahead (it runs under no_grad)."""
    if not ENABLED or not net.training or ME.coords_only() or not (compiling or torch.is_grad_enabled()):
        return False
    lib = _lib.get()
    if lib.device_kernels and ME.PRECISION in (1, 3) and ME.BF16_ROWS:
        return True
    return os.environ.get("CG3D_ENGINE_ANY") == "1" and ME.PRECISION == 0
