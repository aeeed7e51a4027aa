"""Host-side sparse-voxel engine: the slice of the MinkowskiEngine v0.5.4 surface that the
reference's four hot-path modules call, re-implemented over the C-ABI in
include/cagroup3d_hip.h.  Same names and argument meaning as `import MinkowskiEngine as ME`, so
the model files read like the reference's (pcdet/models/backbones_3d/biresnet.py,
dense_heads/cagroup_head.py, roi_heads/cagroup_roi_head.py, detectors/cagroup3d.py).

Design (MI355X-first, not ME's):
  * a coordinate map = int32 [N,4] rows + an open-addressed hash table in HBM; built once per
    tensor stride per batch and cached in the CoordinateManager;
  * a kernel map = dense k-major table nbr[K, N_out] (-1 = absent) and, for the backward pass,
    its transpose nbrT[K, N_in]; cached per (in map, out map, kernel, dilation, kind);
  * convolution = output-stationary implicit GEMM on MFMA (cg3d_spconv_fwd); the data gradient is
    the same kernel on the transposed map; the weight gradient is cg3d_spconv_wgrad;
  * everything is asynchronous on torch's current stream; the only host syncs are the
    row-count read-backs when a NEW coordinate map is created.

Conventions fixed by this engine (ME's own are unverifiable here -- "parity unpinned"):
  kernel offset index k enumerates (ix, iy, iz) with iz fastest; odd kernels are centred
  (-(k//2) .. k//2), even kernels start at 0; weights are [K, Cin, Cout] ([Cin, Cout] for K==1).
"""
import itertools
import math
import os

import numpy as np
import ctypes
from ctypes import c_float, c_int32, c_int64, c_void_p
from enum import Enum

import torch
import torch.nn as nn

from . import _lib
from ._lib import ptr


class _PinnedStage:
    """Pinned staging memory for the many small host tables of a step (segment tables, tile tables, row counts).
    `Tensor.pin_memory()` goes through the caching host allocator, and every table has another size: misses end in
    hipHostMalloc, which was caught stalling the launching thread for 50-100 ms behind a busy device (tools/outliers.py).
    Two fixed halves per (device, stream) instead: a half is reused only after the event recorded when it was left --
    half a ring (tens of steps) earlier -- has completed."""
    HALF = 16 << 20
    _rings = {}

    def __init__(self):
        self.buf = torch.empty(2 * self.HALF, dtype=torch.uint8).pin_memory()
        self.base = self.buf.data_ptr()
        self.half, self.off = 0, 0
        self.left = [None, None]          # event recorded when a half was left

    @classmethod
    def get(cls, stream):
        key = (stream.device_index, stream.cuda_stream)
        ring = cls._rings.get(key)
        if ring is None:
            ring = cls._rings[key] = cls()
        return ring

    def take(self, nbytes, stream_of):
        """Byte offset of a fresh 64-byte-aligned slot; `stream_of()` is asked for the stream only when a half is left."""
        nbytes = (nbytes + 63) & ~63
        if self.off + nbytes > self.HALF:
            ev = torch.cuda.Event()
            ev.record(stream_of())
            self.left[self.half] = ev
            self.half ^= 1
            self.off = 0
            if self.left[self.half] is not None:
                self.left[self.half].synchronize()      # recorded half a ring ago: complete long since
        o = self.half * self.HALF + self.off
        self.off += nbytes
        return o


_NP_OF = {torch.int32: np.int32, torch.int64: np.int64, torch.float32: np.float32, torch.uint8: np.uint8, torch.int16: np.int16,
          torch.float64: np.float64}


def h2d(data, dtype, device):
    """Small host table -> device WITHOUT stalling the stream: pinned staging + an asynchronous copy (a pageable-memory
    copy blocks the host until every kernel queued before it has finished).  numpy into the pinned ring, ONE framework
    op (the output allocation) and one C call (cg3d_h2d_async): the chain of ~10 tensor ops this used to be was
    ~550 of the step's ~5 300 framework ops (74 tables per step)."""
    dev = device if isinstance(device, torch.device) else torch.device(device)
    if dev.type != "cuda":
        t = torch.as_tensor(data, dtype=dtype) if not torch.is_tensor(data) else data.to(dtype)
        return t.to(dev)
    arr = np.ascontiguousarray(data.numpy() if torch.is_tensor(data) else data, dtype=_NP_OF[dtype])
    nbytes = arr.nbytes
    if nbytes == 0 or nbytes > _PinnedStage.HALF:
        return torch.from_numpy(arr).pin_memory().to(dev, non_blocking=True)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    raw = torch._C._cuda_getCurrentRawStream(idx)
    ring = _PinnedStage._rings.get((idx, raw))
    if ring is None:
        ring = _PinnedStage._rings[(idx, raw)] = _PinnedStage()
    o = ring.take(nbytes, lambda: torch.cuda.current_stream(dev))
    ctypes.memmove(ring.base + o, arr.ctypes.data, nbytes)
    out = torch.empty(arr.shape, dtype=dtype, device=dev)
    _lib.get().call("cg3d_h2d_async", c_void_p(out.data_ptr()), c_void_p(ring.base + o), c_int64(nbytes), c_void_p(raw))
    return out


# ----------------------------------------------------------------------------- coordinate maps
class CoordinateMapKey:
    def __init__(self, tensor_stride, uid):
        self.tensor_stride = int(tensor_stride)
        self.uid = uid

    def get_key(self):
        ts = self.tensor_stride
        return ((ts, ts, ts), str(self.uid))

    def get_tensor_stride(self):
        ts = self.tensor_stride
        return (ts, ts, ts)

    def __hash__(self):
        return hash((self.tensor_stride, self.uid))

    def __eq__(self, o):
        return isinstance(o, CoordinateMapKey) and (self.tensor_stride, self.uid) == (o.tensor_stride, o.uid)

    def __repr__(self):
        return "CoordinateMapKey(ts=%d, id=%s)" % (self.tensor_stride, self.uid)


class _CoordMap:
    __slots__ = ("coords", "keys", "vals", "cap", "n", "tensor_stride", "_perms", "_perms_sorted")

    def __init__(self, coords, keys, vals, cap, n, tensor_stride):
        self.coords, self.keys, self.vals, self.cap, self.n = coords, keys, vals, cap, n
        self.tensor_stride = tensor_stride
        self._perms = None
        self._perms_sorted = False


class KernelMap:
    """Kernel map between an input and an output coordinate map.

    nbr[K, n_out] (k-major, -1 = absent) is the dense form every consumer is derived from; the
    convolution kernels run on its compaction into pair lists ordered by (k, out row):
    pair_in / pair_out int32 [P], pair_off (host) int64 [K+1].  The same lists serve the forward
    (in -> out), the data gradient (out -> in, lists swapped) and the weight gradient."""

    def __init__(self, nbr, K, n_in, n_out, make_T):
        self.nbr, self.K, self.n_in, self.n_out = nbr, K, n_in, n_out
        self._nbrT, self._make_T = None, make_T
        self._pairs = None
        self._segs = {}
        self.same_map = False
        self.symmetric = False

    @property
    def nbrT(self):
        if self._nbrT is None:
            self._nbrT = self._make_T()
        return self._nbrT

    @staticmethod
    def identity(n, device):
        """The 1-offset map o -> o (a 1x1x1 convolution; used for grouped per-class linear layers)."""
        nbr = torch.arange(n, dtype=torch.int32, device=device).view(1, n)
        return KernelMap(nbr, 1, n, n, lambda: nbr)

    def _pairs_count(self, row_bounds):
        """First half of `pairs`: the counting launch; returns what `_pairs_finish` needs (the offsets are still on the device)."""
        lib = _lib.get()
        dev = self.nbr.device
        total = self.K * self.n_out
        G = 1 if row_bounds is None else len(row_bounds) - 1
        # ungrouped maps: the same launch also reports where every block of WGRAD_BLOCK_ROWS output rows starts in
        # each offset's list (the blocks are "groups" to the counting kernel) -- `wgrad_segments` cuts along them
        blocks = None
        if WGRAD_ROW_BLOCKS and row_bounds is None and lib.is_device and self.n_out >= WGRAD_BLOCK_MIN_ROWS:
            nb = -(-self.n_out // WGRAD_BLOCK_ROWS)
            blocks = tuple(range(0, nb * WGRAD_BLOCK_ROWS, WGRAD_BLOCK_ROWS)) + (self.n_out,)
            G = nb
        rb = h2d(blocks if blocks is not None else row_bounds, torch.int32, dev) if (blocks is not None or row_bounds is not None) else None
        ws = torch.empty(max(int(lib.raw("cg3d_pairs_ws_bytes")(total)) // 4, 1), dtype=torch.int32, device=dev)
        off = torch.empty(self.K * G + 1, dtype=torch.int32, device=dev)      # (zero-filled by the call)
        lib.call("cg3d_pairs_count", ptr(self.nbr), c_int32(self.K), c_int64(self.n_out), ptr(rb), c_int32(G), ptr(ws),
                 ptr(off), lib.stream())
        return (ws, off, G, blocks, rb)

    def _pairs_finish(self, row_bounds, pending, off_h):
        lib = _lib.get()
        ws, _, G, blocks, _ = pending
        any_hit = next(iter(self._pairs.values()), None)
        off_h = off_h.astype(np.int64)
        if blocks is not None:
            self._blocks = (G, off_h)
            off_h = np.concatenate([off_h[0:self.K * G:G], off_h[-1:]])
        P = int(off_h[-1])
        if any_hit is not None:
            pin, pout = any_hit[0], any_hit[1]       # the lists do not depend on the grouping
        else:
            dev = self.nbr.device
            pin = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
            pout = torch.empty(max(P, 1), dtype=torch.int32, device=dev)
            lib.call("cg3d_pairs_fill", ptr(self.nbr), c_int32(self.K), c_int64(self.n_out), ptr(ws), ptr(pin),
                     ptr(pout), lib.stream())
        hit = (pin, pout, off_h, P)
        self._pairs[row_bounds] = hit
        return hit

    def pairs(self, row_bounds=None):
        """(pair_in, pair_out, pair_off host int64 [K*G+1], P).  row_bounds: host tuple of G+1 output-row
        boundaries (contiguous groups with their own weights) or None for a single group."""
        if self._pairs is None:
            self._pairs = {}
        hit = self._pairs.get(row_bounds)
        if hit is None:
            pending = self._pairs_count(row_bounds)
            off_h = pending[1].cpu().numpy()          # host sync, once per kernel map (and grouping)
            hit = self._pairs_finish(row_bounds, pending, off_h)
        return hit

    def wgrad_segments(self, maxlen, row_bounds=None):
        """Segment table of the weight-gradient launch.  Ungrouped maps of >= WGRAD_BLOCK_MIN_ROWS rows are cut along
        blocks of WGRAD_BLOCK_ROWS OUTPUT ROWS instead of equal pair counts: segment (block b, offset k) = the pairs of
        offset k whose output row lies in block b, launched in the order  k-major inside groups of 8 blocks, so that
        workgroup i (XCD i % 8) works on block 8*(i / 8K) + i % 8: every XCD walks its own blocks one after the other, all
        27 offsets of a block at the same time, and the block's rows (2048 x 512 B of bf16 X and dY rows) are fetched
        into that XCD's L2 once and found there by the other 26 offsets -- with equal-count segments the offsets move
        through the rows at different speeds (sparser offsets cover more rows per pair) and the L2 hit rate of the
        gathers was 33 % (10 % before the XCD-aware order), 3.4 x the tensors' bytes on the memory side.  Entries past
        the last block are empty placeholders (count 0) that keep the alignment."""
        self.pairs(row_bounds)
        blk = getattr(self, "_blocks", None)
        if row_bounds is not None or blk is None or not WGRAD_ROW_BLOCKS:
            return self.segments(maxlen, row_bounds)
        ck = ("wrows",)
        seg = self._segs.get(ck)
        if seg is None:
            nb, ob = blk
            K = self.K
            i = np.arange(-(-nb // N_XCD) * N_XCD * K, dtype=np.int64)
            b = N_XCD * (i // (N_XCD * K)) + i % N_XCD
            k = (i // N_XCD) % K
            slot = k * nb + np.minimum(b, nb - 1)
            start = ob[slot]
            cnt = np.where(b < nb, ob[slot + 1] - start, 0)
            tab = np.stack([k, start, cnt], 1).astype(np.int32)
            seg = (h2d(torch.from_numpy(tab), torch.int32, self.nbr.device), int(tab.shape[0]))
            self._segs[ck] = seg
        return seg

    def tile_plan(self, transposed, row_bounds=None):
        """TilePlan of the map (forward) or of its transpose (data gradient); cached.  With `row_bounds` the tiles are
        cut group by group (`tiles(row_bounds)`: no tile straddles two groups' weights)."""
        ck = ("plan", bool(transposed) and not self.symmetric, row_bounds)        # symmetric: one plan, weights reversed
        transposed = ck[1]
        pl = self._segs.get(ck)
        if pl is None:
            _, _, _, P = self.pairs(row_bounds)
            nbr = self.nbrT if transposed else self.nbr
            # sparse maps (the transposed map of a strided convolution, the map of a transposed one: a row has neighbours
            # only at the offsets of its parity class, occupancy 0.10-0.13): tiles cut from rows grouped by their set of
            # live offsets multiply 2-3 x the rows they need instead of 7-8 x (cg3d_tile_row_order)
            sort_rows = 0
            if TILE_SORT_ROWS and row_bounds is None and self.K <= 32:
                if not self.symmetric and P < TILE_SORT_MAX_OCCUPANCY * self.K * max(nbr.shape[1], 1):
                    sort_rows = 1024
                elif TILE_SORT_IN_TILE:
                    # every other map: the tiles keep their rows (and the distinct input rows they stage), but inside a tile
                    # rows with the same live offsets share 32-row blocks, whose dead (offset, block) pairs the kernel skips
                    sort_rows = 128
            # (kernels of more than 64 offsets, the 5^3 / 9^3 class convolutions: TILE_UCAP_BIGK)
            pl = build_tile_plan(nbr.contiguous(), P, None if row_bounds is None else self.tiles(row_bounds), sort_rows=sort_rows,
                                 ucap=TILE_UCAP_BIGK if self.K > 64 else None)
            self._segs[ck] = pl
        return pl

    def tiles(self, row_bounds, rows=128):
        """int32 [ntile,3] (group, first row, row count <= rows) covering the output rows group by group; cached."""
        ck = ("tiles", row_bounds, rows)
        t = self._segs.get(ck)
        if t is None:
            tab = [(g, r0, min(rows, row_bounds[g + 1] - r0)) for g in range(len(row_bounds) - 1)
                   for r0 in range(row_bounds[g], row_bounds[g + 1], rows)]
            t = (h2d(np.asarray(tab if tab else [(0, 0, 0)], dtype=np.int32), torch.int32, self.nbr.device), len(tab))
            self._segs[ck] = t
        return t

    def segments(self, maxlen, row_bounds=None):
        """int32 [nseg,3] (weight index, start, count<=maxlen) covering every pair; cached.
        Weight index = g*K + k for group g, offset k (weights stacked as [G*K, cin, cout])."""
        ck = (maxlen, row_bounds)
        seg = self._segs.get(ck)
        if seg is None:
            pin, _, off, _ = self.pairs(row_bounds)
            G = 1 if row_bounds is None else len(row_bounds) - 1
            # by the library (cg3d_host_segments: C, outside the interpreter lock) or by its numpy specification
            # (tests/test_me_host.py compares them entry for entry)
            build = _segments_native if SEG_NATIVE else _segments_numpy
            tab = build(off, self.K, G, maxlen, SEG_XCD_ORDER and _lib.get().is_device, row_bounds, self.n_out)
            seg = (h2d(torch.from_numpy(tab), torch.int32, pin.device), int(tab.shape[0]))
            self._segs[ck] = seg
        return seg


SEG_XCD_ORDER = __import__("os").environ.get("CG3D_SEG_XCD", "1") != "0"
SEG_NATIVE = __import__("os").environ.get("CG3D_SEG_NATIVE", "1") != "0"


def _segments_numpy(off, K, G, maxlen, xcd, row_bounds, n_out):
    """The segment table of `KernelMap.segments` (the specification of cg3d_host_segments)."""
    nslot = K * G
    counts = off[1:] - off[:-1]
    nseg_s = (counts + maxlen - 1) // maxlen
    slot = np.repeat(np.arange(nslot, dtype=np.int64), nseg_s)
    first = np.repeat(np.cumsum(nseg_s) - nseg_s, nseg_s)
    start = off[slot] + (np.arange(slot.shape[0], dtype=np.int64) - first) * maxlen
    cnt = np.minimum(maxlen, off[slot + 1] - start)
    widx = (slot % G) * K + slot // G
    tab = np.stack([widx, start, cnt], 1).astype(np.int32)
    if xcd and 64 <= tab.shape[0] <= 4096 and maxlen >= 256:
        tab = tab[_xcd_order(slot, start, cnt, off, G, row_bounds, n_out)]
    return tab


def _segments_native(off, K, G, maxlen, xcd, row_bounds, n_out):
    """int32 [nseg, 3] segment table of `KernelMap.segments` built by cg3d_host_segments."""
    off = np.ascontiguousarray(off, dtype=np.int64)
    cap = int(K) * int(G) + int(off[-1] - off[0]) // int(maxlen) + 1
    out = np.empty((max(cap, 1), 3), dtype=np.int32)
    rb = None if row_bounds is None else np.asarray(row_bounds, dtype=np.int64)
    n = ctypes.c_int64(0)
    rc = _lib.get().raw("cg3d_host_segments")(off.ctypes.data, int(K), int(G), int(maxlen), 1 if xcd else 0,
                                              None if rb is None else rb.ctypes.data, int(n_out), out.ctypes.data, cap,
                                              ctypes.cast(ctypes.pointer(n), ctypes.c_void_p))
    if rc != 0:
        raise _lib.CG3DError("cg3d_host_segments failed with status %d" % rc)
    return out[:n.value]
SELF_MAP_HALF = __import__("os").environ.get("CG3D_SELF_MAP_HALF", "1") != "0"
# Off by default: measured on MI355X (82107 rows, 128 -> 128) the row-block order cuts the memory-side traffic of the launch
# from 3.4 x to 1.4 x the tensors' bytes (L2 hit rate of the gathers 33 % -> 68 %) but runs 97 us against 78 us -- 2.7 x more
# workgroups, each with its 64 KB atomic epilogue and a pipeline fill; the kernel is bound by dependent latencies
# (SQ_WAIT_ANY 46 % of wave cycles, MFMA busy 17 %, LDS 34 %), not by where its rows come from.
WGRAD_ROW_BLOCKS = __import__("os").environ.get("CG3D_WGRAD_ROW_BLOCKS", "0") != "0"
WGRAD_BLOCK_ROWS = int(__import__("os").environ.get("CG3D_WGRAD_BLOCK_ROWS", "2048"))
WGRAD_BLOCK_MIN_ROWS = 16384
N_XCD = 8


def _xcd_order(slot, start, cnt, off, G, row_bounds, n_out):
    """Launch order of the segments of a pair-list kernel (workgroup i runs segment i) that keeps every XCD on ONE row
    range of the tensors.  MI355X hands workgroup i to XCD i % 8, each with its own 4 MB L2; in (offset, row) order an XCD
    gets every 8th segment of every offset -- it streams ALL rows once per offset (27 x the tensor through the fabric,
    no reuse: 5 MB per sweep against 4 MB of L2).  Here the segments are sorted by the position of their rows (the pairs
    of one offset are in output-row order, so the position is estimated from the segment's place in its offset's
    list), cut into 8 equal runs and dealt out round-robin: XCD x sees the 27 offsets of the x-th eighth of the rows,
    sweeping it front to back, and finds the neighbour rows another offset just fetched in its own L2."""
    nslot_cnt = np.maximum(off[slot + 1] - off[slot], 1).astype(np.float64)
    pos = (start - off[slot] + 0.5 * cnt) / nslot_cnt                   # 0..1 within the slot's (offset, group) list
    if row_bounds is not None:
        rb = np.asarray(row_bounds, dtype=np.float64)
        g = slot % G
        pos = (rb[g] + pos * (rb[g + 1] - rb[g])) / max(float(n_out), 1.0)
    order = np.argsort(pos, kind="stable")                              # by position, ties in offset order
    n = order.shape[0]
    run = -(-n // N_XCD)                                                # rank i -> run i // run, place i % run in it
    dest = (np.arange(n) % run) * N_XCD + np.arange(n) // run           # a run shorter than the others leaves holes:
    out = np.full(run * N_XCD, -1, dtype=np.int64)
    out[dest] = order
    return out[out >= 0]                                                # ... closed up (only the last few places shift)


def _build_map_begin(coords_i32, qstride):
    """The launch half of `_build_map`: everything but the host read of the row count."""
    lib = _lib.get()
    lib.check(coords_i32)
    n = coords_i32.shape[0]
    dev = coords_i32.device
    cap = int(lib.raw("cg3d_hash_capacity")(n))
    keys = torch.empty(cap, dtype=torch.int64, device=dev)
    vals = torch.empty(cap, dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(lib.raw("cg3d_coord_map_ws_bytes")(n)) // 4, 1), dtype=torch.int32, device=dev)
    out_coords = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
    uniq = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    inv = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    n_out = torch.empty(2, dtype=torch.int32, device=dev)          # [row count, status], both written by the call
    lib.call("cg3d_coord_map_build", ptr(coords_i32), c_int64(n), c_int32(qstride), ptr(keys), ptr(vals),
             c_int64(cap), ptr(ws), ptr(out_coords), ptr(uniq), ptr(inv), ptr(n_out), lib.stream())
    return (n, cap, keys, vals, out_coords, uniq, inv, n_out)


def _build_map_finish(state, m, status):
    n, cap, keys, vals, out_coords, uniq, inv, _ = state
    if status != 0:
        raise _lib.CG3DError("cg3d_coord_map_build: a coordinate or batch index does not fit the packed voxel key "
                             "(|x|,|y|,|z| < %d, 0 <= batch < %d): the rows would be dropped silently" % (16384, 524288))
    return out_coords[:m], keys, vals, cap, uniq[:m], inv[:n]


def _build_map(coords_i32, qstride, assume_unique=False):
    """coords int32 [n,4] -> (out_coords [m,4], keys, vals, cap, unique_index [m], inverse [n]).
    assume_unique: the caller vouches that the rows are distinct, in-range voxels (the rows of another map of this step):
    m = n without the host read of the count."""
    state = _build_map_begin(coords_i32, qstride)
    if assume_unique and qstride == 1:
        return _build_map_finish(state, state[0], 0)
    m, status = state[7].tolist()  # host sync: the row count sizes every later tensor on this map
    return _build_map_finish(state, m, status)


# Row order of every map inserted from raw coordinates: (batch, Morton(x, y, z)) instead of the order the points arrive in
# (cg3d_morton_order).  The strided maps inherit it.  128 consecutive rows are then a spatially compact patch, which is
# what the LDS-staged tile kernel needs (cg3d_tile_plan_build); no result of the path depends on the row order beyond
# fp32 summation order, and `unique_index` / `inverse_mapping` keep referring to the caller's rows.
MORTON_ROWS = __import__("os").environ.get("CG3D_MORTON_ROWS", "1") != "0"


def _transpose_map(nbr, n_in):
    """nbrT [K, n_in] from nbr [K, n_out] by scattering (cg3d_kernel_map_transpose): no hash lookups."""
    lib = _lib.get()
    K, n_out = nbr.shape
    out = torch.empty((K, max(n_in, 1)), dtype=torch.int32, device=nbr.device)
    lib.call("cg3d_kernel_map_transpose", ptr(nbr), c_int32(K), c_int64(n_out), c_int64(n_in), ptr(out), lib.stream())
    return out[:, :n_in].contiguous() if n_in > 0 else out[:, :0]


def _morton_order(coords_i32):
    """int64 [n]: input row of the i-th row in (batch, Morton) order."""
    lib = _lib.get()
    n = coords_i32.shape[0]
    dev = coords_i32.device
    order = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    ws = torch.empty(max(int(lib.raw("cg3d_morton_order_ws_bytes")(n)), 16), dtype=torch.uint8, device=dev)
    lib.call("cg3d_morton_order", ptr(coords_i32), c_int64(n), ptr(order), ptr(ws), lib.stream())
    return order[:n].long()


_CACHE_LOCK = __import__("threading").RLock()


def _cached(cache, key, build, limit=None):
    """Get-or-build on one of the module-level host caches (offset tables, chunk tables, identity pair lists ...).  The
    coordinate-prefetch worker (detectors/cagroup3d.py::_PrefetchWorker) runs the same code as the main thread, so lookup,
    the size-triggered `clear()` and the insert happen under ONE lock.  An entry built on the WORKER thread (its uploads
    go through the side stream) is published only after that stream has finished them: the main thread would otherwise read
    the table on ITS stream with nothing ordering it after the copy.  (The wait blocks only the worker, once per key.)"""
    with _CACHE_LOCK:
        hit = cache.get(key)
        if hit is not None:
            return hit
        hit = build()
        worker = (__import__("threading").current_thread() is not __import__("threading").main_thread()
                  and torch.cuda.is_available() and _lib.get().is_device)
        if not worker:
            if limit is not None and len(cache) > limit:
                cache.clear()
            cache[key] = hit
            return hit
    # worker thread: wait for its uploads OUTSIDE the lock (every cache lookup of the main thread would stall behind the
    # wait otherwise), then publish; if the main thread built the same key meanwhile, its entry stays
    torch.cuda.current_stream().synchronize()
    with _CACHE_LOCK:
        other = cache.get(key)
        if other is not None:
            return other
        if limit is not None and len(cache) > limit:
            cache.clear()
        cache[key] = hit
        return hit


_offset_cache = {}


def _offsets(kernel_size, spacing, device):
    ck = (int(kernel_size), int(spacing), str(device))
    return _cached(_offset_cache, ck, lambda: _make_offsets(kernel_size, spacing, device))


def _neg_offsets(kernel_size, spacing, device):
    """The negated table (the other direction of a kernel map), cached like the table itself."""
    ck = (int(kernel_size), int(spacing), str(device), "neg")
    return _cached(_offset_cache, ck, lambda: (-_offsets(kernel_size, spacing, device)).contiguous())


def _make_offsets(kernel_size, spacing, device):
    ks = int(kernel_size)
    rng = range(-(ks // 2), ks // 2 + 1) if ks % 2 == 1 else range(0, ks)
    offs = [(x * spacing, y * spacing, z * spacing) for x, y, z in itertools.product(rng, rng, rng)]
    return h2d(offs, torch.int32, device).contiguous()


class CoordinateManager:
    """Owns the coordinate maps and kernel maps of one batch (ME: CoordinateManager)."""

    def __init__(self):
        self._maps = {}
        self._strided = {}
        self._kmaps = {}
        self._uid = itertools.count()

    # -- maps
    def insert(self, coords_i32, tensor_stride=1, sort=False, assume_unique=False):
        """assume_unique: the rows are the (distinct) rows of another map -- no host read of the row count.
        sort: build the map in (batch, Morton) row order (SparseTensor construction: the features are re-indexed through
        `unique_index` / `inverse_mapping` anyway).  Maps at caller-given output coordinates (`conv(x, coordinates)`) keep
        the caller's order: row i of the result belongs to coordinate i."""
        begun = self._insert_begin(coords_i32, sort)
        state = begun[1]
        if assume_unique:
            m, status = state[0], 0
        else:
            m, status = state[7].tolist()  # host sync: the row count sizes every later tensor on this map
        return self._insert_finish(begun, m, status, tensor_stride)

    def _insert_begin(self, coords_i32, sort):
        """Launch half of `insert` (ordering + map build); `_insert_finish` needs the map's (row count, status) from the host."""
        coords_i32 = coords_i32.contiguous()
        order = None
        if sort and MORTON_ROWS and coords_i32.shape[0] > 1:
            _lib.get().check(coords_i32)
            order = _morton_order(coords_i32)
            coords_i32 = coords_i32[order].contiguous()
        return (order, _build_map_begin(coords_i32, 1))

    def _insert_finish(self, begun, m, status, tensor_stride):
        order, state = begun
        out, keys, vals, cap, uniq, inv_s = _build_map_finish(state, m, status)
        if order is not None:
            uniq = order[uniq.long()].to(torch.int32)           # representative rows / inverse map in the caller's row numbering
            inv = torch.empty_like(inv_s)
            inv[order] = inv_s
        else:
            inv = inv_s
        key = CoordinateMapKey(tensor_stride, next(self._uid))
        self._maps[key] = _CoordMap(out, keys, vals, cap, out.shape[0], int(tensor_stride))
        return key, uniq, inv

    def get(self, key):
        return self._maps[key]

    def stride(self, in_key, factor):
        """Map of tensor stride ts*factor: floor(c / s) * s, de-duplicated (strided conv / pool)."""
        ck = (in_key, int(factor))
        if ck not in self._strided:
            src = self._maps[in_key]
            new_ts = src.tensor_stride * int(factor)
            out, keys, vals, cap, _, _ = _build_map(src.coords, new_ts)
            key = CoordinateMapKey(new_ts, next(self._uid))
            self._maps[key] = _CoordMap(out, keys, vals, cap, out.shape[0], new_ts)
            self._strided[ck] = key
        return self._strided[ck]

    # -- kernel maps
    @staticmethod
    def _lookup_map(q_coords, table, offsets, onto_itself=False):
        """onto_itself: the queries are the table's own rows and the offsets a centred odd kernel -- half the lookups."""
        lib = _lib.get()
        K, nq = offsets.shape[0], q_coords.shape[0]
        nbr = torch.empty((K, max(nq, 1)), dtype=torch.int32, device=q_coords.device)
        lib.check(q_coords, offsets)
        lib.call("cg3d_kernel_map_self" if (onto_itself and SELF_MAP_HALF and K % 2 == 1 and nq > 0) else "cg3d_kernel_map",
                 ptr(q_coords), c_int64(nq), ptr(offsets), c_int32(K), ptr(table.keys),
                 ptr(table.vals), c_int64(table.cap), ptr(nbr), lib.stream())
        return nbr[:, :nq] if nq > 0 else nbr[:, :0]

    def kernel_map(self, in_key, out_key, kernel_size, dilation=1, transpose=False):
        ck = (in_key, out_key, int(kernel_size), int(dilation), bool(transpose))
        km = self._kmaps.get(ck)
        if km is None:
            src, dst = self._maps[in_key], self._maps[out_key]
            if not transpose:
                offs = _offsets(kernel_size, src.tensor_stride * dilation, src.coords.device)
                fwd_off = offs                                     # o + off = i   (the other direction: i - off = o)
            else:
                offs = _offsets(kernel_size, dst.tensor_stride * dilation, src.coords.device)
                fwd_off = _neg_offsets(kernel_size, dst.tensor_stride * dilation, src.coords.device)      # o - off = i
            nbr = self._lookup_map(dst.coords, src, fwd_off, in_key == out_key and int(kernel_size) % 2 == 1)
            # the lazy transposed map must not close over `self`: manager -> _kmaps -> KernelMap -> closure -> manager
            # is a reference cycle, and every step's coordinate structures (0.5 GB of device tensors at S50k x 4) then
            # live until the cyclic collector's next gen-2 pass -- tens of GB of garbage in a long run
            nbr = nbr.contiguous()
            n_src = src.n
            km = KernelMap(nbr, offs.shape[0], src.n, dst.n, lambda: _transpose_map(nbr, n_src))
            km.same_map = in_key == out_key          # row groups of the output are row groups of the input
            # a map onto itself with a centred odd kernel: off[K-1-k] == -off[k], hence nbrT[k] == nbr[K-1-k] -- no second
            # hash lookup for the transposed map, and the tile kernel's data gradient reuses the forward plan (wrev)
            km.symmetric = km.same_map and int(kernel_size) % 2 == 1
            if km.symmetric:
                km._make_T = lambda n=km.nbr: n.flip(0).contiguous()
            self._kmaps[ck] = km
        return km


class SparseTensorQuantizationMode(Enum):
    RANDOM_SUBSAMPLE = 0
    UNWEIGHTED_AVERAGE = 1


# ----------------------------------------------------------------------------- autograd ops
def _conv_fwd_raw(x, w3, nbr, bias, n_out):
    """Output-stationary implicit-GEMM form on the dense map (kept for A/B and tests)."""
    lib = _lib.get()
    K, cin, cout = w3.shape
    lib.check(x, w3, nbr, bias)
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    lib.call("cg3d_spconv_fwd", ptr(x), ptr(w3), ptr(nbr), ptr(bias), ptr(y), c_int64(x.shape[0]), c_int64(n_out),
             c_int32(K), c_int32(cin), c_int32(cout), c_int32(0), lib.stream())
    return y


FWD_SEG = 128  # pairs per workgroup of cg3d_spconv_pairs_fwd

# Operand precision of the sparse convolutions' forward / data-gradient MFMAs:
#   0 = fp32 operands (v_mfma_f32_32x32x2_f32, exact products) -- the parity configuration;
#   1 = bf16 operands, fp32 accumulate (v_mfma_f32_32x32x16_bf16) -- BASELINE.json configs[1] "bf16 backbone";
#   3 = SPLIT bf16 operands ("bf16x3"): x = hi + lo, x w = xhi whi + xlo whi + xhi wlo -- three bf16 MFMA products per fp32
#       product, fp32-accurate to ~1e-5 relative (include/cagroup3d_hip.h, cg3d_to_bf16_split).  Rows become [hi | lo | hi]
#       (3 c channels), the contraction side of the weights [Whi ; Whi ; Wlo], and every bf16 kernel runs unchanged on the three
#       times longer contraction.  The two heads run under it (HEAD_PRECISION = 3): configs[1] words its precision as "bf16
#       backbone", the reference's heads are fp32 (cagroup_head.py:227-282), and fp32 MFMA operands cost 16 x the bf16 rate.
# Weights, parameter gradients and statistics stay fp32 in all modes; so do the feature rows of the per-layer path of this file.
# Inside the backbone's launch program (engine.ACT_BF16, default on) activations and activation gradients are STORED as bf16
# rows only (DESIGN.md section 2 "Features, round 5"); CG3D_ACT_BF16=0 keeps fp32 rows there too.
PRECISION = 0
PREC_SPLIT = 3
# Precision of a PART of the step: `precision_scope(p)` overrides PRECISION for the calling thread (the detector runs the two
# heads under `HEAD_PRECISION` when that is set: BASELINE.json configs[1] words its precision as "bf16 backbone").  A Function
# records the precision its forward ran under and its backward runs under the same one, whichever thread the autograd
# engine calls it on; the coordinate-prefetch worker never sees another thread's override.
HEAD_PRECISION = None
_tls_prec = __import__("threading").local()
HEAD_MODES = {"split": 3, "fp32": 0, "bf16": None}


def heads_from_env(default="split"):
    """HEAD_PRECISION for a bf16 run from CG3D_HEADS = split | fp32 | bf16 (the dev tools: the bench's default is split)."""
    return HEAD_MODES[__import__("os").environ.get("CG3D_HEADS", default)]


def _prec():
    v = getattr(_tls_prec, "v", None)
    return PRECISION if v is None else v


class precision_scope:
    def __init__(self, p):
        self.p = p

    def __enter__(self):
        self.old = getattr(_tls_prec, "v", None)
        if self.p is not None:
            _tls_prec.v = self.p
        return self

    def __exit__(self, *exc):
        _tls_prec.v = self.old
        return False


def _ctx_precision(fn):
    """Decorator of a Function.backward: runs it under the precision its forward recorded in ctx.prec."""
    def wrapped(ctx, *a):
        old = getattr(_tls_prec, "v", None)
        _tls_prec.v = getattr(ctx, "prec", None)
        try:
            return fn(ctx, *a)
        finally:
            _tls_prec.v = old
    return wrapped


def _use_bf16(cin):
    """bf16 MFMA operands (plain or split) for a contraction over `cin` channels."""
    return _prec() in (1, 3) and cin % 8 == 0 and cin >= 16


def _split():
    return _prec() == 3


def _kx():
    """Length of the contraction the kernels see, in units of the layer's own: 3 in the split precision."""
    return 3 if _prec() == 3 else 1


def _want_rows16(c):
    """Whether the BatchNorm / ReLU apply kernels should leave the PLAIN bf16 copy of their output rows (the operand of the
    next convolution in the bf16 precision; split operands are written by their own pass, cg3d_to_bf16_split)."""
    return BF16_ROWS and _prec() == 1 and c % 8 == 0 and c >= 16


# bf16 mode: maps whose neighbourhood occupancy P / (K * n_out) is at least this run the output-stationary
# kernel (no atomics, one plain store per output row); sparser maps keep the pair form
IMPLICIT_MIN_OCCUPANCY = float(__import__("os").environ.get("CG3D_IMPLICIT_THR", "0.1"))
IMPLICIT_MIN_TILES = int(__import__("os").environ.get("CG3D_IMPLICIT_TILES", "96"))


def _conv_implicit_bf16(x, w_bf16_t, nbr, bias, n_out, cin, cout, n_pairs, tiles=None):
    """Y[o] = bias + sum_k X[nbr[k, o]] @ W[k] with bf16 operands; w_bf16_t: int16 view of bf16 [K, cout, cin];
    x: fp32 rows (rounded in the kernel) or their int16/bf16 copy from _to_bf16."""
    lib = _lib.get()
    K = nbr.shape[0]
    lib.check(x, w_bf16_t, nbr, bias)
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    prof = KernelProfile.enabled and lib.is_device
    if prof:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rows16 = x.dtype == torch.int16
    kc = cin * _kx() if rows16 else cin        # the contraction the kernel sees (split rows: 3 cin)
    if tiles is None:
        lib.call("cg3d_spconv_fwd", ptr(x), ptr(w_bf16_t), ptr(nbr), ptr(bias), ptr(y), c_int64(x.shape[0]), c_int64(n_out),
                 c_int32(K), c_int32(kc), c_int32(cout), c_int32(2 if rows16 else 1), lib.stream())
    else:       # row groups with their own weights: w_bf16_t stacks G sets of K slots
        lib.call("cg3d_spconv_fwd_tiled", ptr(x), ptr(w_bf16_t), ptr(nbr), ptr(tiles[0]), c_int64(tiles[1]), ptr(bias), ptr(y),
                 c_int64(x.shape[0]), c_int64(n_out), c_int32(K), c_int32(kc), c_int32(cout), c_int32(2 if rows16 else 1),
                 lib.stream())
    if prof:
        ev1.record()
        xb = 2.0 if rows16 else 4.0     # bytes per gathered element
        # SURVEY 8(d) bytes: every tensor once (input rows, output rows, weights, the dense map); last field: the per-pair
        # figure (a gathered row counted once per pair it takes part in) round 1 reported
        KernelProfile.records.append((ev0, ev1, 2.0 * n_pairs * cin * cout,
                                      xb * x.shape[0] * cin + 4.0 * n_out * cout + 2.0 * K * cin * cout + 4.0 * K * n_out,
                                      ("implicit_bf16" + _ksuffix(), K, cin, cout, n_pairs, n_out, 0),
                                      xb * n_pairs * cin + 4.0 * n_out * cout + 2.0 * K * cin * cout + 4.0 * K * n_out))
    return y


BF16_ROWS = __import__("os").environ.get("CG3D_BF16_ROWS", "1") != "0"   # bf16 mode: gather from bf16 row copies


# ----------------------------------------------------------------------------- tile plans (cg3d_spconv_tile_fwd)
FUSED_BN_STATS = __import__("os").environ.get("CG3D_FUSED_BN_STATS", "1") != "0"
class _ZeroArena:
    """Zero-filled fp32 scratch for the per-layer statistics tables (BatchNorm sums, their backward counterparts): ONE
    `torch.zeros` per step and thread instead of a fill launch per layer.  `take(n)` hands out 16-byte aligned slices of the
    current block; a new block is opened when it is used up or at `reset()` (the detector's forward).  Blocks are never
    written back to zero: a slice is handed out once, and whoever holds it (an autograd context, a parameter gradient that
    is a view of it) keeps its block alive."""
    BLOCK = 1 << 21             # floats (8 MB: the step's ~130 tables of BN_SLOTS x 2 x G x C)

    def __init__(self):
        self.buf, self.off = None, 0

    def reset(self):
        self.buf, self.off = None, 0

    def take(self, n, device):
        n4 = -(-int(n) // 4) * 4
        if self.buf is None or self.buf.device != device or self.off + n4 > self.buf.numel():
            self.buf = torch.zeros(max(self.BLOCK, n4), dtype=torch.float32, device=device)
            self.off = 0
        out = self.buf[self.off:self.off + int(n)]
        self.off += n4
        return out


BN_SLOTS = 16          # CG3D_BN_SLOTS of include/cagroup3d_hip.h: slots of a statistics table


def zero_arena():
    a = getattr(_TLS, "arena", None)
    if a is None:
        a = _TLS.arena = _ZeroArena()
    return a


_STATS = {}         # data_ptr of a conv output of THIS forward -> (partials, chunks, rows, channels, output); cleared with _ROWS16
# Whether a training-mode BatchNorm may follow the convolutions of the current forward (set by the detector's forward from
# `self.training`): in evaluation nobody consumes the partial sums, so they are neither computed nor kept (every entry
# holds its conv output alive)
WANT_BN_STATS = True
TILE_ROWS = 128
TILE_UCAP = int(__import__("os").environ.get("CG3D_TILE_UCAP", "511"))    # LDS rows per pass: (ucap+1) x 128 B = 64 KB


# rows per pass for K > 64.  255 would let two workgroups of the 128-offset-block kernel share a CU (32 KB of rows + 32 KB of table under
# the exchange's 64 KB) -- measured: 307-317 -> 515-541 us per launch on the 9^3 class maps (profiles/r06_ab_bigk_two_wgs.txt): a
# 128-row tile of a 9^3 map touches ~1 000 distinct rows, and twice the passes cost far more than the second workgroup returns
TILE_UCAP_BIGK = int(__import__("os").environ.get("CG3D_TILE_UCAP_BIGK", "511"))
TILE_SORT_ROWS = __import__("os").environ.get("CG3D_TILE_SORT_ROWS", "1") != "0"
TILE_SORT_MAX_OCCUPANCY = 0.2       # pairs / (K * rows) below which a map's tiles are cut from signature-sorted rows
# (window 128 = inside the tile: only useful to a kernel that skips dead 32-row blocks -- tried, not kept: spconv_tile2.hip)
TILE_SORT_IN_TILE = __import__("os").environ.get("CG3D_TILE_SORT_IN_TILE", "0") != "0"


def pairs_many(items):
    """`kmap.pairs(row_bounds)` for several (kmap, row_bounds) at once: all counting launches first, ONE host read of all the
    offset tables, then the fill launches -- the class branches need four lists in a row (one blocking read instead of four)."""
    todo = []
    for km, rb in items:
        if km._pairs is None:
            km._pairs = {}
        if rb not in km._pairs and not any(k is km and r == rb for k, r, _ in todo):
            todo.append((km, rb, km._pairs_count(rb)))
    if todo:
        host = torch.cat([p[1] for _, _, p in todo]).cpu().numpy() if len(todo) > 1 else todo[0][2][1].cpu().numpy()
        o = 0
        for km, rb, p in todo:
            n = p[1].shape[0]
            km._pairs_finish(rb, p, host[o:o + n])
            o += n
    return [km.pairs(rb) for km, rb in items]


class TilePlan:
    """A kernel map re-encoded per tile of 128 output rows (include/cagroup3d_hip.h, cg3d_tile_plan_build)."""
    __slots__ = ("slots", "live", "pass_tab", "npass", "ulist", "cursor", "maxpass", "ucap", "ntile", "tiles", "K", "n_out", "order")

    def tensors(self):
        return [self.slots, self.live, self.pass_tab, self.npass, self.ulist, self.cursor, self.tiles, self.order]


def build_tile_plan(nbr, n_pairs, tiles=None, ucap=None, sort_rows=False):
    """nbr int32 [K, n_out] (k-major) -> TilePlan.  `n_pairs` (host int, >= the number of nbr >= 0) sizes `ulist`;
    tiles: None or (device int32 [ntile,3], ntile); sort_rows: 0 / False, or the window (True = 1024, or 128) of
    cg3d_tile_row_order: the tiles are cut from the permuted rows (plan.order: position -> output row)."""
    lib = _lib.get()
    lib.check(nbr)
    K, n_out = nbr.shape
    dev = nbr.device
    p = TilePlan()
    p.K, p.n_out = K, n_out
    p.ucap = int(ucap or TILE_UCAP)
    p.tiles = tiles[0] if tiles is not None else None
    p.ntile = int(tiles[1]) if tiles is not None else -(-n_out // TILE_ROWS)
    p.maxpass = K
    nt = max(p.ntile, 1)
    p.slots = torch.empty((nt, K, TILE_ROWS), dtype=torch.int16, device=dev)
    p.live = torch.empty((nt, K), dtype=torch.uint8, device=dev)
    p.pass_tab = torch.empty((nt, p.maxpass, 4), dtype=torch.int32, device=dev)
    p.npass = torch.empty(nt, dtype=torch.int32, device=dev)
    p.ulist = torch.empty(max(int(n_pairs), 1), dtype=torch.int32, device=dev)
    p.cursor = torch.empty(2, dtype=torch.int32, device=dev)
    p.order = None
    if sort_rows and tiles is None and K <= 32 and n_out > 0:
        p.order = torch.empty(n_out, dtype=torch.int32, device=dev)
        lib.call("cg3d_tile_row_order", ptr(nbr), c_int32(K), c_int64(n_out), c_int32(1024 if sort_rows is True else int(sort_rows)),
                 ptr(p.order), lib.stream())
    lib.call("cg3d_tile_plan_build", ptr(nbr), c_int32(K), c_int64(n_out), ptr(p.tiles), c_int64(p.ntile), c_int32(p.ucap),
             c_int32(p.maxpass), ptr(p.slots), ptr(p.live), ptr(p.pass_tab), ptr(p.npass), ptr(p.ulist),
             c_int64(p.ulist.shape[0]), ptr(p.cursor), ptr(p.order), lib.stream())
    return p


def _conv_tile(x16, wf, plan, bias, cin, cout, n_in, n_pairs=0, ksplit=1, groups=1, wrev=False, want_stats=False):
    """Y[o] = bias + sum_k X[nbr[k, o]] @ W[k] through a tile plan; x16: int16 view of the bf16 rows, wf: the weights in
    MFMA fragment order (cg3d_spconv_prep_weights_frag)."""
    lib = _lib.get()
    lib.check(x16, wf, bias)
    y = torch.empty((plan.n_out, cout), dtype=torch.float32, device=x16.device)
    stats = None
    if want_stats and WANT_BN_STATS and FUSED_BN_STATS and ksplit == 1 and plan.tiles is None and cout <= 512 and plan.ntile > 0:
        # per-workgroup sum / sum of squares of the output channels, accumulated while the tiles are stored: the BatchNorm
        # that follows derives mean / variance from this table instead of reading Y again (cg3d_bn_apply_sums)
        stats = zero_arena().take(BN_SLOTS * 2 * cout, x16.device)       # the layer's zero-based statistics table [slots][2][cout]
        if len(_STATS) > 64:
            _STATS.clear()
        _STATS[y.data_ptr()] = (stats, 1, plan.n_out, cout, y)           # holds y: its address cannot be reused while the entry lives
    prof = KernelProfile.enabled and lib.is_device
    if prof:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    lib.call("cg3d_spconv_tile_fwd", ptr(x16), ptr(wf), ptr(plan.slots), ptr(plan.live), ptr(plan.pass_tab), ptr(plan.npass),
             ptr(plan.ulist), c_int32(plan.maxpass), c_int32(plan.ucap), ptr(plan.tiles), c_int64(plan.ntile), ptr(plan.order), ptr(bias), ptr(y),
             c_int64(n_in), c_int64(plan.n_out), c_int32(plan.K), c_int32(cin * _kx()), c_int32(cout), c_int32(ksplit),
             c_int32(1 if wrev else 0), ptr(stats), lib.stream())
    if prof:
        ev1.record()
        # SURVEY 8(d) bytes: every input row once, every output row once, the weights once, the map once (2-byte slots)
        kx = _kx()          # (split operands: the rows and the weights the launch reads are three times as long)
        wbytes = 2.0 * kx * groups * plan.K * cin * cout
        KernelProfile.records.append((ev0, ev1, 2.0 * n_pairs * cin * cout,
                                      2.0 * kx * n_in * cin + 4.0 * plan.n_out * cout + wbytes + 2.0 * plan.K * plan.n_out,
                                      ("tile_bf16" + _ksuffix(), plan.K, cin, cout, n_pairs, plan.n_out, 0),
                                      2.0 * kx * n_pairs * cin + 4.0 * plan.n_out * cout + wbytes + 2.0 * plan.K * plan.n_out))
    return y


def _prep_frag(w3, want_t=True, want_p=False):
    """fp32 [K, cin, cout] -> (Wf_t, Wf) int16 views of the bf16 weights in MFMA fragment order (either may be None);
    answered from the step's weight arena when the layer is recorded there."""
    e = _planned_single(w3, want_p, True)
    if e is not None:
        return (e[3] if want_t else None), (e[4] if want_p else None)
    lib = _lib.get()
    K, cin, cout = w3.shape
    x = _kx()
    wt = torch.empty((K, cout, cin * x), dtype=torch.int16, device=w3.device) if want_t else None
    wp = torch.empty((K, cin, cout * x), dtype=torch.int16, device=w3.device) if want_p else None
    if x == 3:
        lib.call("cg3d_spconv_prep_weights_split", ptr(w3), ptr(None), ptr(wt), ptr(wp), c_int32(1), c_int64(K), c_int32(cin),
                 c_int32(cout), c_int32(1), lib.stream())
    else:
        lib.call("cg3d_spconv_prep_weights_frag", ptr(w3), ptr(None), ptr(wt), ptr(wp), c_int32(1), c_int64(K), c_int32(cin),
                 c_int32(cout), lib.stream())
    return wt, wp


# Coordinate-only dry run (CAGroup3D.prefetch_coordinates): modules build every coordinate / kernel map, pair list and
# segment table they will need -- with their host reads -- but launch no feature kernel and return uninitialised
# feature tensors of the right shape.  Run on a side stream for the NEXT batch while the GPU is still busy with the
# current step's backward, it takes the data-dependent host syncs out of the timed forward pass.
# The flag is PER THREAD: the dry run of batch i+1 may run on a worker thread (CAGroup3D.prefetch_coordinates_async)
# while the main thread is in the real forward / backward of batch i.
_TLS = __import__("threading").local()


def coords_only():
    return getattr(_TLS, "coords_only", False)


def set_coords_only(flag):
    _TLS.coords_only = bool(flag)


def __getattr__(name):              # `ME.COORDS_ONLY` (read-only view of this thread's flag)
    if name == "COORDS_ONLY":
        return coords_only()
    raise AttributeError(name)


def _fake(n, c, like):
    return torch.empty((n, c), dtype=torch.float32, device=like.device)


def _walk_tensors(o, seen):
    if torch.is_tensor(o):
        if o.is_cuda:
            seen.append(o)
    elif isinstance(o, (list, tuple)):
        for v in o:
            _walk_tensors(v, seen)
    elif isinstance(o, dict):
        for v in o.values():
            _walk_tensors(v, seen)
    elif isinstance(o, _CoordMap):
        _walk_tensors([o.coords, o.keys, o.vals, o._perms], seen)
    elif isinstance(o, KernelMap):
        _walk_tensors([o.nbr, o._nbrT, o._pairs, o._segs], seen)
    elif isinstance(o, TilePlan):
        _walk_tensors(o.tensors(), seen)


def release_to_stream(mgr, extra, stream):
    """Every tensor the coordinate manager holds was allocated on the prefetch stream; tell the caching allocator that
    `stream` uses them too (their blocks may only be recycled after that stream's pending work)."""
    seen = []
    _walk_tensors([mgr._maps, mgr._kmaps, extra], seen)
    for t in seen:
        t.record_stream(stream)


def record_cached(tables, stream):
    """Tables out of the host caches (built, possibly, on the prefetch stream) that a launch program will read on `stream`:
    tell the allocator once per tensor -- they are long-lived and shared by many programs, so the mark rides on the tensor."""
    seen = []
    _walk_tensors(tables, seen)
    sid = stream.cuda_stream
    for t in seen:
        if getattr(t, "_cg3d_rec", None) != sid:
            t.record_stream(stream)
            t._cg3d_rec = sid


# bf16 row copies written by the BatchNorm apply kernels, keyed by the address of the fp32 tensor they mirror.  The entry
# holds the fp32 tensor itself, so its storage cannot be recycled while the entry exists; entries are consumed by the
# convolution that gathers from them and the table is emptied at the start of every detector forward.
_ROWS16 = {}


def rows16_of(x, keep=False):
    e = _ROWS16.get(x.data_ptr()) if keep else _ROWS16.pop(x.data_ptr(), None)
    return e[1] if (e is not None and e[0].shape == x.shape and e[0].dtype == x.dtype) else None


_ROWS48 = {}       # split operand rows of forward activations of THIS forward (cleared with _ROWS16): data_ptr -> (x, rows)


def _to_split(x, keep=False):
    """fp32 [N, C] -> int16 [N, 3 C] split operand rows [hi | lo | hi] (cg3d_to_bf16_split).  keep: remember the copy for the
    other layers that read the same activation in this forward."""
    e = _ROWS48.get(x.data_ptr())
    if e is not None and e[0].shape == x.shape and e[0].dtype == x.dtype:
        return e[1]
    lib = _lib.get()
    n, c = x.shape
    out = torch.empty((n, 3 * c), dtype=torch.int16, device=x.device)
    lib.call("cg3d_to_bf16_split", ptr(x), ptr(out), c_int64(n), c_int32(c), lib.stream())
    if keep:
        if len(_ROWS48) > 64:
            _ROWS48.clear()
        _ROWS48[x.data_ptr()] = (x, out)
    return out


def _to_bf16(x, keep=False):
    """fp32 [N, C] -> int16 view of the bf16 rows (cg3d_to_bf16; one streaming pass, halves every later gather).
    keep: leave a BatchNorm-written copy registered (forward activations may feed several convolutions).
    Split precision: the [N, 3 C] split operand rows instead."""
    if _prec() == 3:
        return _to_split(x, keep)
    ready = rows16_of(x, keep)
    if ready is not None:
        return ready
    lib = _lib.get()
    out = torch.empty(x.shape, dtype=torch.int16, device=x.device)
    lib.call("cg3d_to_bf16", ptr(x), ptr(out), c_int64(x.numel()), lib.stream())
    return out


class _WeightPlan:
    """The bf16 weight copies of a whole step in ONE launch (cg3d_spconv_prep_weights_bf16_table).

    Self-learning: the first time a weight (or a group of per-class weights) asks for its copies it is converted on the
    spot and recorded; from then on `prepare_weights()` -- called by the detector at the start of every forward --
    converts every recorded weight whose tensor version changed (i.e. after every optimizer step, never in inference)
    into one persistent arena with a single launch, and the per-layer requests are answered from the arena."""
    singles = {}        # (data_ptr, kind) -> [w3, need_plain, version, wt_view, wp_view]; kind bit 0: MFMA fragment order (tile kernel),
    #                     bit 1: split operands (three-part contraction, cg3d_spconv_prep_weights_split)
    groups = {}         # (ptrs, transposed, kind) -> [weights, versions, out_view]
    table = None        # device int64 [nrows, 6]
    nrows = 0
    dirty = False
    keep = None         # the arenas
    live = False        # inside a detector forward that called prepare_weights(): arena answers are valid
    pending = True      # the weights may have changed since the last conversion
    gen = 0             # bumped whenever the arenas are rebuilt (addresses handed out before are stale: engine.Compiled.usable)
    # Early / late rows.  The first module of the detector (the backbone) needs its own weights' copies only; everything else --
    # six sevenths of the model's 126 M parameters sit in the class branches -- is first read milliseconds later.  `early` is the
    # set of data_ptrs of the weights the first module reads (set_early_weights); their rows lead the table and are converted on
    # the current stream, the rest on the late stream (late_stream()), behind the optimizer's update of the same parameters
    # (optim.ClippedAdamW: the late rows of ITS table run there too) and beside the backbone's forward pass.
    early = None        # None: no split
    n_early = 0         # table rows [0, n_early) are early

    @classmethod
    def reset(cls):
        with _CACHE_LOCK:
            cls.singles, cls.groups, cls.table, cls.nrows, cls.dirty, cls.keep = {}, {}, None, 0, False, None
            cls.live, cls.pending = False, True
            cls.n_early = 0
            cls.gen += 1

    @classmethod
    def _rebuild(cls, device):
        import numpy as _np
        def kx(kind):
            return 3 if kind & 2 else 1
        n_t = sum(e[0].numel() * kx(k[1]) for k, e in cls.singles.items()) + sum(sum(w.numel() for w in g[0]) * kx(k[2]) for k, g in cls.groups.items() if k[1])
        n_p = sum(e[0].numel() * kx(k[1]) for k, e in cls.singles.items() if e[1]) + sum(sum(w.numel() for w in g[0]) * kx(k[2]) for k, g in cls.groups.items() if not k[1])
        arena_t = torch.empty(max(n_t, 1), dtype=torch.int16, device=device)
        arena_p = torch.empty(max(n_p, 1), dtype=torch.int16, device=device)
        rows, ot, op = [], 0, 0
        late_rows = []
        early = cls.early

        def add(w, off_t, off_p, kind=0):
            K, cin, cout = w.shape
            per = cin * cout
            x = kx(kind)                                  # a slot's copies are 3 cin cout elements each in the split form
            tiles = -(-cin // 64) * -(-cout // 64)
            k = _np.repeat(_np.arange(K, dtype=_np.int64), tiles)
            t = _np.tile(_np.arange(tiles, dtype=_np.int64), K)
            r = _np.empty((K * tiles, 6), dtype=_np.int64)
            r[:, 0] = w.data_ptr() + k * per * 4
            r[:, 1] = 0 if off_t is None else arena_t.data_ptr() + (off_t + k * per * x) * 2
            r[:, 2] = 0 if off_p is None else arena_p.data_ptr() + (off_p + k * per * x) * 2
            r[:, 3], r[:, 4], r[:, 5] = cin, cout, t | ((3 << 29) if kind & 1 else 0) | ((1 << 28) if kind & 2 else 0)
            (rows if (early is None or w.data_ptr() in early) else late_rows).append(r)
        for (_, kind), e in cls.singles.items():
            w = e[0]
            K, cin, cout = w.shape
            x = kx(kind)
            add(w, ot, op if e[1] else None, kind)
            e[3] = arena_t[ot:ot + w.numel() * x].view(K, cout, cin * x)
            ot += w.numel() * x
            if e[1]:
                e[4] = arena_p[op:op + w.numel() * x].view(K, cin, cout * x)
                op += w.numel() * x
            e[2] = -1
        for (ptrs, transposed, kind), g in cls.groups.items():
            ws = g[0]
            K, cin, cout = ws[0].shape
            x = kx(kind)
            base = ot if transposed else op
            for i, w in enumerate(ws):
                add(w, base + i * w.numel() * x if transposed else None, None if transposed else base + i * w.numel() * x, kind)
            tot = sum(w.numel() for w in ws) * x
            if transposed:
                g[2] = arena_t[ot:ot + tot].view(len(ws) * K, cout, cin * x)
                ot += tot
            else:
                g[2] = arena_p[op:op + tot].view(len(ws) * K, cin, cout * x)
                op += tot
            g[1] = None
        cls.n_early = sum(r.shape[0] for r in rows)
        rows += late_rows
        tab = _np.concatenate(rows) if rows else _np.zeros((0, 6), dtype=_np.int64)
        cls.table, cls.nrows = h2d(torch.from_numpy(tab), torch.int64, device), int(tab.shape[0])
        cls.keep, cls.dirty = (arena_t, arena_p), False
        cls.gen += 1


# Two ways of keeping the late parameters' work out of the device-bound half of the step (CG3D_LATE_WEIGHTS; both measured, both
# bit-identical to the one-launch forms, neither on by default):
#   defer   the late rows of the optimizer's update and of the weight conversion are NOT launched where the early rows are; they
#           wait in `_DEFERRED` and are launched -- on the late stream, in the order they were deferred -- by `run_late()`, which
#           the dense head calls right after its first blocking read: the device-bound half of the step is over there, the
#           host-bound half begins, and the device has room for the HBM streaming under the host's work (on the CURRENT stream
#           the ten blocking reads that follow each wait for it: no gain at all).  Five alternating 30-step bench pairs on a quiet
#           box: 23.58-23.66 ms per step against 23.55-23.83 (-0.14 ms, 0.6 %) -- for an optimizer step that returns with the class
#           branches' 73 M parameters not yet updated.  Not worth a default;
#   stream  the late rows on a stream of their own beside the next backbone forward (round 5, first try): median step 24.5 /
#           23.5 / 24.7 ms against 23.5 / 23.9 / 23.9 -- the HBM streaming beside the backbone's first layers slows them by what
#           it saves, and one more stream on four hardware queues adds outliers;
#   0       (default) one launch for all rows.
# Whatever reads a late parameter outside this order runs `run_late()` first: the arena look-ups of late weights do, an
# evaluation forward does at its start, `finish_weights()` and `optim.ClippedAdamW.finish_late()` do.
_LATE_STREAMS = {}
LATE_MODE = os.environ.get("CG3D_LATE_WEIGHTS", "0")
LATE_MODE = {"1": "stream", "": "0"}.get(LATE_MODE, LATE_MODE)
LATE_WEIGHTS = LATE_MODE in ("stream", "defer")
PREFETCH_THREAD_NAME = "cg3d-coordinate-prefetch"        # (pcdet/models/detectors/cagroup3d.py: the dry run's worker)
_DEFERRED = []                # launches of late rows, in order (LATE_MODE "defer")


def defer(fn):
    _DEFERRED.append(fn)


def run_late(join=True):
    """Launch the deferred late rows.  They go to the late stream (behind the current stream's position): the blocking reads that
    follow on the current stream -- ten of them between the head's first read and the class program -- must not wait for a
    millisecond of HBM streaming nobody needs yet.  join: the current stream waits for them (whoever is about to read a late
    parameter: `_late_needed`, `finish_weights`, the optimizer's `finish_late`); the head passes False right after its first
    read and lets the first look-up of a late weight join.  Only the issuing thread launches: the dry run's worker reads early
    weights only, and its current stream is not the step's."""
    import threading
    if threading.current_thread() is not threading.main_thread():
        # the dry run's worker only ever looks up early weights; anybody else on another thread (a checkpoint thread serialising
        # parameters after clip_and_step) would read class-branch parameters whose AdamW rows have not run: refuse, loudly
        if join and _DEFERRED and threading.current_thread().name != PREFETCH_THREAD_NAME:
            raise RuntimeError("cagroup3d_amd.me.run_late: deferred late-parameter work exists and the caller is not the issuing thread -- "
                               "call optimizer.finish_late() on the training thread before reading parameters elsewhere")
        return
    if _DEFERRED:
        dev = torch.cuda.current_device()
        ls = late_stream(dev)
        ls.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ls):
            while _DEFERRED:
                _DEFERRED.pop(0)()
        late_mark(dev)
    if join:
        late_weights_ready()


def _late_needed(w):
    """A look-up of a late weight's copies while late rows are still deferred or in flight: launch them / wait for them first."""
    if (_DEFERRED or _LATE_PENDING) and _WeightPlan.early is not None and w.data_ptr() not in _WeightPlan.early:
        run_late()


def late_stream(device):
    """This process's stream for work on parameters the step reads late (the heads' AdamW rows and weight copies)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _LATE_STREAMS.get(idx)
    if s is None:
        s = _LATE_STREAMS[idx] = torch.cuda.Stream(device=idx)
    return s


_LATE_PENDING = set()         # device indices whose late stream holds work the main stream has not waited for


def late_mark(device):
    """Something was put on the late stream of `device` that the main stream must wait for before it reads late parameters."""
    dev = torch.device(device)
    _LATE_PENDING.add(dev.index if dev.index is not None else torch.cuda.current_device())


def set_early_weights(params):
    """Declare the parameters whose copies the detector reads in the device-bound half of its step (an iterable of tensors;
    None: no split).  Everything else is late."""
    P = _WeightPlan
    with _CACHE_LOCK:
        P.early = None if params is None else {p.data_ptr() for p in params}
        P.dirty = bool(P.singles or P.groups)


def late_weights_ready():
    """The current stream waits for the late stream (stream mode; no-op when nothing is pending there)."""
    while _LATE_PENDING:
        idx = _LATE_PENDING.pop()
        torch.cuda.current_stream(idx).wait_stream(_LATE_STREAMS[idx])


_lib.on_switch.append(lambda: _WeightPlan.reset())      # the plan's arena and recorded weights live in ONE library's memory


def prepare_weights(training=True, split=False):
    """Start of a detector forward: convert the recorded convolution weights -- one launch -- and let the per-layer
    requests of THIS forward be answered from the arena (until `finish_weights()`).

    split: the caller promises `late_weights_ready()` before anything but the early weights' copies is read (set_early_weights):
    the late rows are then converted on the late stream.

    Training forwards always convert: a fused optimizer step changes the weights WITHOUT bumping their tensor version
    (checked: torch.optim.AdamW(fused=True) leaves `_version` at 0), so versions cannot tell.  Inference forwards
    convert after a training forward, when a version changed (load_state_dict, in-place edits), and the first time.
    No-op on a CPU library and in fp32 mode."""
    lib = _lib.get()
    P = _WeightPlan
    P.live = False
    if not lib.device_kernels or (PRECISION not in (1, 3) and HEAD_PRECISION not in (1, 3)) or not (P.singles or P.groups):
        return
    if len(P.singles) + len(P.groups) > 512:
        P.reset()
        return
    # (the prefetch worker records weights too -- engine.Builder._planned -> _planned_single inserts into P.singles while it
    # compiles the next batch's program: the scans and updates below run under the lock the inserts take)
    with _CACHE_LOCK:
        if P.dirty:
            dev = next(iter(P.singles.values()))[0].device if P.singles else next(iter(P.groups.values()))[0][0].device
            P._rebuild(dev)
            P.pending = True
        need = training or P.pending or any(e[2] != e[0]._version for e in P.singles.values()) or \
            any(g[1] != tuple(w._version for w in g[0]) for g in P.groups.values())
        if need:
            ne = P.n_early if (split and LATE_WEIGHTS and P.early is not None) else P.nrows
            if 0 < ne < P.nrows and LATE_MODE == "defer":
                lib.call("cg3d_spconv_prep_weights_bf16_table", ptr(P.table), c_int64(ne), lib.stream())
                tab, late_ptr, nlate = P.table, P.table.data_ptr() + ne * 6 * 8, P.nrows - ne        # (the closure keeps the table alive)
                defer(lambda: (tab, lib.call("cg3d_spconv_prep_weights_bf16_table", late_ptr, c_int64(nlate), lib.stream())))
            elif 0 < ne < P.nrows:
                lib.call("cg3d_spconv_prep_weights_bf16_table", ptr(P.table), c_int64(ne), lib.stream())
                ls = late_stream(P.table.device)
                ls.wait_stream(torch.cuda.current_stream())             # behind the optimizer's early rows / whatever wrote the weights here
                lib.call("cg3d_spconv_prep_weights_bf16_table", P.table.data_ptr() + ne * 6 * 8, c_int64(P.nrows - ne), ls.cuda_stream)
                late_mark(P.table.device)
            else:
                run_late()                             # (an optimizer step may have left the late parameters' update pending)
                lib.call("cg3d_spconv_prep_weights_bf16_table", ptr(P.table), c_int64(P.nrows), lib.stream())
            for e in P.singles.values():
                e[2] = e[0]._version
            for g in P.groups.values():
                g[1] = tuple(w._version for w in g[0])
        P.pending = bool(training)          # after a training forward the weights change behind our back
        P.live = True


def finish_weights():
    """End of the detector forward: from here on the arena may be stale (an optimizer step may follow)."""
    run_late()
    _WeightPlan.live = False


def _wkind(frag):
    """Key of a weight's prepared copies: bit 0 fragment order, bit 1 split operands (this thread's precision)."""
    return (1 if frag else 0) | (2 if _prec() == 3 else 0)


def _planned_single(w3, need_plain, frag=False):
    P = _WeightPlan
    _late_needed(w3)
    kind = _wkind(frag)
    e = P.singles.get((w3.data_ptr(), kind))
    if e is not None and e[0].shape == w3.shape and (e[1] or not need_plain):
        if P.live and e[3] is not None and e[2] == w3._version:
            return e
        return None
    if _lib.get().device_kernels and w3.dim() == 3:
        with _CACHE_LOCK:
            P.singles[(w3.data_ptr(), kind)] = [w3.detach(), need_plain or (e is not None and e[1]), -1, None, None]
            P.dirty = True
    return None


def _prep_bf16_t(w3):
    """fp32 [K, cin, cout] -> int16 view of bf16 [K, cout, cin] (cg3d_spconv_prep_weights_bf16)."""
    e = _planned_single(w3, False)
    if e is not None:
        return e[3]
    lib = _lib.get()
    K, cin, cout = w3.shape
    if _prec() == 3:
        out = torch.empty((K, cout, 3 * cin), dtype=torch.int16, device=w3.device)
        lib.call("cg3d_spconv_prep_weights_split", ptr(w3), ptr(None), ptr(out), ptr(None), c_int32(1), c_int64(K), c_int32(cin),
                 c_int32(cout), c_int32(0), lib.stream())
        return out
    out = torch.empty((K, cout, cin), dtype=torch.int16, device=w3.device)
    lib.call("cg3d_spconv_prep_weights_bf16", ptr(w3), ptr(out), c_int64(K), c_int32(cin), c_int32(cout), lib.stream())
    return out


def _prep_bf16_both(w3):
    """One launch: (bf16 [K, cout, cin] for the forward, bf16 [K, cin, cout] for the data gradient)."""
    e = _planned_single(w3, True)
    if e is not None:
        return e[3], e[4]
    lib = _lib.get()
    K, cin, cout = w3.shape
    x = _kx()
    wt = torch.empty((K, cout, cin * x), dtype=torch.int16, device=w3.device)
    wp = torch.empty((K, cin, cout * x), dtype=torch.int16, device=w3.device)
    if x == 3:
        lib.call("cg3d_spconv_prep_weights_split", ptr(w3), ptr(None), ptr(wt), ptr(wp), c_int32(1), c_int64(K), c_int32(cin),
                 c_int32(cout), c_int32(0), lib.stream())
    else:
        lib.call("cg3d_spconv_prep_weights_bf16_multi", ptr(w3), ptr(None), ptr(wt), ptr(wp), c_int32(1), c_int64(K),
                 c_int32(cin), c_int32(cout), lib.stream())
    return wt, wp


_wptr_cache = {}


def _prep_bf16_group(weights, transposed, frag=False):
    """Per-group weights (G tensors [K, cin, cout], never stacked in fp32) -> int16 view of the stacked bf16 buffer
    [G*K, cout, cin] (transposed) or [G*K, cin, cout]; frag: each [cin, cout] block in MFMA fragment order instead
    (the tile kernel's operand, cg3d_spconv_prep_weights_frag)."""
    lib = _lib.get()
    G, (K, cin, cout) = len(weights), weights[0].shape
    _late_needed(weights[0])
    key = tuple(w.data_ptr() for w in weights)
    kind = _wkind(frag)
    g = _WeightPlan.groups.get((key, transposed, kind))
    if g is not None:
        if _WeightPlan.live and g[2] is not None and g[1] == tuple(w._version for w in weights):
            return g[2]
    elif lib.device_kernels:
        with _CACHE_LOCK:
            _WeightPlan.groups[(key, transposed, kind)] = [[w.detach() for w in weights], None, None]
            _WeightPlan.dirty = True
    tab = _cached(_wptr_cache, key, lambda: h2d(list(key), torch.int64, weights[0].device) if lib.is_device
                  else torch.tensor(key, dtype=torch.int64), 64)
    x = _kx()
    out = torch.empty((G * K, cout, cin * x) if transposed else (G * K, cin, cout * x), dtype=torch.int16, device=weights[0].device)
    if x == 3:
        lib.call("cg3d_spconv_prep_weights_split", ptr(None), ptr(tab), ptr(out if transposed else None), ptr(None if transposed else out),
                 c_int32(G), c_int64(K), c_int32(cin), c_int32(cout), c_int32(1 if frag else 0), lib.stream())
        return out
    lib.call("cg3d_spconv_prep_weights_frag" if frag else "cg3d_spconv_prep_weights_bf16_multi", ptr(None), ptr(tab),
             ptr(out if transposed else None), ptr(None if transposed else out), c_int32(G), c_int64(K), c_int32(cin),
             c_int32(cout), lib.stream())
    return out


def _seg_len_fwd():
    return FWD_SEG if _lib.get().is_device else (1 << 30)


def _ksuffix():
    """Kind suffix of a profiled launch that runs split operands (three bf16 products per algorithmic product)."""
    return "x3" if _prec() == 3 else ""


class KernelProfile:
    """Optional live timing of the dominant kernel (k_spconv_pairs) with HIP events on the launch
    stream; bench.py turns it on for the timed region (roofline.achieved)."""
    enabled = False
    wgrad = False  # also time cg3d_spconv_pairs_wgrad (dev tool; the bench roofline is fwd/dgrad only)
    records = []   # (start_event, end_event, flops, SURVEY-8(d) bytes, meta, per-pair bytes)

    @classmethod
    def reset(cls):
        cls.records = []

    @classmethod
    def summary(cls):
        """Per kernel kind: launches, total ms, algorithmic flops and bytes."""
        out = {}
        for ev0, ev1, flops, nbytes, meta, pbytes in cls.records:
            d = out.setdefault(meta[0], {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "bytes_per_pair": 0.0})
            d["launches"] += 1
            d["ms"] += ev0.elapsed_time(ev1)
            d["flops"] += float(flops)
            d["bytes"] += float(nbytes)
            d["bytes_per_pair"] += float(pbytes)
        return out


def _conv_pairs(x, w3, pin, pout, seg, nseg, bias, n_out, n_pairs=0, w_bf16_t=None):
    """Y[pout] += X[pin] @ w3[slot].  `w_bf16_t` (optional): the weights already as bf16 [slots, cout, cin]."""
    lib = _lib.get()
    if isinstance(w3, tuple):                 # (slots, cin, cout): the weights only exist as the prepared bf16 buffer
        K, cin, cout = w3
        w3 = None
    else:
        K, cin, cout = w3.shape
    lib.check(x, w3, pin, pout, seg, bias, w_bf16_t)
    # Y is initialised inside the C call (accumulate = 0: a memset, or a bias broadcast kernel, on the same stream): one
    # host-side op fewer per launch than a torch fill -- 34 of them per step
    y = torch.empty((n_out, cout), dtype=torch.float32, device=x.device)
    prec = 1 if _use_bf16(cin) else 0
    if prec and _split() and x.dtype != torch.int16:
        x = _to_split(x)                      # (no rounding-on-the-fly form of the split: the rows are always expanded)
    wptr = w3
    kc = cin
    if prec == 1:
        kc = cin * _kx()
        if w_bf16_t is None:                  # converted on the spot (temporaries: not recorded in the step's weight plan)
            w_bf16_t = torch.empty((K, cout, kc), dtype=torch.int16, device=x.device)
            if _split():
                lib.call("cg3d_spconv_prep_weights_split", ptr(w3), ptr(None), ptr(w_bf16_t), ptr(None), c_int32(1), c_int64(K),
                         c_int32(cin), c_int32(cout), c_int32(0), lib.stream())
            else:
                lib.call("cg3d_spconv_prep_weights_bf16", ptr(w3), ptr(w_bf16_t), c_int64(K), c_int32(cin), c_int32(cout),
                         lib.stream())
        wptr = w_bf16_t
    prof = KernelProfile.enabled and lib.is_device
    if prof:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    rows16 = x.dtype == torch.int16
    lib.call("cg3d_spconv_pairs_fwd", ptr(x), ptr(wptr), ptr(pin), ptr(pout), ptr(seg), c_int64(nseg), ptr(bias), ptr(y),
             c_int64(n_out), c_int32(kc), c_int32(cout), c_int32(2 if rows16 else prec), c_int32(0), lib.stream())
    if prof:
        ev1.record()
        # algorithmic work of one launch: 2*P*cin*cout flops; bytes = every gathered input row and every
        # scattered output row once (the PMC passes in profiles/ count the atomic payload once as well)
        # + the weights once + the two pair lists
        wb = 2.0 if prec == 1 else 4.0
        xb = 2.0 if rows16 else 4.0
        KernelProfile.records.append((ev0, ev1, 2.0 * n_pairs * cin * cout,
                                      xb * x.shape[0] * cin + 4.0 * n_out * cout + wb * K * cin * cout + 8.0 * n_pairs,
                                      (("pairs_bf16" + _ksuffix()) if prec else "pairs", K, cin, cout, n_pairs, n_out, nseg),
                                      n_pairs * (xb * cin + 4.0 * cout) + wb * K * cin * cout + 8.0 * n_pairs))
    return y


_WGRAD_BF16_WGS = int(__import__("os").environ.get("CG3D_WGRAD_BF16_WGS", "2048"))
_WGRAD_BF16_MIN = int(__import__("os").environ.get("CG3D_WGRAD_BF16_MIN", "256"))


def _wgrad_seg_len(P, cin, cout, precision=0, K=27):
    if not _lib.get().is_device:
        return 1 << 30
    if precision == 1:
        tiles = (-(-cin // (128 if cin > 64 else 64))) * (-(-cout // (128 if cout > 64 else 64)))
        # The kernel runs 2 workgroups per CU = 512 slots on MI355X; a launch of 1.3 x 512 workgroups costs two full
        # rounds, so the segment length is chosen to land just under a whole number of rounds (every offset adds one
        # short tail segment).  Measured optimum: ~4 rounds when several channel tiles share the gathered rows
        # through L2, 1-2 rounds (fewer 64 KB atomic epilogues) when a single tile covers the layer.
        # Round 5, the 8-wave 128 x 128 kernel (profiles/r05_wgrad_segments.txt): ONE round also when several channel tiles
        # share the rows -- 256 -> 256 @ 23 015 rows 136.6 -> 99.6 us, 128 -> 256 @ 82 107 rows 160.7 -> 126.6 us, 512 -> 512 @ 5 330 rows
        # 132.1 -> 114.3 us at 512 instead of 2 048 workgroups: every extra workgroup of a multi-tile layer repeats the
        # gather of its pairs' rows and ends with another 64 KB of atomics
        rounds = 1 if tiles > 1 else (2 if max(cin, cout) <= 64 else 1)
        slots = max(512 * rounds * _WGRAD_BF16_WGS // 2048 - K * tiles, 64)
        per = -(-P * tiles // slots)
        return max(_WGRAD_BF16_MIN, -(-per // 64) * 64)
    if min(cin, cout) <= 32 <= max(cin, cout):
        # k_spconv_pairs_wgrad_skinny: every workgroup ends with (wide x narrow) atomics onto the same few addresses --
        # few, long segments worked by 16 waves when the narrow side is large, many short ones when it is 1-8 channels
        wgs = WGRAD_SKINNY_WGS if min(cin, cout) > 8 else 1024
        return max(256, -(-(-(-P // wgs)) // 64) * 64)
    t = 128 if (cin >= 128 and cout >= 128) else 64          # tile edge of the kernel the library will pick
    tiles = ((cin + t - 1) // t) * ((cout + t - 1) // t)
    per = -(-P * tiles // 2048)             # aim at >= 2048 workgroups
    return max(256, -(-per // 256) * 256)


WGRAD_SKINNY_WGS = int(__import__("os").environ.get("CG3D_WGRAD_SKINNY_WGS", "64"))      # measured at cout = 18: 58 / 66 / 88 us at 64 / 128 / 256
TILE_MIN_ROWS = int(__import__("os").environ.get("CG3D_TILE_MIN_ROWS", "4096"))
# largest kernel volume of an ungrouped map on the tile kernel (the plan builder cuts K > 32 into runs of offsets, one wave each;
# the RoI head's 5^3 convolution at given coordinates, cagroup_roi_head.py:69, is the K = 125 case)
TILE_MAX_K = int(__import__("os").environ.get("CG3D_TILE_MAX_K", "32"))
LINEAR_KERNEL = __import__("os").environ.get("CG3D_LINEAR_KERNEL", "1") != "0"
LINEAR_WGRAD_SMALL = __import__("os").environ.get("CG3D_LINEAR_WGRAD_SMALL", "1") != "0"   # 1 k - 8 k rows: bf16-rows pair kernel, not the library
TILE_KERNEL = __import__("os").environ.get("CG3D_TILE_KERNEL", "1") != "0"
GROUP_TILE_KERNEL = __import__("os").environ.get("CG3D_GROUP_TILE_KERNEL", "1") != "0"


def _use_tile(kmap, K, cin, cout, n_rows, row_bounds):
    """The LDS-staged tile kernel (cg3d_spconv_tile_fwd): bf16 mode with bf16 row copies, channel counts its register tile
    covers (64-channel chunks in, 64 or multiples of 128 out), enough rows to give every CU a tile.  Since the rows reach
    LDS by LDS-DMA it beats the dense-map kernel on every S50k layer shape -- same-map, strided (2-3 passes per tile) and
    transposed maps, 64 channels included (`profiles/r02_tile_vs_dense_map.txt`)."""
    return (TILE_KERNEL and _lib.get().device_kernels and _prec() in (1, 3) and BF16_ROWS and row_bounds is None
            and 1 < K <= TILE_MAX_K and cin % 64 == 0 and cout % 64 == 0 and n_rows >= TILE_MIN_ROWS)


def _wgrad_prec(cin, cout, have_rows16):
    """Precision argument of cg3d_spconv_pairs_wgrad for a layer of this thread's precision: 0 fp32 operands, 1 bf16 operands
    rounded on the fly from fp32 rows, 2 rows stored as bf16, 3 split rows (three accumulating bf16 passes).  The split
    precision never rounds an operand to a single bf16: without 8-channel multiples on both sides it is the fp32 kernel."""
    wprec = 1 if (_use_bf16(cin) and cout % 4 == 0 and cout >= 16) else 0
    if _prec() == 3:
        return 3 if (wprec and have_rows16 and cout % 8 == 0) else 0
    if wprec and have_rows16 and cout % 8 == 0:
        return 2
    return wprec


class SparseConvFunction(torch.autograd.Function):
    """Y = conv(X, W) on a kernel map; gather -> MFMA -> atomic scatter over the pair lists.

    With `row_bounds` (host tuple of G+1 output-row boundaries) the output rows form G groups with
    their own weights: weight is [G*K, cin, cout] and group g uses weight[g*K + k]."""

    @staticmethod
    def _implicit(kmap, P, cin, cout, n_rows, row_bounds):
        """Output-stationary kernel (no atomics) or pair form?  bf16 mode, ungrouped maps with >= 10 % neighbourhood
        occupancy and enough output tiles (128 rows x 128 channels each) to give the chip ~100 workgroups."""
        return (row_bounds is None and _use_bf16(cin) and kmap.K > 1
                and P >= IMPLICIT_MIN_OCCUPANCY * kmap.K * max(min(kmap.n_out, kmap.n_in), 1)
                and -(-n_rows // 128) * -(-cout // 128) >= IMPLICIT_MIN_TILES)

    @staticmethod
    def warm(kmap, K, cin, cout, row_bounds=None, backward=True):
        """Build (and cache on the map) everything forward / backward of this layer will read from the host."""
        _, _, _, P = kmap.pairs(row_bounds)
        if _use_tile(kmap, K, cin, cout, kmap.n_out, row_bounds):
            kmap.tile_plan(False)
        elif not SparseConvFunction._implicit(kmap, P, cin, cout, kmap.n_out, row_bounds):
            kmap.segments(_seg_len_fwd(), row_bounds)
        if backward:
            if _use_tile(kmap, K, cout, cin, kmap.n_in, row_bounds):
                kmap.tile_plan(True)
            elif SparseConvFunction._implicit(kmap, P, cout, cin, kmap.n_in, row_bounds):
                _ = kmap.nbrT
            else:
                kmap.segments(_seg_len_fwd(), row_bounds)
            wprec = _wgrad_prec(cin, cout, True)
            (kmap.wgrad_segments if wprec else kmap.segments)(_wgrad_seg_len(P, cin, cout, 1 if wprec else 0, K), row_bounds)

    @staticmethod
    def forward(ctx, x, weight, bias, kmap, row_bounds=None):
        ctx.prec = _prec()
        x = x.contiguous()
        w3 = weight.contiguous()
        ctx.kmap, ctx.has_bias, ctx.row_bounds = kmap, bias is not None, row_bounds
        pin, pout, _, P = kmap.pairs(row_bounds)
        b = bias.contiguous() if bias is not None else None
        cin, cout = w3.shape[1], w3.shape[2]
        # bf16 mode: one streaming conversion of the input rows, then every gather of this layer (forward and
        # weight gradient) moves half the bytes
        xg = _to_bf16(x, keep=True) if (BF16_ROWS and _use_bf16(cin)) else x
        wt = wp = None
        K = w3.shape[0]
        tile_f = _use_tile(kmap, K, cin, cout, kmap.n_out, row_bounds)
        tile_b = ctx.needs_input_grad[0] and _use_tile(kmap, K, cout, cin, kmap.n_in, row_bounds)
        ctx.tile_b = tile_b
        if tile_f or tile_b:
            # fragment-ordered copies (one launch; from the step's arena when recorded): forward and data-gradient operand
            wt, wp = _prep_frag(w3, tile_f, tile_b)
            if not tile_f:
                wt = _prep_bf16_t(w3)
            ctx.save_for_backward(x, w3, xg if xg is not x else None, wp)
            if tile_f:
                return _conv_tile(xg, wt, kmap.tile_plan(False), b, cin, cout, kmap.n_in, P, want_stats=True)
        elif _use_bf16(cin):
            # both bf16 copies of the weights in one launch; the plain one is the data gradient's operand
            wt, wp = _prep_bf16_both(w3) if _use_bf16(cout) else (_prep_bf16_t(w3), None)
            ctx.save_for_backward(x, w3, xg if xg is not x else None, wp)
        else:
            ctx.save_for_backward(x, w3, xg if xg is not x else None, wp)
        if SparseConvFunction._implicit(kmap, P, cin, cout, kmap.n_out, row_bounds):
            return _conv_implicit_bf16(xg, wt, kmap.nbr, b, kmap.n_out, cin, cout, P)
        seg, nseg = kmap.segments(_seg_len_fwd(), row_bounds)
        return _conv_pairs(xg, w3, pin, pout, seg, nseg, b, kmap.n_out, P, w_bf16_t=wt)

    @staticmethod
    @_ctx_precision
    def backward(ctx, dy):
        x, w3, xb, wp = ctx.saved_tensors
        kmap, rb = ctx.kmap, ctx.row_bounds
        dy = dy.contiguous()
        lib = _lib.get()
        pin, pout, _, P = kmap.pairs(rb)
        KK, cin, cout = w3.shape
        dx = dw = db = None
        dyg = _to_bf16(dy) if (BF16_ROWS and _use_bf16(cout)) else dy     # shared by dgrad and wgrad
        if ctx.needs_input_grad[0]:
            if getattr(ctx, "tile_b", False):
                # the swapped problem on the plan of the transposed map; operand = the plain fragment-ordered copy
                dx = _conv_tile(dyg, wp, kmap.tile_plan(True), None, cout, cin, kmap.n_out, P, wrev=kmap.symmetric)
            elif SparseConvFunction._implicit(kmap, P, cout, cin, kmap.n_in, rb):
                # the swapped problem's bf16 [K, cout'=cin, cin'=cout] weights are W itself, cast
                dx = _conv_implicit_bf16(dyg, wp if wp is not None else _prep_bf16_both(w3)[1],
                                         kmap.nbrT, None, kmap.n_in, cout, cin, P)
            else:
                seg, nseg = kmap.segments(_seg_len_fwd(), rb)
                if _use_bf16(cout):
                    dx = _conv_pairs(dyg, (KK, cout, cin), pout, pin, seg, nseg, None, kmap.n_in, P,
                                     w_bf16_t=wp if wp is not None else _prep_bf16_both(w3)[1])
                else:
                    wt = w3.transpose(1, 2).contiguous()
                    dx = _conv_pairs(dy, wt, pout, pin, seg, nseg, None, kmap.n_in, P)   # lists swapped
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w3)
            wprec = _wgrad_prec(cin, cout, xb is not None)
            xw, dyw = x, dy
            if wprec >= 2:
                xw, dyw = xb, (dyg if dyg is not dy else _to_bf16(dy))
            seg, nseg = (kmap.wgrad_segments if wprec else kmap.segments)(_wgrad_seg_len(P, cin, cout, 1 if wprec else 0, KK), rb)
            lib.check(xw, dyw, pin, pout, seg)
            prof = KernelProfile.enabled and KernelProfile.wgrad and lib.is_device
            if prof:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
            lib.call("cg3d_spconv_pairs_wgrad", ptr(xw), ptr(dyw), ptr(pin), ptr(pout), ptr(seg), c_int64(nseg), ptr(dw),
                     c_int32(KK), c_int32(cin), c_int32(cout), c_int32(wprec), lib.stream())
            if prof:
                ev1.record()
                eb = 2.0 if wprec >= 2 else 4.0
                KernelProfile.records.append((ev0, ev1, 2.0 * P * cin * cout,
                                              eb * (kmap.n_in * cin + kmap.n_out * cout) + 4.0 * KK * cin * cout + 8.0 * P,
                                              ("wgrad_bf16x3" if wprec == 3 else "wgrad_bf16" if wprec == 2 else ("wgrad_bf16_fp32rows" if wprec else "wgrad"),
                                               KK, cin, cout, P, kmap.n_out, nseg),
                                              eb * P * (cin + cout) + 4.0 * KK * cin * cout))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None, None


class GroupedConvFunction(torch.autograd.Function):
    """SparseConvFunction for G row groups whose weights are G separate parameters (the class branches,
    cagroup_head.py:183-188) in the bf16 mode: the per-class weights go straight into the stacked bf16 operand
    buffers (no 215 MB fp32 stack per step for the 9^3 branch), the weight gradient is computed into one stacked
    buffer and handed back as per-class views.  Maps from a coordinate map onto itself with >= 10 % neighbourhood
    occupancy (the 9^3 and 5^3 class convolutions) run the output-stationary kernel on group-aligned row tiles."""

    @staticmethod
    def _tiled(kmap, P, K, closed):
        return closed and kmap.same_map and K > 1 and P >= IMPLICIT_MIN_OCCUPANCY * K * max(kmap.n_out, 1)

    @staticmethod
    def _lds_tile(kmap, K, cin, cout, closed):
        """The LDS-staged tile kernel on group-aligned tiles: closed same-map groups (forward and data gradient share the
        row groups), bf16 row copies, channel counts the kernel's register tile covers.  The rows of a pass are staged once
        for all of its slot-table blocks, so the 5^3 / 9^3 class convolutions (K = 125 / 729) gather each distinct
        neighbour row once per pass instead of once per offset."""
        return (GROUP_TILE_KERNEL and closed and kmap.same_map and K > 1 and _lib.get().device_kernels and _prec() in (1, 3) and BF16_ROWS
                and cin % 64 == 0 and cout % 64 == 0 and (cin == 64 or cin % 128 == 0) and (cout == 64 or cout % 128 == 0))

    @staticmethod
    def _ksplit(plan, K):
        """Offset shares per tile (atomics into a zeroed output) when the tiles alone cannot fill the chip."""
        ncu = 512 if (K > 64 and plan.ucap <= 255) else 256        # (workgroup slots: two per CU for the 128-offset-block kernel on 255-row passes)
        return max(1, min(8, K // 16, ncu // max(plan.ntile, 1)))

    @staticmethod
    def forward(ctx, x, kmap, row_bounds, closed, *weights):
        ctx.prec = _prec()
        x = x.contiguous()
        G, (K, cin, cout) = len(weights), weights[0].shape
        ctx.kmap, ctx.row_bounds, ctx.shape, ctx.closed = kmap, row_bounds, (G, K, cin, cout), closed
        pin, pout, _, P = kmap.pairs(row_bounds)
        xg = _to_bf16(x, keep=True) if (BF16_ROWS or _split()) else x
        ctx.save_for_backward(x, xg if xg is not x else None, *weights)
        lds_tile = ctx.lds_tile = GroupedConvFunction._lds_tile(kmap, K, cin, cout, closed)
        wt = _prep_bf16_group(weights, True, lds_tile)
        # the data gradient's operand: free while the step's arena is live (backward runs after the forward closed it)
        ctx.wp_plain = _prep_bf16_group(weights, False, lds_tile) if (_WeightPlan.live and ctx.needs_input_grad[0]) else None
        if lds_tile:
            plan = kmap.tile_plan(False, row_bounds)
            return _conv_tile(xg, wt, plan, None, cin, cout, kmap.n_in, P, GroupedConvFunction._ksplit(plan, K), G)
        if GroupedConvFunction._tiled(kmap, P, K, closed):
            return _conv_implicit_bf16(xg, wt, kmap.nbr, None, kmap.n_out, cin, cout, P, kmap.tiles(row_bounds))
        seg, nseg = kmap.segments(_seg_len_fwd(), row_bounds)
        return _conv_pairs(xg, (G * K, cin, cout), pin, pout, seg, nseg, None, kmap.n_out, P, w_bf16_t=wt)

    @staticmethod
    @_ctx_precision
    def backward(ctx, dy):
        x, xb = ctx.saved_tensors[:2]
        weights = ctx.saved_tensors[2:]
        kmap, rb = ctx.kmap, ctx.row_bounds
        G, K, cin, cout = ctx.shape
        lib = _lib.get()
        dy = dy.contiguous()
        pin, pout, _, P = kmap.pairs(rb)
        dyg = _to_bf16(dy) if (BF16_ROWS or _split()) else dy
        dx = None
        if ctx.needs_input_grad[0]:
            wp = ctx.wp_plain if ctx.wp_plain is not None else _prep_bf16_group(weights, False, ctx.lds_tile)
            if ctx.lds_tile:
                plan = kmap.tile_plan(True, rb)
                dx = _conv_tile(dyg, wp, plan, None, cout, cin, kmap.n_out, P, GroupedConvFunction._ksplit(plan, K), G,
                                wrev=kmap.symmetric)
            elif GroupedConvFunction._tiled(kmap, P, K, ctx.closed):
                dx = _conv_implicit_bf16(dyg, wp, kmap.nbrT, None, kmap.n_in, cout, cin, P, kmap.tiles(rb))
            else:
                seg, nseg = kmap.segments(_seg_len_fwd(), rb)
                dx = _conv_pairs(dyg, (G * K, cout, cin), pout, pin, seg, nseg, None, kmap.n_in, P, w_bf16_t=wp)
        dws = [None] * G
        if any(ctx.needs_input_grad[4:]):
            dw = torch.empty((G * K, cin, cout), dtype=torch.float32, device=x.device)
            wprec = (3 if _split() else 2) if (xb is not None and dyg is not dy) else 1
            xw, dyw = (xb, dyg) if wprec >= 2 else (x, dy)
            seg, nseg = kmap.segments(_wgrad_seg_len(P, cin, cout, 1, G * K), rb)
            lib.check(xw, dyw, pin, pout, seg)
            lib.call("cg3d_spconv_pairs_wgrad", ptr(xw), ptr(dyw), ptr(pin), ptr(pout), ptr(seg), c_int64(nseg), ptr(dw),
                     c_int32(G * K), c_int32(cin), c_int32(cout), c_int32(wprec), lib.stream())
            dws = [dw[g * K:(g + 1) * K] for g in range(G)]
        return (dx, None, None, None) + tuple(dws)


def grouped_conv(x, weights, kmap, row_bounds, closed=False):
    """Convolution of G contiguous row groups with their own weights ([K, cin, cout] each); a pair (in, out) uses the
    weights of the OUTPUT row's group.  closed=True promises that no pair crosses a group boundary (the class branches:
    every group has its own batch index) -- then the data gradient may be tiled by input-row groups as well and the
    atomic-free output-stationary kernel is used on dense enough maps."""
    weights = [w.view(-1, w.shape[-2], w.shape[-1]) for w in weights]
    cin, cout = weights[0].shape[1], weights[0].shape[2]
    if _use_bf16(cin) and _use_bf16(cout) and _lib.get().device_kernels:
        return GroupedConvFunction.apply(x, kmap, row_bounds, bool(closed), *weights)
    w = torch.stack(weights, dim=0)
    return SparseConvFunction.apply(x, w.view(-1, cin, cout), None, kmap, row_bounds)


_ident_cache = {}


def _identity_pairs(n, seglen, device):
    """arange pair list + (0, start, count) segment table for an n-row dense weight gradient (cached)."""
    ck = (n, seglen, str(device))

    def build():
        ar = None
        for (n2, _, d2), v in _ident_cache.items():
            if n2 == n and d2 == str(device):
                ar = v[0]
        if ar is None:
            ar = torch.arange(n, dtype=torch.int32, device=device)
        starts = np.arange(0, n, seglen, dtype=np.int64)
        tab = np.stack([np.zeros_like(starts), starts, np.minimum(seglen, n - starts)], 1).astype(np.int32)
        return (ar, h2d(torch.from_numpy(tab), torch.int32, device), int(tab.shape[0]))
    return _cached(_ident_cache, ck, build, 256)


class LinearFunction(torch.autograd.Function):
    """y = x @ w (+ bias) for the 1x1x1 convolutions.  Forward and data gradient are plain library GEMMs;
    the weight gradient x^T @ dy has a tiny output and a contraction over every row of the sparse tensor, which
    the library runs on a handful of workgroups -- it goes through the split-over-rows wgrad kernel instead
    (cg3d_spconv_pairs_wgrad on the identity pair list)."""
    MIN_ROWS = 8192
    OWN_MIN_ROWS = int(__import__("os").environ.get("CG3D_LINEAR_MIN_ROWS", "1"))      # (1 024 until late in round 3: below it the library GEMM's host cost, 25-60 us per call, is the larger part)

    @staticmethod
    def _skinny(n, a, b):
        """Many rows x (few channels on one side): the library picks 16 x 256 tiles for these (0.25 ms for a
        155 k x 64 x 3 product); the identity-map pair kernel streams the rows once instead."""
        return n >= LinearFunction.MIN_ROWS and min(a, b) < 32 and _lib.get().device_kernels

    @staticmethod
    def _rows_gemm(x, w, bias):
        n, (cin, cout) = x.shape[0], w.shape
        ar, seg, nseg = _identity_pairs(n, 128, x.device)
        if _split():
            # a narrow side: the fp32 pair kernel is cheap here (and exact) -- no split copy of 150 k rows for 3 output channels
            with precision_scope(0):
                return _conv_pairs(x.contiguous(), w.contiguous().view(1, cin, cout), ar, ar, seg, nseg, bias, n, n)
        return _conv_pairs(x.contiguous(), w.contiguous().view(1, cin, cout), ar, ar, seg, nseg, bias, n, n)

    @staticmethod
    def _own(n, cin, cout):
        """The hand-written streaming kernel (cg3d_linear_fwd, csrc/linear.hip): bench precision, bf16 row copies, channel
        counts in multiples of 64 on both sides (the data gradient is the same kernel with the roles swapped)."""
        return (LINEAR_KERNEL and _prec() in (1, 3) and BF16_ROWS and n >= LinearFunction.OWN_MIN_ROWS
                and cin % 64 == 0 and cout % 64 == 0 and cin >= 64 and cout >= 64)

    @staticmethod
    def _own_gemm(x16, wf, bias, n, cin, cout, want_stats=False):
        lib = _lib.get()
        y = torch.empty((n, cout), dtype=torch.float32, device=x16.device)
        cin = cin * _kx()                      # the contraction the kernel sees (split rows: [hi | lo | hi])
        # few rows x a long contraction (DAPPM: 32-284 rows x 1024 channels): one workgroup would walk 16 chunks one latency
        # at a time -- the chunks go to separate workgroups instead, partial products stored and summed by the same call
        units, nchunk = -(-n // 128) * (cout // (128 if cout % 128 == 0 else 64)), cin // 64
        ksplit = min(nchunk, 64 // max(units, 1)) if (units <= 16 and nchunk >= 4) else 1
        stats = part = None
        if ksplit > 1:
            part = torch.empty((ksplit, n, cout), dtype=torch.float32, device=x16.device)
        elif want_stats and WANT_BN_STATS and FUSED_BN_STATS and cout <= 1024:
            stats = zero_arena().take(BN_SLOTS * 2 * cout, x16.device)
            if len(_STATS) > 64:
                _STATS.clear()
            _STATS[y.data_ptr()] = (stats, 1, n, cout, y)
        lib.check(x16, wf, bias)
        lib.call("cg3d_linear_fwd", ptr(x16), ptr(wf), ptr(bias), ptr(y), c_int64(n), c_int32(cin), c_int32(cout), c_int32(max(ksplit, 1)),
                 ptr(stats), ptr(part), lib.stream())
        return y

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.prec = _prec()
        ctx.has_bias = bias is not None
        n, (cin, cout) = x.shape[0], w.shape
        ctx.own = LinearFunction._own(n, cin, cout)
        if ctx.own:
            x = x.contiguous()
            x16 = _to_bf16(x, keep=True)
            wt, wp = _prep_frag(w.contiguous().view(1, cin, cout), True, ctx.needs_input_grad[0])
            ctx.save_for_backward(x, w, x16, wp)
            return LinearFunction._own_gemm(x16, wt, bias.contiguous() if bias is not None else None, n, cin, cout, want_stats=True)
        ctx.save_for_backward(x, w, None, None)
        if LinearFunction._skinny(x.shape[0], w.shape[0], w.shape[1]):
            return LinearFunction._rows_gemm(x, w, bias.contiguous() if bias is not None else None)
        return torch.addmm(bias, x, w) if bias is not None else x @ w

    @staticmethod
    @_ctx_precision
    def backward(ctx, dy):
        x, w, x16, wp = ctx.saved_tensors
        dx = dw = db = None
        dy16 = None
        if ctx.own:
            dy = dy.contiguous()
            dy16 = _to_bf16(dy)
        if ctx.needs_input_grad[0]:
            if ctx.own:
                if wp is None:
                    _, wp = _prep_frag(w.contiguous().view(1, w.shape[0], w.shape[1]), False, True)
                dx = LinearFunction._own_gemm(dy16, wp, None, x.shape[0], w.shape[1], w.shape[0])
            elif LinearFunction._skinny(x.shape[0], w.shape[0], w.shape[1]):
                dx = LinearFunction._rows_gemm(dy, w.t(), None)
            else:
                dx = dy @ w.t()
        if ctx.needs_input_grad[1]:
            n, (cin, cout) = x.shape[0], w.shape
            lib = _lib.get()
            if (n < LinearFunction.MIN_ROWS and not (ctx.own and LINEAR_WGRAD_SMALL)) or not lib.device_kernels:
                dw = x.t() @ dy
            else:
                xc, dyc = x.contiguous(), dy.contiguous()
                wprec = _wgrad_prec(cin, cout, x16 is not None and dy16 is not None)
                if wprec >= 2:
                    xc, dyc = x16, dy16                     # both operands as bf16 rows: half the gather traffic
                ar, seg, nseg = _identity_pairs(n, _wgrad_seg_len(n, cin, cout, 1 if wprec else 0, 1), x.device)
                dw = torch.empty_like(w)
                lib.check(xc, dyc, ar, seg, dw)
                lib.call("cg3d_spconv_pairs_wgrad", ptr(xc), ptr(dyc), ptr(ar), ptr(ar), ptr(seg), c_int64(nseg), ptr(dw),
                         c_int32(1), c_int32(cin), c_int32(cout), c_int32(wprec), lib.stream())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def linear(x, w, bias=None):
    return LinearFunction.apply(x, w, bias)


class LinearTFunction(torch.autograd.Function):
    """y = x @ w^T (+ bias) with w stored [cout, cin] like nn.Linear (the RoI head's FC layers, reference
    roi_heads/cagroup_roi_head.py:37-55) on the same kernel as `linear`: for the [cout, cin] tensor the PLAIN fragment copy is
    the forward operand (contraction over its second index) and the transposed copy the data gradient's; the weight
    gradient dy^T x is the pair kernel with the operands swapped.  The library GEMM costs the host 25-70 us per call."""

    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.prec = _prec()
        n, (cout, cin) = x.shape[0], w.shape
        x = x.contiguous()
        x16 = _to_bf16(x, keep=True)
        wt, wp = _prep_frag(w.contiguous().view(1, cout, cin), ctx.needs_input_grad[0], True)
        ctx.save_for_backward(x16, w, wt)
        ctx.has_bias = bias is not None
        return LinearFunction._own_gemm(x16, wp, bias.contiguous() if bias is not None else None, n, cin, cout, want_stats=True)

    @staticmethod
    @_ctx_precision
    def backward(ctx, dy):
        x16, w, wt = ctx.saved_tensors
        n, (cout, cin) = dy.shape[0], w.shape
        dy = dy.contiguous()
        dy16 = _to_bf16(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if wt is None:
                wt, _ = _prep_frag(w.contiguous().view(1, cout, cin), True, False)
            dx = LinearFunction._own_gemm(dy16, wt, None, n, cout, cin)
        if ctx.needs_input_grad[1]:
            lib = _lib.get()
            ar, seg, nseg = _identity_pairs(n, _wgrad_seg_len(n, cout, cin, 1, 1), dy.device)
            dw = torch.empty_like(w)
            lib.check(x16, dy16, ar, seg, dw)
            lib.call("cg3d_spconv_pairs_wgrad", ptr(dy16), ptr(x16), ptr(ar), ptr(ar), ptr(seg), c_int64(nseg), ptr(dw),
                     c_int32(1), c_int32(cout), c_int32(cin), c_int32(3 if _split() else 2), lib.stream())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db


def linear_t(x, w, bias=None):
    """nn.Linear's y = x @ w^T + bias (w [cout, cin]); the own kernel in the bench precision with 64-multiple channels."""
    if x.dim() == 2 and LinearFunction._own(x.shape[0], w.shape[1], w.shape[0]):
        return LinearTFunction.apply(x, w, bias)
    return torch.nn.functional.linear(x, w, bias)


def _sorted_counts(b, n_batch):
    """Host list [violations, count_0 .. count_{n_batch-1}] of an int32 / int64 id vector (or strided id column) in ONE launch
    and one host read (cg3d_count_sorted_ids), or None when the column is not of that kind / not on the bound library."""
    lib = _lib.get()
    if b.dim() != 1 or b.dtype not in (torch.int32, torch.int64) or b.is_cuda != lib.is_device or n_batch > 8192 or n_batch < 1:
        return None
    counts = torch.empty(n_batch + 1, dtype=torch.int64, device=b.device)
    lib.call("cg3d_count_sorted_ids", ptr(b), c_int64(b.shape[0]), c_int32(b.stride(0)), c_int32(1 if b.dtype == torch.int64 else 0),
             c_int32(n_batch), ptr(counts), lib.stream())
    h = counts.tolist()
    return [h[-1]] + h[:-1]


def sorted_batch_counts(b, n_batch):
    """Rows per batch index (host list of n_batch ints) when the batch-index column `b` is non-decreasing with values in
    [0, n_batch), else None; one host read."""
    b = b.reshape(-1) if b.dim() != 1 else b
    if b.numel() == 0:
        return [0] * n_batch
    if b.dtype.is_floating_point:
        b = b.long()
    host = _sorted_counts(b, n_batch)
    if host is None:
        bad = ((b[1:] < b[:-1]).sum() + (b[-1] >= n_batch) + (b[0] < 0)).view(1).long()
        host = torch.cat([bad, count_ids(b.long(), n_batch)]).tolist()
    return host[1:] if host[0] == 0 else None


def rows_by_batch(b, n_batch=None, info=None):
    """Per-batch row index lists (ascending) of a batch-index column: torch.split(stable argsort, counts) with ONE host
    read.  Rows of every map of this build are batch-major ((batch, Morton) order, strided maps in first-occurrence order
    of their parents), so the sort is normally the identity: the number of descents rides along with the counts and the
    ~10 merge-sort launches are only paid when it is not zero.  info (a dict): info["sorted"] = the column was batch-major,
    i.e. the lists are consecutive row ranges."""
    b = b.reshape(-1) if b.dim() != 1 else b
    if b.numel() == 0:
        return []
    if not b.is_cuda:
        b = b.long()
        order = torch.sort(b, stable=True)[1]
        return list(torch.split(order, torch.bincount(b, minlength=n_batch or 0).tolist()))
    # (batch size not given: 1 024 bins, the trailing empty ones dropped; a larger index shows up as a violation)
    host = _sorted_counts(b if b.dtype in (torch.int32, torch.int64) else b.long(), n_batch if n_batch is not None else 1024)
    if host is not None and host[0] == 0:
        counts = host[1:]
        if n_batch is None:
            while len(counts) > 1 and counts[-1] == 0:
                counts.pop()
        if info is not None:
            info["sorted"] = True
        return list(torch.split(torch.arange(b.numel(), device=b.device), counts))
    b = b.long()
    # descents, plus (known batch size) ids outside [0, n_batch): `count_ids` would drop those silently and torch.split
    # would then fail with an opaque size error -- they take the sort + bincount path below, which extends the list
    desc = (b[1:] < b[:-1]).sum().view(1)
    if n_batch is not None:
        desc = desc + (b[-1] >= n_batch) + (b[0] < 0)
    if n_batch is None:
        host = torch.cat([desc, b[-1:]]).tolist()          # batch-major: the last row holds the largest index
        if host[0] == 0:
            counts = count_ids(b, host[1] + 1).tolist()    # (second read only on this path without a known batch size)
            if info is not None:
                info["sorted"] = True
            return list(torch.split(torch.arange(b.numel(), device=b.device), counts))
    else:
        host = torch.cat([desc, count_ids(b, n_batch)]).tolist()
        if host[0] == 0:
            if info is not None:
                info["sorted"] = True
            return list(torch.split(torch.arange(b.numel(), device=b.device), host[1:]))
    if int(b.min()) < 0:
        raise ValueError("rows_by_batch: negative batch index")
    order = torch.sort(b, stable=True)[1]
    return list(torch.split(order, torch.bincount(b, minlength=n_batch or 0).tolist()))


def count_ids(ids, m):
    """torch.bincount(ids, minlength=m) for ids known to lie in [0, m) (others are ignored): one histogram launch
    (cg3d_count_ids) on an int32 / int64 id vector or strided id column -- bincount first scans its input for min and max (two
    single-workgroup-chain reductions, 14 + 23 us on 150 k ids)."""
    n = ids.numel()
    if n == 0:
        return torch.zeros(m, dtype=torch.int64, device=ids.device)
    lib = _lib.get()
    if m > 8192 or ids.dim() != 1 or ids.dtype not in (torch.int32, torch.int64) or ids.is_cuda != lib.is_device:
        return torch.bincount(ids.reshape(-1).long(), minlength=m)[:m]
    counts = torch.empty(m, dtype=torch.int64, device=ids.device)
    lib.call("cg3d_count_ids", ptr(ids), c_int64(n), c_int32(ids.stride(0)), c_int32(1 if ids.dtype == torch.int64 else 0), c_int32(m),
             ptr(counts), lib.stream())
    return counts


class GatherRowsFunction(torch.autograd.Function):
    """out = F[idx] with a scatter-add backward (atomics) instead of torch's sort-based index_put."""

    @staticmethod
    def forward(ctx, feats, idx):
        lib = _lib.get()
        feats = feats.contiguous()
        idx = idx.to(torch.int32).contiguous()
        lib.check(feats, idx)
        n, c = idx.shape[0], feats.shape[1]
        out = torch.empty((n, c), dtype=torch.float32, device=feats.device)
        lib.call("cg3d_gather_rows", ptr(feats), ptr(idx), ptr(out), c_int64(n), c_int32(c), lib.stream())
        ctx.save_for_backward(idx)
        ctx.n_src = feats.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        lib = _lib.get()
        dout = dout.contiguous()
        c = dout.shape[1]
        df = torch.zeros((ctx.n_src, c), dtype=torch.float32, device=dout.device)
        lib.call("cg3d_scatter_add_rows", ptr(dout), ptr(idx), ptr(df), c_int64(idx.shape[0]), c_int32(c), lib.stream())
        return df, None


def gather_rows(feats, idx):
    return GatherRowsFunction.apply(feats, idx)


ROI_CONTRACT = __import__("os").environ.get("CG3D_ROI_CONTRACT", "1") != "0"
ROI_CONTRACT_PARTIALS = __import__("os").environ.get("CG3D_ROI_CONTRACT_PARTIALS", "1") != "0"
ROI_CONTRACT_UNITS = int(__import__("os").environ.get("CG3D_ROI_CONTRACT_UNITS", "256"))      # workgroups of the split contraction (measured: 256 = 23 + 6 us, 512 = 18 + 10 us)
_roi_pair_cache = {}


def _roi_pairs(R, G, seglen, device):
    """Pair list of the per-RoI contraction's weight gradient: offset g pairs row r * G + g of the gathered grid rows
    with output row r.  int32 in [G * R], out [G * R] (offset-major), segment table (g, start, count <= seglen); cached."""
    ck = (R, G, seglen, str(device))

    def build():
        r, g = np.arange(R, dtype=np.int64), np.arange(G, dtype=np.int64)
        pin = (r[None, :] * G + g[:, None]).reshape(-1).astype(np.int32)
        pout = np.tile(r, G).astype(np.int32)
        starts = np.arange(0, R, seglen, dtype=np.int64)
        tab = np.stack([np.repeat(g, len(starts)), (g[:, None] * R + starts[None, :]).reshape(-1),
                        np.tile(np.minimum(seglen, R - starts), G)], 1).astype(np.int32)
        return (h2d(torch.from_numpy(pin), torch.int32, device), h2d(torch.from_numpy(pout), torch.int32, device),
                h2d(torch.from_numpy(tab), torch.int32, device), int(tab.shape[0]))
    return _cached(_roi_pair_cache, ck, build, 16)


class RoiContractFunction(torch.autograd.Function):
    """pooled[r] = sum_g feats[idx[r * G + g]] @ W[g]: the per-RoI G = 7^3 grid -> centre convolution of the RoI head
    (reference roi_heads/cagroup_roi_head.py:74-91 builds a fake 7^3 sparse tensor per RoI and convolves it with a
    kernel-size-7 MinkowskiConvolution evaluated at the centre; as a product it is [R, G C] x [G C, C2]).

    bench precision: the grid rows are gathered as bf16 (half the bytes of the fp32 gather they replace), the contraction
    runs on cg3d_linear_fwd with the G C = 43 904 channels split over the chip (ksplit), the data gradient is the same
    kernel on the other fragment-ordered copy of the weights followed by the scatter-add of the gather, the weight gradient
    the bf16-rows pair kernel on the (r G + g, r) pair list."""

    @staticmethod
    def available(feats, w):
        G, C, C2 = w.shape
        # (plain bf16 operands only: the split rows of the gathered grid points would interleave [hi | lo | hi] per point,
        # not per contraction -- the split precision gathers fp32 rows and runs `linear` on the flattened product)
        return ROI_CONTRACT and _prec() == 1 and BF16_ROWS and C % 64 == 0 and C2 % 64 == 0 and feats.shape[1] == C

    @staticmethod
    def forward(ctx, feats, idx, w):
        ctx.prec = _prec()
        lib = _lib.get()
        G, C, C2 = w.shape
        feats = feats.contiguous()
        idx = idx.to(torch.int32).contiguous()
        R = idx.shape[0] // G
        x16 = _to_bf16(feats, keep=True)
        g16 = torch.empty((R * G, C), dtype=torch.int16, device=feats.device)
        lib.check(x16, idx, g16)
        # a bf16 row of C channels moves as a row of C / 2 four-byte words
        lib.call("cg3d_gather_rows", ptr(x16), ptr(idx), ptr(g16), c_int64(R * G), c_int32(C // 2), lib.stream())
        wt, wp = _prep_frag(w.contiguous().view(1, G * C, C2), True, ctx.needs_input_grad[0])
        tiles, ny, nchunk = -(-R // 128), C2 // (128 if C2 % 128 == 0 else 64), G * C // 64
        ksplit = max(1, min(256, nchunk, ROI_CONTRACT_UNITS // max(tiles * ny, 1)))
        y = torch.empty((R, C2), dtype=torch.float32, device=feats.device)
        part = torch.empty((ksplit, R, C2), dtype=torch.float32, device=feats.device) if ksplit > 1 and ROI_CONTRACT_PARTIALS else None
        lib.call("cg3d_linear_fwd", ptr(g16), ptr(wt), ptr(None), ptr(y), c_int64(R), c_int32(G * C), c_int32(C2), c_int32(ksplit),
                 ptr(None), ptr(part), lib.stream())
        ctx.save_for_backward(g16, idx, w, wp)
        ctx.n_src = feats.shape[0]
        return y

    @staticmethod
    @_ctx_precision
    def backward(ctx, dy):
        g16, idx, w, wp = ctx.saved_tensors
        lib = _lib.get()
        G, C, C2 = w.shape
        R = dy.shape[0]
        dy16 = _to_bf16(dy.contiguous())
        dfeats = dw = None
        if ctx.needs_input_grad[0]:
            if wp is None:
                _, wp = _prep_frag(w.contiguous().view(1, G * C, C2), False, True)
            dflat = torch.empty((R, G * C), dtype=torch.float32, device=dy.device)
            lib.call("cg3d_linear_fwd", ptr(dy16), ptr(wp), ptr(None), ptr(dflat), c_int64(R), c_int32(C2), c_int32(G * C), c_int32(1),
                     ptr(None), ptr(None), lib.stream())
            dfeats = torch.zeros((ctx.n_src, C), dtype=torch.float32, device=dy.device)
            lib.call("cg3d_scatter_add_rows", ptr(dflat), ptr(idx), ptr(dfeats), c_int64(R * G), c_int32(C), lib.stream())
        if ctx.needs_input_grad[2]:
            pin, pout, seg, nseg = _roi_pairs(R, G, 512 if lib.is_device else 1 << 30, dy.device)
            dw = torch.empty_like(w)
            lib.call("cg3d_spconv_pairs_wgrad", ptr(g16), ptr(dy16), ptr(pin), ptr(pout), ptr(seg), c_int64(nseg), ptr(dw),
                     c_int32(G), c_int32(C), c_int32(C2), c_int32(2), lib.stream())
        return dfeats, None, dw


def roi_contract(feats, idx, w):
    """sum_g feats[idx[r G + g]] @ w[g] -> [R, C2] (w [G, C, C2]); the fp32 form gathers and multiplies with the library."""
    if RoiContractFunction.available(feats, w):
        return RoiContractFunction.apply(feats, idx, w)
    G, C, C2 = w.shape
    if _split() and LinearFunction._own(idx.shape[0] // G, G * C, C2):
        return linear(gather_rows(feats, idx).view(-1, G * C), w.view(G * C, C2))
    return gather_rows(feats, idx).view(-1, G * C) @ w.view(G * C, C2)


class ImplicitConvFunction(torch.autograd.Function):
    """Same convolution on the dense map (cg3d_spconv_fwd / cg3d_spconv_wgrad): deterministic, no atomics
    in forward / data gradient; wins only when the neighbourhood occupancy is high."""

    @staticmethod
    def forward(ctx, x, weight, bias, kmap):
        x = x.contiguous()
        w3 = weight.contiguous()
        ctx.save_for_backward(x, w3)
        ctx.kmap, ctx.has_bias = kmap, bias is not None
        return _conv_fwd_raw(x, w3, kmap.nbr, bias.contiguous() if bias is not None else None, kmap.n_out)

    @staticmethod
    def backward(ctx, dy):
        x, w3 = ctx.saved_tensors
        kmap = ctx.kmap
        dy = dy.contiguous()
        lib = _lib.get()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _conv_fwd_raw(dy, w3.transpose(1, 2).contiguous(), kmap.nbrT, None, kmap.n_in)
        if ctx.needs_input_grad[1]:
            K, cin, cout = w3.shape
            dw = torch.empty_like(w3)
            lib.check(x, dy, kmap.nbr)
            lib.call("cg3d_spconv_wgrad", ptr(x), ptr(dy), ptr(kmap.nbr), ptr(dw), c_int64(x.shape[0]),
                     c_int64(kmap.n_out), c_int32(K), c_int32(cin), c_int32(cout), c_int32(0), lib.stream())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None


class InterpolateFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feats, idx, w):
        lib = _lib.get()
        feats = feats.contiguous()
        lib.check(feats, idx, w)
        nq, c = idx.shape[0], feats.shape[1]
        out = torch.empty((nq, c), dtype=torch.float32, device=feats.device)
        lib.call("cg3d_interp_fwd", ptr(feats), ptr(idx), ptr(w), ptr(out), c_int64(nq), c_int32(c), lib.stream())
        ctx.save_for_backward(idx, w)
        ctx.n_src = feats.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, w = ctx.saved_tensors
        lib = _lib.get()
        dout = dout.contiguous()
        c = dout.shape[1]
        df = torch.zeros((ctx.n_src, c), dtype=torch.float32, device=dout.device)
        lib.call("cg3d_interp_bwd", ptr(dout), ptr(idx), ptr(w), ptr(df), c_int64(idx.shape[0]), c_int32(c),
                 lib.stream())
        return df, None, None


def interp_tables(src_map, q):
    """Corner rows int32 [nq, 8] (-1 = absent) and weights float32 [nq, 8] of the trilinear interpolation of the coordinate
    map `src_map` at the continuous coordinates q float32 [nq, 4] (cg3d_interp_map): coordinates only."""
    lib = _lib.get()
    nq = q.shape[0]
    idx = torch.empty((max(nq, 1), 8), dtype=torch.int32, device=q.device)
    w = torch.empty((max(nq, 1), 8), dtype=torch.float32, device=q.device)
    lib.check(q)
    lib.call("cg3d_interp_map", ptr(q), c_int64(nq), c_int32(src_map.tensor_stride), ptr(src_map.keys), ptr(src_map.vals),
             c_int64(src_map.cap), ptr(idx), ptr(w), lib.stream())
    return idx[:nq], w[:nq]


class ScatterMeanFunction(torch.autograd.Function):
    """out[m] = mean of F[i] over {(j, i): map[j, i] == m}; map int32 [J, n_in]."""

    @staticmethod
    def forward(ctx, feats, smap, n_out):
        lib = _lib.get()
        feats = feats.contiguous()
        lib.check(feats, smap)
        J, n_in = smap.shape
        c = feats.shape[1]
        out = torch.empty((n_out, c), dtype=torch.float32, device=feats.device)
        cnt = torch.empty(max(n_out, 1), dtype=torch.float32, device=feats.device)
        lib.call("cg3d_scatter_mean_fwd", ptr(feats), ptr(smap), c_int32(J), ptr(out), ptr(cnt), c_int64(n_in),
                 c_int64(n_out), c_int32(c), lib.stream())
        ctx.save_for_backward(smap, cnt)
        ctx.n_out = n_out
        return out

    @staticmethod
    def backward(ctx, dout):
        smap, cnt = ctx.saved_tensors
        lib = _lib.get()
        dout = dout.contiguous()
        J, n_in = smap.shape
        c = dout.shape[1]
        df = torch.empty((n_in, c), dtype=torch.float32, device=dout.device)
        lib.call("cg3d_scatter_mean_bwd", ptr(dout), ptr(cnt), ptr(smap), c_int32(J), ptr(df), c_int64(n_in),
                 c_int64(ctx.n_out), c_int32(c), lib.stream())
        return df, None, None


# ----------------------------------------------------------------------------- fused BatchNorm (+res) (+act)
ACT_NONE, ACT_RELU, ACT_ELU = 0, 1, 2
_BN_CHUNK = 256
# most chunks (= workgroups, = atomic additions per address of the statistics table / CG3D_BN_SLOTS) of a reduce launch
# (measured on MI355X, 155 773 x 64 bf16 rows: 23.8 / 19.2 / 22.0 / 31.3 us per backward-statistics launch at 1024 / 512 / 256 / 128)
BN_RED_DIV = int(__import__("os").environ.get("CG3D_BN_RED_DIV", "0"))
BN_CHUNK_SCALE = int(__import__("os").environ.get("CG3D_BN_CHUNK_SCALE", "1"))       # rows per apply chunk, in units of the round-3 rule
BN_RED_CHUNKS = int(__import__("os").environ.get("CG3D_BN_RED_CHUNKS", "512"))
_chunk_cache = {}


BN_CHUNKS_NATIVE = __import__("os").environ.get("CG3D_BN_CHUNKS_NATIVE", "1") != "0"


def _bn_chunks(bounds, device, C=64, _force_numpy=False):
    """Chunk tables for row groups `bounds` (host tuple of G+1 offsets), cached on the device:
    reduce table (<=1024 chunks per group: few, long chunks for the statistics kernels + their group offsets),
    apply table (many workgroups for the streaming kernels), group_n float32 [G].
    A 256-thread workgroup covers 256 / (C/4) rows per trip (4 trips in flight), so a chunk is 8 trips' worth of rows:
    128 rows at C = 64 down to 8 at C = 1024 -- with 128-row chunks for every C the 5330 x 512 and 1229 x 1024 layers
    ran 42 / 10 workgroups, each thread walking 64 / 128 rows one latency at a time (23 / 41 us for 11 / 5 MB)."""
    rpb = 256 // max(1, min(C // 4, 256))
    step_rows = max(8, min(128 * BN_CHUNK_SCALE, 8 * rpb * BN_CHUNK_SCALE))
    ck = (bounds, device, step_rows, C if BN_RED_DIV else 0)

    def build_native():
        # the five tables by the library (cg3d_host_bn_chunks: C, outside the interpreter lock); `build` below is the specification
        b = np.asarray(bounds, dtype=np.int64)
        G = b.shape[0] - 1
        cap = 6 * (int(b[-1] - b[0]) // step_rows + G + 2) + 4 * G + 32
        flat = np.empty(cap, np.int32)
        offs, sizes = np.zeros(5, np.int64), np.zeros(5, np.int64)
        nred, napp, tot = ctypes.c_int64(0), ctypes.c_int64(0), ctypes.c_int64(0)

        def ref(x):
            return ctypes.cast(ctypes.pointer(x), ctypes.c_void_p)
        rc = _lib.get().raw("cg3d_host_bn_chunks")(b.ctypes.data, G, step_rows, max(step_rows, C // BN_RED_DIV if BN_RED_DIV else 0),
                                                   BN_RED_CHUNKS, flat.ctypes.data, cap, offs.ctypes.data, sizes.ctypes.data,
                                                   ref(nred), ref(napp), ref(tot))
        if rc != 0:
            raise _lib.CG3DError("cg3d_host_bn_chunks failed with status %d" % rc)
        dev_flat = h2d(torch.from_numpy(flat[:tot.value]), torch.int32, device)
        o, z = offs.tolist(), sizes.tolist()

        def piece(i):
            return dev_flat[o[i]:o[i] + z[i]]
        return (piece(0).view(-1, 3), int(nred.value), piece(2), piece(3).view(torch.float32), piece(1).view(-1, 3), int(napp.value),
                piece(4).view(torch.float32).view(-1, 1))

    def build():
        b = np.asarray(bounds, dtype=np.int64)
        ng = b[1:] - b[:-1]

        def table(step):                      # step: int64 [G] rows per chunk of each group
            nch = -(-ng // np.maximum(step, 1))                        # chunks per group (0 for an empty group)
            gco = np.concatenate([[0], np.cumsum(nch)])
            g = np.repeat(np.arange(len(ng), dtype=np.int64), nch)
            j = np.arange(int(gco[-1]), dtype=np.int64) - gco[g]
            r0 = b[g] + j * step[g]
            rows = np.stack([g, r0, np.minimum(step[g], b[g + 1] - r0)], 1).astype(np.int32)
            if rows.shape[0] == 0:
                rows = np.zeros((1, 3), np.int32)
            return rows, int(gco[-1]), gco.astype(np.int32)
        # (wide layers: a reduce workgroup ends with 2 C atomics whatever its rows -- at 8 rows per chunk the 1 229 x 1 024 layer
        # issued 315 k atomics for 1.3 M elements: at least C / BN_RED_DIV rows per chunk)
        red, nred, gco = table(np.maximum(max(step_rows, C // BN_RED_DIV if BN_RED_DIV else 0), -(-ng // BN_RED_CHUNKS)))
        app, napp, _ = table(np.full_like(ng, step_rows))
        ns = np.maximum(ng, 1).astype(np.float64)
        unb = (ns / np.maximum(ns - 1, 1)).astype(np.float32)          # biased -> unbiased variance
        # ONE upload for the five tables (they were five: ~17 us of host time each, twice per step for the class branches'
        # ever-changing bounds): 4-byte words, every piece at a 16-byte boundary
        parts = [red.reshape(-1), app.reshape(-1), gco, ns.astype(np.float32).view(np.int32), unb.view(np.int32)]
        offs, tot = [], 0
        for q in parts:
            offs.append(tot)
            tot += (q.size + 3) & ~3
        flat = np.zeros(tot, np.int32)
        for q, o in zip(parts, offs):
            flat[o:o + q.size] = q
        dev_flat = h2d(torch.from_numpy(flat), torch.int32, device)

        def piece(i):
            return dev_flat[offs[i]:offs[i] + parts[i].size]
        return (piece(0).view(-1, 3), nred, piece(2), piece(3).view(torch.float32), piece(1).view(-1, 3), napp,
                piece(4).view(torch.float32).view(-1, 1))
    return _cached(_chunk_cache, ck, build_native if (BN_CHUNKS_NATIVE and not _force_numpy) else build, 512)


class FusedBNActFunction(torch.autograd.Function):
    """y = act(BN_g(x) + residual) over row groups; statistics and all kernels are HIP (cg3d_bn_*).
    Returns (y, batch_mean [G,C], batch_var_biased [G,C])."""

    @staticmethod
    def forward(ctx, x, gamma, beta, residual, bounds, act, use_batch, mean_in, var_in, eps, running=None, sync=None):
        ctx.prec = _prec()
        # running: None or (running_mean [G*C], running_var, num_batches_tracked, momentum), updated in the statistics launch
        # sync: None or (process group,): statistics over every rank's rows (--sync_bn; reference tools/train.py:118-119)
        lib = _lib.get()
        x = x.contiguous()
        N, C = x.shape
        G = len(bounds) - 1
        chunks, nchunk, gco, group_n, achunks, nachunk, _ = _bn_chunks(bounds, x.device, C)
        gamma, beta = gamma.contiguous().view(G, C), beta.contiguous().view(G, C)
        res = residual.contiguous() if residual is not None else None
        lib.check(x, gamma, beta, res, chunks)
        y = torch.empty_like(x)
        # bf16 mode: the apply kernels also write the bf16 row copy the neighbouring convolution gathers from
        # (forward: y16 -> its input; backward: dx16 -> its output gradient), instead of separate cg3d_to_bf16 passes
        want16 = _want_rows16(C)
        y16 = torch.empty(x.shape, dtype=torch.int16, device=x.device) if want16 else None
        if use_batch:
            # (from the zero block: an EMPTY group has no chunk, hence no workgroup that writes its mean / variance)
            mv = zero_arena().take(2 * G * C, x.device).view(2, G, C)
            mean, var = mv[0], mv[1]
            rm, rv, nbt, mom = running if running is not None else (None, None, None, 0.0)
            lib.check(rm, rv, nbt)
            pre = _STATS.pop(x.data_ptr(), None) if G == 1 else None
            if pre is not None and pre[2] == N and pre[3] == C:
                sums = pre[0]            # the producing convolution already summed its output per channel (tile kernel epilogue)
            else:
                sums = zero_arena().take(BN_SLOTS * 2 * G * C, x.device)
                lib.call("cg3d_bn_sums", ptr(x), ptr(chunks), c_int64(nchunk), c_int32(G), c_int32(C), ptr(sums), lib.stream())
            if sync is not None:
                # (the cached group_n holds max(rows, 1): a group that is empty HERE must add 0 rows to the global count)
                rows = h2d(torch.from_numpy(np.diff(np.asarray(bounds, dtype=np.int64)).astype(np.float32)), torch.float32, x.device)
                sums, group_n = _all_ranks_sums(sums, rows, G * C, sync[0])
                if nachunk == 0:
                    nachunk = 1        # no rows HERE: one empty chunk still writes mean / var and the running statistics of the global batch
            # mean / variance are derived from the table inside the apply launch (and written out for the backward pass)
            lib.call("cg3d_bn_apply_sums", ptr(x), ptr(res), ptr(achunks), c_int64(nachunk), c_int32(G), c_int32(C), ptr(sums),
                     ptr(group_n), c_float(eps), ptr(gamma), ptr(beta), c_int32(act), ptr(y), ptr(y16), ptr(mean), ptr(var),
                     ptr(rm), ptr(rv), ptr(nbt), c_float(mom), lib.stream())
            # the backward pass's table (sum dz, sum dz * xhat), reserved now: the block is zero and nobody else gets this slice
            ctx.dsums = zero_arena().take(BN_SLOTS * 2 * G * C, x.device) if any(ctx.needs_input_grad[:3]) else None
        else:
            mean, var = mean_in.contiguous().view(G, C), var_in.contiguous().view(G, C)
            lib.call("cg3d_bn_apply", ptr(x), ptr(res), ptr(achunks), c_int64(nachunk), c_int32(C), ptr(mean), ptr(var),
                     c_float(eps), ptr(gamma), ptr(beta), c_int32(act), ptr(y), ptr(y16), lib.stream())
            ctx.dsums = None
        ctx.want16 = want16
        ctx.sync = sync if use_batch else None
        if want16:
            _ROWS16[y.data_ptr()] = (y, y16)
        ctx.save_for_backward(x, y, mean, var, gamma, chunks, gco, group_n, achunks)
        ctx.meta = (nchunk, G, C, act, bool(use_batch), residual is not None, float(eps), nachunk)
        ctx.set_materialize_grads(False)      # or autograd fills a zero gradient for `mean` and `var` on every backward
        # (group_n: the row counts the statistics were taken over -- every rank's rows under --sync_bn; the caller's
        # running-variance update needs them)
        n_rows = group_n.detach()
        ctx.mark_non_differentiable(mean, var, n_rows)
        return y, mean, var, n_rows

    @staticmethod
    @_ctx_precision
    def backward(ctx, dy, _dm, _dv, _dn=None):
        x, y, mean, var, gamma, chunks, gco, group_n, achunks = ctx.saved_tensors
        nchunk, G, C, act, use_batch, has_res, eps, nachunk = ctx.meta
        if dy is None:
            if ctx.sync is None:
                return (None,) * 12
            # --sync_bn: the peers are about to all-reduce this layer's backward table -- a rank-local "no gradient reached
            # this output" must not skip the collective (the ranks would pair different all-reduces, or hang)
            dy = torch.zeros_like(x)
        lib = _lib.get()
        dy = dy.contiguous()
        dsums = ctx.dsums if getattr(ctx, "dsums", None) is not None else torch.zeros(BN_SLOTS * 2 * G * C, dtype=torch.float32, device=x.device)
        ctx.dsums = None                                  # (a second backward through the same node gets a fresh table)
        lib.call("cg3d_bn_bwd_sums", ptr(dy), ptr(x), ptr(y), ptr(chunks), c_int64(nchunk), c_int32(G), c_int32(C), ptr(mean),
                 ptr(var), c_float(eps), c_int32(act), ptr(dsums), lib.stream())
        local = None
        if ctx.sync is not None:
            # dx needs the sums over every rank's rows (group_n already holds the global row counts); the parameters'
            # gradients stay this rank's own sums, which the gradient exchange averages like every other gradient
            local = dsums.view(BN_SLOTS, 2, G, C).sum(0)
            dsums, _ = _all_ranks_sums(dsums, None, G * C, ctx.sync[0])
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if has_res else None
        dx16 = torch.empty(x.shape, dtype=torch.int16, device=x.device) if ctx.want16 else None
        dpar = zero_arena().take(2 * G * C, x.device).view(2, G, C)          # zeros: an empty group's gradients stay 0
        dbeta, dgamma = dpar[0], dpar[1]
        # dbeta / dgamma: the slot sums of the table, added up inside the apply launch (and written out as the gradients)
        lib.call("cg3d_bn_bwd_apply_sums", ptr(dy), ptr(x), ptr(y), ptr(achunks), c_int64(nachunk), c_int32(G), c_int32(C), ptr(mean),
                 ptr(var), c_float(eps), ptr(gamma), ptr(dsums), ptr(group_n), c_int32(act), c_int32(1 if use_batch else 0),
                 ptr(dx), ptr(dx16), ptr(dres), ptr(dbeta), ptr(dgamma), lib.stream())
        if dx16 is not None:
            _ROWS16[dx.data_ptr()] = (dx, dx16)
        if local is not None:
            dbeta, dgamma = local[0], local[1]
        return dx, dgamma, dbeta, dres, None, None, None, None, None, None, None, None


def _all_ranks_sums(table, group_n, gc, group):
    """A statistics table [CG3D_BN_SLOTS][2][G][C] (and the groups' row counts) summed over the ranks of `group`: one
    all-reduce of 2 G C (+ G) floats.  Returns a fresh table holding the totals in slot 0, and the global row counts."""
    import torch.distributed as dist
    tot = table.view(BN_SLOTS, 2 * gc).sum(0)
    buf = torch.cat([tot, group_n.to(tot.dtype)]) if group_n is not None else tot
    dist.all_reduce(buf, group=group)
    out = torch.zeros_like(table)
    out[:2 * gc] = buf[:2 * gc]
    return out, (buf[2 * gc:].clamp(min=1).contiguous() if group_n is not None else None)


def _sync_group_of(bn):
    """(process group,) when `bn` is a torch.nn.SyncBatchNorm in training mode inside an initialised job of > 1 ranks."""
    if not isinstance(bn, torch.nn.SyncBatchNorm) or not bn.training:
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return (group,) if dist.get_world_size(group) > 1 else None


class BNStack:
    """The parameters and running statistics of G BatchNorm1d modules (the class branches: one module per class, reference
    cagroup_head.py:183-188) as slices of ONE [G, C] tensor each, so that a grouped launch addresses them as a plain array
    (no torch.stack per step, the running statistics of all groups updated inside the apply launch).  The modules keep their
    own Parameter / buffer objects -- only their `.data` moves into the stacked storage; `state_dict` keys and shapes are
    unchanged.  A `.to()` / `load` that gives a module fresh storage is noticed (address check) and the stack rebuilt."""
    ATTRS = ("weight", "bias", "running_mean", "running_var", "num_batches_tracked")
    __slots__ = ATTRS + ("first", "last", "C")

    @staticmethod
    def of(bns):
        b0 = bns[0]
        st = b0.__dict__.get("_cg3d_stack")
        if st is not None and len(bns) == st.weight.shape[0] and st.valid(bns):
            return st
        st = BNStack(bns)
        b0.__dict__["_cg3d_stack"] = st
        return st

    def valid(self, bns):
        C, w0, m0 = self.C * 4, self.first[0], self.first[3]
        for g, b in enumerate(bns):           # every module's weight and running mean; the other three of the last module
            if b.weight.data_ptr() != w0 + g * C or b.running_mean.data_ptr() != m0 + g * C:
                return False
        b, g = bns[-1], self.last
        return (b.running_var.data_ptr() == self.first[1] + g * C and b.bias.data_ptr() == self.first[2] + g * C
                and b.num_batches_tracked.data_ptr() == self.first[4] + g * 8)

    def __init__(self, bns):
        with torch.no_grad():
            for a in BNStack.ATTRS:
                stacked = torch.stack([getattr(b, a).data for b in bns]).contiguous()
                for g, b in enumerate(bns):
                    getattr(b, a).data = stacked[g]
                setattr(self, a, stacked)
        self.C, self.last = self.weight.shape[1], len(bns) - 1
        self.first = (self.weight.data_ptr(), self.running_var.data_ptr(), self.bias.data_ptr(), self.running_mean.data_ptr(),
                      self.num_batches_tracked.data_ptr())


def _stackable(bns):
    b0 = bns[0]
    return (len(bns) > 1 and b0.track_running_stats and b0.momentum is not None and b0.weight is not None
            and all(type(b) is type(b0) and b.momentum == b0.momentum for b in bns))


def fused_bn_act(feats, bns, bounds=None, act=ACT_NONE, residual=None):
    """BatchNorm1d modules `bns` (one per contiguous row group of `bounds`) + residual + activation in
    two launches (statistics, apply); updates the modules' running statistics like nn.BatchNorm1d."""
    if coords_only():
        return feats
    bns = list(bns)
    G, N, C = len(bns), feats.shape[0], feats.shape[1]
    if bounds is None:
        bounds = (0, N)
    assert len(bounds) == G + 1 and bounds[-1] == N
    if C % 4 != 0:      # not on this path (every normalised tensor has C % 64 == 0)
        out = torch.cat([bn(feats[bounds[g]:bounds[g + 1]]) for g, bn in enumerate(bns)], 0)
        out = out if residual is None else out + residual
        return torch.relu(out) if act == ACT_RELU else (torch.nn.functional.elu(out) if act == ACT_ELU else out)
    b0 = bns[0]
    use_batch = b0.training or not b0.track_running_stats
    if G == 1:
        gamma, beta = b0.weight.view(1, C), b0.bias.view(1, C)
    else:
        gamma, beta = torch.stack([b.weight for b in bns]), torch.stack([b.bias for b in bns])
    mean_in = var_in = None
    if not use_batch:
        mean_in = torch.stack([b.running_mean for b in bns]) if G > 1 else b0.running_mean.view(1, C)
        var_in = torch.stack([b.running_var for b in bns]) if G > 1 else b0.running_var.view(1, C)
    track = b0.training and b0.track_running_stats
    running = None
    if track and G == 1 and b0.momentum is not None:
        running = (b0.running_mean, b0.running_var, b0.num_batches_tracked, float(b0.momentum))
    elif track and G > 1 and GROUPED_BN_STACK and _stackable(bns):
        # all groups' running statistics as [G, C] arrays (BNStack): updated inside the apply launch like the single group's
        st = BNStack.of(bns)
        running = (st.running_mean, st.running_var, st.num_batches_tracked, float(b0.momentum))
    sync = _sync_group_of(b0) if use_batch else None
    y, mean, var, n_rows = FusedBNActFunction.apply(feats, gamma, beta, residual, tuple(bounds), act, use_batch, mean_in, var_in,
                                                    b0.eps, running, sync)
    if track and running is None:
        with torch.no_grad():
            m = b0.momentum
            if sync is not None:
                n_all = n_rows.view(-1, 1)
                unb = var * (n_all / (n_all - 1).clamp(min=1))
            else:
                unb = var * _bn_chunks(tuple(bounds), feats.device, C)[6]
            rms, rvs = [b.running_mean for b in bns], [b.running_var for b in bns]
            torch._foreach_lerp_(rms, list(mean.unbind(0)), m)            # r += m * (stat - r)
            torch._foreach_lerp_(rvs, list(unb.unbind(0)), m)
            torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
    return y


GROUPED_BN_STACK = __import__("os").environ.get("CG3D_BN_STACK", "1") != "0"
_unit_cache = {}
ADD_RELU = __import__("os").environ.get("CG3D_ADD_RELU", "1") != "0"


def _unit_bn(C, device):
    """(zeros, ones) float32 [1, C]: the identity BatchNorm (mean 0, variance 1, gamma 1, beta 0, eps 0)."""
    ck = (C, str(device))
    return _cached(_unit_cache, ck, lambda: (torch.zeros((1, C), dtype=torch.float32, device=device),
                                             torch.ones((1, C), dtype=torch.float32, device=device)))


class AddReluFunction(torch.autograd.Function):
    """y = relu(a [+ b]) on feature rows in ONE pass that also leaves the bf16 row copy the next convolution gathers from --
    the `x = relu(lo + down(hi))` joins of BiResNet (reference backbones_3d/biresnet.py:358-406) were an add, a ReLU and a
    cg3d_to_bf16 launch (26 bytes per element instead of 14).  It is cg3d_bn_apply with the identity normalisation
    (mean 0, variance 1, eps 0: rsqrt(1) = 1, so (x - 0) * 1 * 1 + 0 + b is exact) -- no new kernel."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.prec = _prec()
        lib = _lib.get()
        a = a.contiguous()
        b = b.contiguous() if b is not None else None
        N, C = a.shape
        zeros, ones = _unit_bn(C, a.device)
        _, _, _, _, achunks, nachunk, _ = _bn_chunks((0, N), a.device, C)
        y = torch.empty_like(a)
        want16 = _want_rows16(C)
        y16 = torch.empty(a.shape, dtype=torch.int16, device=a.device) if want16 else None
        lib.check(a, b, achunks)
        lib.call("cg3d_bn_apply", ptr(a), ptr(b), ptr(achunks), c_int64(nachunk), c_int32(C), ptr(zeros), ptr(ones), c_float(0.0),
                 ptr(ones), ptr(zeros), c_int32(ACT_RELU), ptr(y), ptr(y16), lib.stream())
        if want16:
            _ROWS16[y.data_ptr()] = (y, y16)
        ctx.save_for_backward(y)
        ctx.two = b is not None
        return y

    @staticmethod
    @_ctx_precision
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dz = torch.ops.aten.threshold_backward(dy, y, 0.0)
        return dz, (dz if ctx.two else None)


def add_relu(a, b=None):
    """relu(a + b) (b None: relu(a)) for SparseTensors on the same map or feature rows."""
    sp = isinstance(a, SparseTensor)
    fa = a.F if sp else a
    fb = (b.F if isinstance(b, SparseTensor) else b) if b is not None else None
    if sp and isinstance(b, SparseTensor):
        a._same_map(b)
    if coords_only():
        return a
    if ADD_RELU and fa.dim() == 2 and fa.shape[1] % 4 == 0 and fa.dtype == torch.float32 and fa.shape[0] > 0:
        out = AddReluFunction.apply(fa, fb)
    else:
        out = torch.relu(fa if fb is None else fa + fb)
    return a._like(out) if sp else out


# ----------------------------------------------------------------------------- SparseTensor
class SparseTensor:
    """ME.SparseTensor: features [N, C] on a coordinate map.

    `coordinates` may be float (floored, like ME) or int; duplicates are merged: the default mode
    keeps the FIRST row of a voxel, UNWEIGHTED_AVERAGE averages all rows of a voxel."""

    def __init__(self, features=None, coordinates=None, tensor_stride=1, coordinate_map_key=None,
                 coordinate_manager=None, quantization_mode=SparseTensorQuantizationMode.RANDOM_SUBSAMPLE, **_):
        if coordinates is not None:
            assert features is not None
            self.coordinate_manager = coordinate_manager if coordinate_manager is not None else CoordinateManager()
            if coordinates.dtype.is_floating_point:
                coordinates = torch.floor(coordinates)
            ci = coordinates.to(torch.int32).contiguous()
            self._from_map(self.coordinate_manager.insert(ci, int(tensor_stride), sort=True), ci.shape[0], features, quantization_mode)
        else:
            assert coordinate_map_key is not None and coordinate_manager is not None
            self.coordinate_manager, self.coordinate_map_key = coordinate_manager, coordinate_map_key
            self.F = features
            self.unique_index = self.inverse_mapping = None
            self.rows_batch_major = False
        assert self.F.shape[0] == self._map.n, (self.F.shape, self._map.n)

    def _from_map(self, inserted, n_rows, features, quantization_mode):
        key, uniq, inv = inserted
        self.rows_batch_major = bool(MORTON_ROWS and n_rows > 1)    # (batch, Morton) row order: the batch column ascends
        self.coordinate_map_key = key
        self.unique_index, self.inverse_mapping = uniq, inv
        n_out = uniq.shape[0]
        if n_out == n_rows and not (MORTON_ROWS and n_out > 1):
            self.F = features                       # one row per voxel, rows in the caller's order
        elif quantization_mode == SparseTensorQuantizationMode.UNWEIGHTED_AVERAGE:
            self.F = ScatterMeanFunction.apply(features, inv.view(1, -1), n_out)
        else:
            self.F = gather_rows(features, uniq) if (features.dim() == 2 and features.dtype == torch.float32) else features[uniq.long()]

    @classmethod
    def build_many(cls, specs):
        """Several tensors from raw coordinates -- each dict(features, coordinates, tensor_stride=1, quantization_mode=...) -- with
        ONE host read for all their row counts: every map's ordering and build launches first, then the read, then the rest
        (the class branches build the fine and the coarse tensor of all classes back to back)."""
        begun = []
        for sp in specs:
            coordinates = sp["coordinates"]
            if coordinates.dtype.is_floating_point:
                coordinates = torch.floor(coordinates)
            ci = coordinates.to(torch.int32).contiguous()
            mgr = CoordinateManager()
            begun.append((mgr, ci.shape[0], mgr._insert_begin(ci, True)))
        host = torch.cat([b[2][1][7] for b in begun]).tolist()
        out = []
        for i, (sp, (mgr, n_rows, bg)) in enumerate(zip(specs, begun)):
            t = cls.__new__(cls)
            t.coordinate_manager = mgr
            t._from_map(mgr._insert_finish(bg, host[2 * i], host[2 * i + 1], int(sp.get("tensor_stride", 1))), n_rows, sp["features"],
                        sp.get("quantization_mode", SparseTensorQuantizationMode.RANDOM_SUBSAMPLE))
            assert t.F.shape[0] == t._map.n
            out.append(t)
        return out

    # -- accessors
    @property
    def _map(self):
        return self.coordinate_manager.get(self.coordinate_map_key)

    @property
    def C(self):
        return self._map.coords

    @property
    def coordinates(self):
        return self._map.coords

    @property
    def features(self):
        return self.F

    @property
    def tensor_stride(self):
        ts = self._map.tensor_stride
        return [ts, ts, ts]

    @property
    def device(self):
        return self.F.device

    def __len__(self):
        return self.F.shape[0]

    @property
    def decomposition_permutations(self):
        """Per-scene row index lists (ascending): `rows_by_batch`."""
        m = self._map
        if m._perms is None:
            if m.n == 0:
                m._perms = []
            else:
                info = {}
                m._perms = rows_by_batch(m.coords[:, 0], info=info)
                m._perms_sorted = bool(info.get("sorted"))
        return m._perms

    def batch_row_starts(self, n_batch):
        """Host list: first row of every scene when the rows of the map are batch-major and all n_batch scenes are present
        (then scene b is the row range [starts[b], starts[b + 1])), else None.  No device read beyond the one
        `decomposition_permutations` makes."""
        perms = self.decomposition_permutations
        if not getattr(self._map, "_perms_sorted", False) or len(perms) != n_batch or any(len(p) == 0 for p in perms):
            return None
        starts, r = [], 0
        for p in perms:
            starts.append(r)
            r += len(p)
        return starts

    @property
    def decomposed_coordinates(self):
        c = self.C
        return [c[p, 1:] for p in self.decomposition_permutations]

    @property
    def decomposed_features(self):
        return [self.F[p] for p in self.decomposition_permutations]

    def features_at_coordinates(self, query):
        """Trilinear interpolation of this tensor at continuous coordinates [nq,4] (b,x,y,z)."""
        if coords_only():
            return _fake(query.shape[0], self.F.shape[1], self.F)
        idx, w = interp_tables(self._map, query.to(torch.float32).contiguous())
        return InterpolateFunction.apply(self.F, idx, w)

    def _same_map(self, o):
        assert self.coordinate_map_key == o.coordinate_map_key and self.coordinate_manager is o.coordinate_manager, \
            "sparse tensors live on different coordinate maps"

    def _like(self, feats):
        return SparseTensor(features=feats, coordinate_map_key=self.coordinate_map_key,
                            coordinate_manager=self.coordinate_manager)

    def __add__(self, o):
        self._same_map(o)
        return self._like(self.F + o.F)

    def __iadd__(self, o):
        self._same_map(o)
        self.F = self.F + o.F
        return self


def cat(*tensors):
    for t in tensors[1:]:
        tensors[0]._same_map(t)
    if coords_only():
        return tensors[0]._like(_fake(tensors[0].F.shape[0], sum(t.F.shape[1] for t in tensors), tensors[0].F))
    return tensors[0]._like(torch.cat([t.F for t in tensors], dim=1))


# ----------------------------------------------------------------------------- modules
class _ConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, bias=False, dimension=3):
        super().__init__()
        assert dimension == 3
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.dilation = int(kernel_size), int(stride), int(dilation)
        self.kernel_volume = self.kernel_size ** 3
        shape = (self.kernel_volume, in_channels, out_channels) if self.kernel_volume > 1 else (in_channels, out_channels)
        self.kernel = nn.Parameter(torch.empty(shape, dtype=torch.float32))
        self.bias = nn.Parameter(torch.zeros(1, out_channels, dtype=torch.float32)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # ME default: uniform(-stdv, stdv), stdv = 1/sqrt(in_channels * kernel_volume)
        stdv = 1.0 / math.sqrt(self.in_channels * self.kernel_volume)
        with torch.no_grad():
            self.kernel.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)

    def _w3(self):
        return self.kernel.view(self.kernel_volume, self.in_channels, self.out_channels)

    def extra_repr(self):
        return "in=%d, out=%d, kernel_size=%d, stride=%d, dilation=%d" % (
            self.in_channels, self.out_channels, self.kernel_size, self.stride, self.dilation)


class MinkowskiConvolution(_ConvBase):
    """Sparse convolution; `forward(x, coordinates)` evaluates it at caller-given output coordinates
    (reference: cagroup_roi_head.py:69)."""

    def forward(self, x, coordinates=None, return_inverse=False):
        """return_inverse: `coordinates` may hold duplicates (they are merged, rows in first-occurrence order); the result is
        (tensor, inverse int32 [len(coordinates)]) with inverse[i] = output row of coordinate i."""
        mgr = x.coordinate_manager
        inv = None
        if coordinates is not None:
            out_key, _, inv = mgr.insert(coordinates.to(torch.int32).contiguous(), 1)
        elif self.stride > 1:
            out_key = mgr.stride(x.coordinate_map_key, self.stride)
        else:
            out_key = x.coordinate_map_key
        bias = self.bias.view(-1) if self.bias is not None else None
        if self.kernel_volume == 1 and coordinates is None and self.stride == 1:
            out = _fake(x.F.shape[0], self.out_channels, x.F) if coords_only() else linear(x.F, self.kernel, bias)
        else:
            km = mgr.kernel_map(x.coordinate_map_key, out_key, self.kernel_size, self.dilation, False)
            if coords_only():
                SparseConvFunction.warm(km, self.kernel_volume, self.in_channels, self.out_channels, None, self.training)
                out = _fake(km.n_out, self.out_channels, x.F)
            else:
                out = SparseConvFunction.apply(x.F, self._w3(), bias, km)
        res = SparseTensor(features=out, coordinate_map_key=out_key, coordinate_manager=mgr)
        return (res, inv) if return_inverse else res


class MinkowskiConvolutionTranspose(_ConvBase):
    """Transposed convolution onto the EXISTING map of tensor stride ts/stride (biresnet.py:309)."""

    def forward(self, x, coordinates=None):
        mgr = x.coordinate_manager
        in_ts = x._map.tensor_stride
        assert in_ts % self.stride == 0
        out_ts = in_ts // self.stride
        if coordinates is not None:
            out_key, _, _ = mgr.insert(coordinates.to(torch.int32).contiguous(), out_ts)
        else:
            cands = [k for k in mgr._maps if k.tensor_stride == out_ts and mgr._strided.get((k, self.stride)) == x.coordinate_map_key]
            assert cands, "transposed convolution needs the finer map it was strided from"
            out_key = cands[0]
        km = mgr.kernel_map(x.coordinate_map_key, out_key, self.kernel_size, self.dilation, True)
        bias = self.bias.view(-1) if self.bias is not None else None
        if coords_only():
            SparseConvFunction.warm(km, self.kernel_volume, self.in_channels, self.out_channels, None, self.training)
            return SparseTensor(features=_fake(km.n_out, self.out_channels, x.F), coordinate_map_key=out_key, coordinate_manager=mgr)
        out = SparseConvFunction.apply(x.F, self._w3(), bias, km)
        return SparseTensor(features=out, coordinate_map_key=out_key, coordinate_manager=mgr)


class MinkowskiGenerativeConvolutionTranspose(MinkowskiConvolutionTranspose):
    """Called with target coordinates on the CAGroup3D path (cagroup_head.py:274)."""


class MinkowskiAvgPooling(nn.Module):
    """Strided average over PRESENT inputs only (biresnet.py:109-127)."""

    def __init__(self, kernel_size, stride=1, dilation=1, dimension=3):
        super().__init__()
        assert dimension == 3 and dilation == 1
        self.kernel_size, self.stride = int(kernel_size), int(stride)

    def pool_map(self, mgr, in_key):
        """(pmap int32 [27, n_in], output map key): cg3d_pool_map, cached on the coordinate manager."""
        lib = _lib.get()
        src = mgr.get(in_key)
        out_key = mgr.stride(in_key, self.stride) if self.stride > 1 else in_key
        dst = mgr.get(out_key)
        ck = ("pool", in_key, out_key, self.kernel_size)
        pmap = mgr._kmaps.get(ck)
        if pmap is None:
            assert self.kernel_size // 2 * src.tensor_stride <= dst.tensor_stride, "pool kernel wider than 2*stride+1"
            pmap = torch.empty((27, max(src.n, 1)), dtype=torch.int32, device=src.coords.device)
            lib.call("cg3d_pool_map", ptr(src.coords), c_int64(src.n), c_int32(dst.tensor_stride),
                     c_int32((self.kernel_size - 1) // 2 * src.tensor_stride), ptr(dst.keys), ptr(dst.vals),
                     c_int64(dst.cap), ptr(pmap), lib.stream())
            pmap = pmap[:, :src.n].contiguous() if src.n > 0 else pmap[:, :0]
            mgr._kmaps[ck] = pmap
        return pmap, out_key

    def forward(self, x):
        mgr = x.coordinate_manager
        pmap, out_key = self.pool_map(mgr, x.coordinate_map_key)
        dst = mgr.get(out_key)
        out = _fake(dst.n, x.F.shape[1], x.F) if coords_only() else ScatterMeanFunction.apply(x.F, pmap, dst.n)
        return SparseTensor(features=out, coordinate_map_key=out_key, coordinate_manager=mgr)


class MinkowskiBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)

    def forward(self, x, act=ACT_NONE, residual=None):
        """BN, optionally fused with a residual add and ReLU/ELU (one statistics + one apply launch)."""
        return x._like(fused_bn_act(x.F, [self.bn], None, act, residual.F if residual is not None else None))


class _Pointwise(nn.Module):
    def forward(self, x):
        return x if coords_only() else x._like(self.fn(x.F))


class MinkowskiReLU(_Pointwise):
    def __init__(self, inplace=False):
        super().__init__()
        self.fn = nn.ReLU(inplace=False)

    def forward(self, x):
        return add_relu(x)          # one pass that also writes the bf16 row copy (AddReluFunction)


class MinkowskiELU(_Pointwise):
    def __init__(self, inplace=False):
        super().__init__()
        self.fn = nn.ELU()


class Sequential(nn.Sequential):
    """nn.Sequential that runs `MinkowskiBatchNorm` followed by `MinkowskiReLU/ELU` as ONE fused launch pair
    (statistics + apply).  Child indices -- and therefore state_dict keys -- are those of nn.Sequential."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if isinstance(m, MinkowskiBatchNorm) and isinstance(nxt, (MinkowskiReLU, MinkowskiELU)):
                x = m(x, act=ACT_RELU if isinstance(nxt, MinkowskiReLU) else ACT_ELU)
                i += 2
            else:
                x = m(x)
                i += 1
        return x


class utils:  # noqa: N801  (mirrors ME.utils)
    @staticmethod
    def kaiming_normal_(tensor, mode="fan_out", nonlinearity="relu"):
        """ME.utils.kaiming_normal_ on a [K, Cin, Cout] kernel (biresnet.py:329)."""
        if tensor.dim() == 3:
            K, cin, cout = tensor.shape
        else:
            (cin, cout), K = tensor.shape, 1
        fan = (cout if mode == "fan_out" else cin) * K
        gain = nn.init.calculate_gain(nonlinearity)
        with torch.no_grad():
            return tensor.normal_(0, gain / math.sqrt(fan))
