/*
 * cagroup3d_program.h -- launch programs: a whole network pass as ONE call.
 *
 * Why: a training step of the detector is ~700 C-ABI calls; issued one by one from the host language each costs 5-10 us of
 * argument marshalling on top of its launch, and the tensor-framework glue around it (an allocation and an autograd node
 * per op) several times that -- the step was bound by the HOST (30 ms of issue time against 27 ms of kernels, profiles/r04_*).
 * A program is the same sequence of calls written down as a table: int64 [nops][CG3D_PROG_STRIDE], row = { opcode, the
 * arguments of that entry point in declaration order }, pointers as addresses, int32 / int64 as themselves, float as the bit
 * pattern of the float32 in the low 32 bits.  cg3d_run_program walks the table and makes every call on `stream`; nothing is
 * allocated, nothing is synchronised.  The caller (cagroup3d_amd/engine.py) lays all activations, gradients and scratch out in
 * one arena and writes their addresses into the rows, so a forward or backward pass of the BiResNet backbone (reference
 * pcdet/models/backbones_3d/biresnet.py:358-406) is one call from the host language.
 *
 * The opcode table is part of the ABI (append only).  An opcode's arguments are EXACTLY those of the entry point named next
 * to it (see cagroup3d_hip.h), without the trailing stream.
 */
#ifndef CAGROUP3D_PROGRAM_H
#define CAGROUP3D_PROGRAM_H

#include "cagroup3d_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CG3D_PROG_STRIDE 24 /* int64 per row: opcode + up to 23 arguments */

enum {
    CG3D_OP_NOP = 0,
    CG3D_OP_MEMSET = 1,            /* dst, byte value, nbytes                                   (hipMemsetAsync)      */
    CG3D_OP_COPY2D = 2,            /* dst, dst pitch, src, src pitch, row bytes, rows           (device -> device)    */
    CG3D_OP_TO_BF16 = 3,           /* cg3d_to_bf16                                                                    */
    CG3D_OP_TILE_FWD = 4,          /* cg3d_spconv_tile_fwd                                                            */
    CG3D_OP_SPCONV_FWD = 5,        /* cg3d_spconv_fwd                                                                 */
    CG3D_OP_SPCONV_FWD_TILED = 6,  /* cg3d_spconv_fwd_tiled                                                           */
    CG3D_OP_PAIRS_FWD = 7,         /* cg3d_spconv_pairs_fwd                                                           */
    CG3D_OP_PAIRS_WGRAD = 8,       /* cg3d_spconv_pairs_wgrad                                                         */
    CG3D_OP_LINEAR_FWD = 9,        /* cg3d_linear_fwd                                                                 */
    CG3D_OP_BN_SUMS = 10,          /* cg3d_bn_sums                                                                    */
    CG3D_OP_BN_APPLY_SUMS = 11,    /* cg3d_bn_apply_sums                                                              */
    CG3D_OP_BN_APPLY = 12,         /* cg3d_bn_apply                                                                   */
    CG3D_OP_BN_BWD_SUMS = 13,      /* cg3d_bn_bwd_sums                                                                */
    CG3D_OP_BN_BWD_APPLY_SUMS = 14,/* cg3d_bn_bwd_apply_sums                                                          */
    CG3D_OP_BN_BWD_APPLY = 15,     /* cg3d_bn_bwd_apply                                                               */
    CG3D_OP_INTERP_MAP = 16,       /* cg3d_interp_map                                                                 */
    CG3D_OP_INTERP_FWD = 17,       /* cg3d_interp_fwd                                                                 */
    CG3D_OP_INTERP_BWD = 18,       /* cg3d_interp_bwd                                                                 */
    CG3D_OP_GATHER_ROWS = 19,      /* cg3d_gather_rows                                                                */
    CG3D_OP_SCATTER_ADD_ROWS = 20, /* cg3d_scatter_add_rows                                                           */
    CG3D_OP_SCATTER_MEAN_FWD = 21, /* cg3d_scatter_mean_fwd                                                           */
    CG3D_OP_SCATTER_MEAN_BWD = 22, /* cg3d_scatter_mean_bwd                                                           */
    CG3D_OP_EVENT_RECORD = 23,     /* event handle from cg3d_event_create                       (hipEventRecord)      */
    CG3D_OP_TO_BF16_SPLIT = 24,    /* cg3d_to_bf16_split                                                              */
    CG3D_OP_FROM_BF16 = 25,        /* cg3d_from_bf16                                                                  */
    CG3D_OP_EVENT_WAIT = 26,       /* event handle: the row's queue waits for the event         (hipStreamWaitEvent)  */
    CG3D_OP_COUNT
};

/* Lanes.  Bits 32-39 of a row's first word name the QUEUE the row is issued on (cg3d_run_program_lanes: lane l -> streams[l];
 * cg3d_run_program, or a lane >= nstreams: everything on the one stream -- table order is always a valid sequential order).
 * Rows of one lane run in table order; between lanes only CG3D_OP_EVENT_RECORD (on the producer's lane) followed, later in the
 * table, by CG3D_OP_EVENT_WAIT (on the consumer's lane) orders anything.  The BiResNet backbone is two chains between its
 * joins (reference biresnet.py:378-394: the stride-8/16/32 layers and the stride-4 layers): on one queue the stride-16/32
 * launches (42-180 tiles) leave most of the 256 CUs idle one after the other; on two queues the stride-4 chain fills them. */
#define CG3D_PROG_LANE_SHIFT 32
#define CG3D_PROG_MAX_LANES 8

/* Runs rows [0, nops) of `prog` in order on `stream`.  Returns CG3D_OK, or the status of the first failing call with its row
 * index in *fail_at (may be NULL); rows after it are not run.  An unknown opcode is CG3D_ERR_ARG. */
int cg3d_run_program(const int64_t *prog, int64_t nops, cg3d_stream_t stream, int64_t *fail_at);

/* The same with one stream per lane (1 <= nstreams <= CG3D_PROG_MAX_LANES).  Nothing is synchronised with the host; the caller's
 * table ends with the waits that bring every lane back to lane 0 (engine.py: _schedule). */
int cg3d_run_program_lanes(const int64_t *prog, int64_t nops, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at);

/* The event edges between the lanes of a table and the order its rows are issued in, derived from what every row reads and
 * writes (the const / non-const pointer parameters of the entry point its opcode names: cg3d_program_roles).
 *   prog          int64 [n][CG3D_PROG_STRIDE]: rows in emission order (a valid sequential order), lanes in word 0.  An address
 *                 >= 2^CG3D_PROG_REGION_SHIFT is region-relative (region = address >> CG3D_PROG_REGION_SHIFT, resolved by the
 *                 caller when the pass is run); the others are absolute;
 *   starts        the block starts (offsets) of the caller's allocations, region after region, ascending inside a region;
 *   region_first  int64 [CG3D_PROG_REGIONS + 1]: starts[region_first[r] .. region_first[r + 1]) belong to region r.  A pointer
 *                 refers to the block it falls into; an absolute pointer is a block by itself;
 *   cuts          row indices at which the caller splits the table (a callback runs between the parts): every part starts with
 *                 the other lanes waiting for lane 0's position and ends with lane 0 waiting for all of them;
 *   out           int64 [cap][CG3D_PROG_STRIDE]: the rows plus EVENT_RECORD / EVENT_WAIT rows { opcode | lane, slot, 1 } -- the
 *                 caller replaces the slot numbers 0 .. *n_events - 1 by event handles (cg3d_event_create_sync) before running;
 *                 inside a part the lanes' sequences are merged round robin (the table is also the host's issue order);
 *   index         int64 [n]: where row i went;   cut_index int64 [ncut]: where the table is to be split for cuts[k].
 * A row needs an edge from another lane when it reads a block that lane wrote last, or writes a block that lane read or wrote.
 * Returns CG3D_ERR_ARG when cap is too small (n + 2 * n * lanes always suffices) or a lane exceeds CG3D_PROG_MAX_LANES. */
#define CG3D_PROG_REGION_SHIFT 56
#define CG3D_PROG_REGIONS 16
int cg3d_program_schedule(const int64_t *prog, int64_t n, const int64_t *starts, const int64_t *region_first, const int64_t *cuts,
                          int32_t ncut, int64_t *out, int64_t cap, int64_t *index, int64_t *cut_index, int64_t *n_out,
                          int64_t *n_events);
/* bit c of *rd / *wr: column c of a row of `opcode` is a pointer the call reads / writes */
int cg3d_program_roles(int32_t opcode, uint32_t *rd, uint32_t *wr);

/* Host-side table of a pair-list launch (no device work; the host language's numpy form cost the issuing thread ~80 us per
 * table, ~70 tables per training step over its two threads).  Rows { weight index, start, count <= maxlen } cover the pairs of
 * every (offset k, group g) slot in slot order (slot = k * G + g; off int64 [K * G + 1] = the slots' first pairs; weight index
 * = g * K + k).  xcd_order != 0 and 64 <= rows <= 4096 and maxlen >= 256: the rows are re-ordered so that workgroup i (XCD
 * i % 8) sweeps ONE eighth of the tensor's rows through all offsets -- sorted (stable) by the position of a segment inside its
 * slot's list (mapped through row_bounds [G + 1] / n_rows when the groups are row ranges), cut into 8 runs, dealt out round
 * robin.  out int32 [cap][3]; *nseg rows written; CG3D_ERR_ARG when cap is too small (K * G + pairs / maxlen + 1 suffices). */
int cg3d_host_segments(const int64_t *off, int32_t K, int32_t G, int64_t maxlen, int32_t xcd_order, const int64_t *row_bounds,
                       int64_t n_rows, int32_t *out, int64_t cap, int64_t *nseg);

/* Host-side chunk tables of the grouped BatchNorm launches (cg3d_bn_sums / _apply_sums / _bwd_*; no device work), packed for ONE
 * upload: flat int32 [*total] holds five pieces, each starting at a multiple of 4 words --
 *   0 reduce chunks [max(nred, 1)][3] = { group, first row, rows }: rows per chunk of group g = max(red_min_rows,
 *     ceil(rows_g / red_chunks)) (few, long chunks for the statistics kernels);
 *   1 apply chunks [max(napp, 1)][3]: step_rows rows per chunk;  2 the reduce table's first chunk per group, int32 [G + 1];
 *   3 max(rows_g, 1) as float32 [G];  4 rows_g / max(rows_g - 1, 1) as float32 [G] (biased -> unbiased variance).
 * bounds int64 [G + 1] ascending; offs / sizes int64 [5] (words); cap = words available in flat. */
int cg3d_host_bn_chunks(const int64_t *bounds, int32_t G, int64_t step_rows, int64_t red_min_rows, int64_t red_chunks, int32_t *flat,
                        int64_t cap, int64_t *offs, int64_t *sizes, int64_t *nred, int64_t *napp, int64_t *total);

/* cg3d_run_program_lanes on a table that still holds region-relative addresses and event slots: every word >= 2^CG3D_PROG_REGION_SHIFT
 * of a row becomes bases[word >> shift] + (word's low bits) -- the host language's copy / mask / add over the whole table cost
 * ~0.15 ms per pass, seven passes per training step -- and RECORD / WAIT rows { opcode, slot, 1 } take events[slot].  Before the
 * first row zero_bytes bytes at zero_ptr are cleared on streams[0] (the pass's zero-filled region; 0 bytes: nothing). */
int cg3d_run_program_bound(const int64_t *prog, int64_t nops, const int64_t *bases, const int64_t *events, int64_t nevents,
                           void *zero_ptr, int64_t zero_bytes, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at);

/* Timing events for the rows of a program (CG3D_OP_EVENT_RECORD): handles are hipEvent_t on the device library; the oracle
 * hands out dummies and reports 0 ms. */
int cg3d_event_create(int64_t *handle);
int cg3d_event_create_sync(int64_t *handle); /* ordering only, no time stamps (hipEventDisableTiming) */
int cg3d_event_destroy(int64_t handle);
int cg3d_event_elapsed_ms(int64_t start, int64_t stop, float *ms); /* waits for `stop` */

#ifdef __cplusplus
}
#endif

/* --------------------------------------------------------------------------------------------------------------------
 * The dispatcher itself, shared by the device library and the CPU oracle (it contains no arithmetic).  A translation unit
 * that defines CG3D_PROGRAM_IMPL before including this header gets `cg3d_program_dispatch`; it must provide
 *   CG3D_PROG_MEMSET(dst, value, nbytes, stream)                      -> status
 *   CG3D_PROG_COPY2D(dst, dpitch, src, spitch, width, height, stream) -> status
 *   CG3D_PROG_EVENT_RECORD(handle, stream)                            -> status
 *   CG3D_PROG_EVENT_WAIT(handle, stream)                              -> status
 * ------------------------------------------------------------------------------------------------------------------ */
#ifdef CG3D_PROGRAM_IMPL
static inline float cg3d_prog_f(int64_t v) {
    union { uint32_t u; float f; } c;
    c.u = (uint32_t)(v & 0xffffffffll);
    return c.f;
}
#define CG3D_A_P(T, i) ((T)(intptr_t)a[i])
#define CG3D_A_I(i) ((int32_t)a[i])
#define CG3D_A_L(i) ((int64_t)a[i])
#define CG3D_A_F(i) cg3d_prog_f(a[i])
static int cg3d_program_dispatch(const int64_t *row, cg3d_stream_t s) {
    const int64_t *a = row + 1;
    switch ((int)row[0]) {
    case CG3D_OP_NOP: return CG3D_OK;
    case CG3D_OP_MEMSET: return CG3D_PROG_MEMSET(CG3D_A_P(void *, 0), CG3D_A_I(1), CG3D_A_L(2), s);
    case CG3D_OP_COPY2D:
        return CG3D_PROG_COPY2D(CG3D_A_P(void *, 0), CG3D_A_L(1), CG3D_A_P(const void *, 2), CG3D_A_L(3), CG3D_A_L(4), CG3D_A_L(5), s);
    case CG3D_OP_TO_BF16: return cg3d_to_bf16(CG3D_A_P(const float *, 0), CG3D_A_P(uint16_t *, 1), CG3D_A_L(2), s);
    case CG3D_OP_TILE_FWD:
        return cg3d_spconv_tile_fwd(CG3D_A_P(const uint16_t *, 0), CG3D_A_P(const uint16_t *, 1), CG3D_A_P(const uint16_t *, 2),
                                    CG3D_A_P(const uint8_t *, 3), CG3D_A_P(const int32_t *, 4), CG3D_A_P(const int32_t *, 5),
                                    CG3D_A_P(const int32_t *, 6), CG3D_A_I(7), CG3D_A_I(8), CG3D_A_P(const int32_t *, 9), CG3D_A_L(10),
                                    CG3D_A_P(const int32_t *, 11), CG3D_A_P(const float *, 12), CG3D_A_P(float *, 13), CG3D_A_L(14),
                                    CG3D_A_L(15), CG3D_A_I(16), CG3D_A_I(17), CG3D_A_I(18), CG3D_A_I(19), CG3D_A_I(20),
                                    CG3D_A_P(float *, 21), s);
    case CG3D_OP_SPCONV_FWD:
        return cg3d_spconv_fwd(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const int32_t *, 2),
                               CG3D_A_P(const float *, 3), CG3D_A_P(float *, 4), CG3D_A_L(5), CG3D_A_L(6), CG3D_A_I(7), CG3D_A_I(8),
                               CG3D_A_I(9), CG3D_A_I(10), s);
    case CG3D_OP_SPCONV_FWD_TILED:
        return cg3d_spconv_fwd_tiled(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const int32_t *, 2),
                                     CG3D_A_P(const int32_t *, 3), CG3D_A_L(4), CG3D_A_P(const float *, 5), CG3D_A_P(float *, 6),
                                     CG3D_A_L(7), CG3D_A_L(8), CG3D_A_I(9), CG3D_A_I(10), CG3D_A_I(11), CG3D_A_I(12), s);
    case CG3D_OP_PAIRS_FWD:
        return cg3d_spconv_pairs_fwd(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const int32_t *, 2),
                                     CG3D_A_P(const int32_t *, 3), CG3D_A_P(const int32_t *, 4), CG3D_A_L(5),
                                     CG3D_A_P(const float *, 6), CG3D_A_P(float *, 7), CG3D_A_L(8), CG3D_A_I(9), CG3D_A_I(10),
                                     CG3D_A_I(11), CG3D_A_I(12), s);
    case CG3D_OP_PAIRS_WGRAD:
        return cg3d_spconv_pairs_wgrad(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const int32_t *, 2),
                                       CG3D_A_P(const int32_t *, 3), CG3D_A_P(const int32_t *, 4), CG3D_A_L(5), CG3D_A_P(float *, 6),
                                       CG3D_A_I(7), CG3D_A_I(8), CG3D_A_I(9), CG3D_A_I(10), s);
    case CG3D_OP_LINEAR_FWD:
        return cg3d_linear_fwd(CG3D_A_P(const uint16_t *, 0), CG3D_A_P(const uint16_t *, 1), CG3D_A_P(const float *, 2),
                               CG3D_A_P(float *, 3), CG3D_A_L(4), CG3D_A_I(5), CG3D_A_I(6), CG3D_A_I(7), CG3D_A_P(float *, 8),
                               CG3D_A_P(float *, 9), s);
    case CG3D_OP_BN_SUMS:
        return cg3d_bn_sums(CG3D_A_P(const float *, 0), CG3D_A_P(const int32_t *, 1), CG3D_A_L(2), CG3D_A_I(3), CG3D_A_I(4),
                            CG3D_A_P(float *, 5), s);
    case CG3D_OP_BN_APPLY_SUMS:
        return cg3d_bn_apply_sums(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const int32_t *, 2), CG3D_A_L(3),
                                  CG3D_A_I(4), CG3D_A_I(5), CG3D_A_P(const float *, 6), CG3D_A_P(const float *, 7), CG3D_A_F(8),
                                  CG3D_A_P(const float *, 9), CG3D_A_P(const float *, 10), CG3D_A_I(11), CG3D_A_P(float *, 12),
                                  CG3D_A_P(uint16_t *, 13), CG3D_A_P(float *, 14), CG3D_A_P(float *, 15), CG3D_A_P(float *, 16),
                                  CG3D_A_P(float *, 17), CG3D_A_P(int64_t *, 18), CG3D_A_F(19), s);
    case CG3D_OP_BN_APPLY:
        return cg3d_bn_apply(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const int32_t *, 2), CG3D_A_L(3),
                             CG3D_A_I(4), CG3D_A_P(const float *, 5), CG3D_A_P(const float *, 6), CG3D_A_F(7),
                             CG3D_A_P(const float *, 8), CG3D_A_P(const float *, 9), CG3D_A_I(10), CG3D_A_P(float *, 11),
                             CG3D_A_P(uint16_t *, 12), s);
    case CG3D_OP_BN_BWD_SUMS:
        return cg3d_bn_bwd_sums(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const float *, 2),
                                CG3D_A_P(const int32_t *, 3), CG3D_A_L(4), CG3D_A_I(5), CG3D_A_I(6), CG3D_A_P(const float *, 7),
                                CG3D_A_P(const float *, 8), CG3D_A_F(9), CG3D_A_I(10), CG3D_A_P(float *, 11), s);
    case CG3D_OP_BN_BWD_APPLY_SUMS:
        return cg3d_bn_bwd_apply_sums(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const float *, 2),
                                      CG3D_A_P(const int32_t *, 3), CG3D_A_L(4), CG3D_A_I(5), CG3D_A_I(6), CG3D_A_P(const float *, 7),
                                      CG3D_A_P(const float *, 8), CG3D_A_F(9), CG3D_A_P(const float *, 10), CG3D_A_P(const float *, 11),
                                      CG3D_A_P(const float *, 12), CG3D_A_I(13), CG3D_A_I(14), CG3D_A_P(float *, 15),
                                      CG3D_A_P(uint16_t *, 16), CG3D_A_P(float *, 17), CG3D_A_P(float *, 18), CG3D_A_P(float *, 19), s);
    case CG3D_OP_BN_BWD_APPLY:
        return cg3d_bn_bwd_apply(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const float *, 2),
                                 CG3D_A_P(const int32_t *, 3), CG3D_A_L(4), CG3D_A_I(5), CG3D_A_P(const float *, 6),
                                 CG3D_A_P(const float *, 7), CG3D_A_F(8), CG3D_A_P(const float *, 9), CG3D_A_P(const float *, 10),
                                 CG3D_A_P(const float *, 11), CG3D_A_P(const float *, 12), CG3D_A_I(13), CG3D_A_I(14),
                                 CG3D_A_P(float *, 15), CG3D_A_P(uint16_t *, 16), CG3D_A_P(float *, 17), s);
    case CG3D_OP_INTERP_MAP:
        return cg3d_interp_map(CG3D_A_P(const float *, 0), CG3D_A_L(1), CG3D_A_I(2), CG3D_A_P(const uint64_t *, 3),
                               CG3D_A_P(const int32_t *, 4), CG3D_A_L(5), CG3D_A_P(int32_t *, 6), CG3D_A_P(float *, 7), s);
    case CG3D_OP_INTERP_FWD:
        return cg3d_interp_fwd(CG3D_A_P(const float *, 0), CG3D_A_P(const int32_t *, 1), CG3D_A_P(const float *, 2),
                               CG3D_A_P(float *, 3), CG3D_A_L(4), CG3D_A_I(5), s);
    case CG3D_OP_INTERP_BWD:
        return cg3d_interp_bwd(CG3D_A_P(const float *, 0), CG3D_A_P(const int32_t *, 1), CG3D_A_P(const float *, 2),
                               CG3D_A_P(float *, 3), CG3D_A_L(4), CG3D_A_I(5), s);
    case CG3D_OP_GATHER_ROWS:
        return cg3d_gather_rows(CG3D_A_P(const float *, 0), CG3D_A_P(const int32_t *, 1), CG3D_A_P(float *, 2), CG3D_A_L(3),
                                CG3D_A_I(4), s);
    case CG3D_OP_SCATTER_ADD_ROWS:
        return cg3d_scatter_add_rows(CG3D_A_P(const float *, 0), CG3D_A_P(const int32_t *, 1), CG3D_A_P(float *, 2), CG3D_A_L(3),
                                     CG3D_A_I(4), s);
    case CG3D_OP_SCATTER_MEAN_FWD:
        return cg3d_scatter_mean_fwd(CG3D_A_P(const float *, 0), CG3D_A_P(const int32_t *, 1), CG3D_A_I(2), CG3D_A_P(float *, 3),
                                     CG3D_A_P(float *, 4), CG3D_A_L(5), CG3D_A_L(6), CG3D_A_I(7), s);
    case CG3D_OP_SCATTER_MEAN_BWD:
        return cg3d_scatter_mean_bwd(CG3D_A_P(const float *, 0), CG3D_A_P(const float *, 1), CG3D_A_P(const int32_t *, 2),
                                     CG3D_A_I(3), CG3D_A_P(float *, 4), CG3D_A_L(5), CG3D_A_L(6), CG3D_A_I(7), s);
    case CG3D_OP_EVENT_RECORD: return CG3D_PROG_EVENT_RECORD(CG3D_A_L(0), s);
    case CG3D_OP_EVENT_WAIT: return CG3D_PROG_EVENT_WAIT(CG3D_A_L(0), s);
    case CG3D_OP_FROM_BF16: return cg3d_from_bf16(CG3D_A_P(const uint16_t *, 0), CG3D_A_P(float *, 1), CG3D_A_L(2), s);
    case CG3D_OP_TO_BF16_SPLIT:
        return cg3d_to_bf16_split(CG3D_A_P(const float *, 0), CG3D_A_P(uint16_t *, 1), CG3D_A_L(2), CG3D_A_I(3), s);
    default: return CG3D_ERR_ARG;
    }
}
static int cg3d_program_run(const int64_t *prog, int64_t nops, cg3d_stream_t stream, int64_t *fail_at) {
    if (nops < 0 || (nops > 0 && !prog)) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < nops; i++) {
        const int rc = cg3d_program_dispatch(prog + i * CG3D_PROG_STRIDE, stream);
        if (rc != CG3D_OK) {
            if (fail_at) *fail_at = i;
            return rc;
        }
    }
    return CG3D_OK;
}

/* ---- cg3d_program_schedule ---------------------------------------------------------------------------------------------- */
#include <stdlib.h>
#include <string.h>
#define CG3D_B(c) (1u << (c))
static int cg3d_program_roles_impl(int32_t op, uint32_t *rd, uint32_t *wr) {
    uint32_t r = 0, w = 0;
    switch (op) {
    case CG3D_OP_NOP: case CG3D_OP_EVENT_RECORD: case CG3D_OP_EVENT_WAIT: break;
    case CG3D_OP_MEMSET: w = (CG3D_B(1)); break;
    case CG3D_OP_COPY2D: r = (CG3D_B(3)); w = (CG3D_B(1)); break;
    case CG3D_OP_TO_BF16: case CG3D_OP_TO_BF16_SPLIT: case CG3D_OP_FROM_BF16: r = (CG3D_B(1)); w = (CG3D_B(2)); break;
    case CG3D_OP_TILE_FWD: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4) | CG3D_B(5) | CG3D_B(6) | CG3D_B(7) | CG3D_B(10) | CG3D_B(12) | CG3D_B(13)); w = (CG3D_B(14) | CG3D_B(22)); break;
    case CG3D_OP_SPCONV_FWD: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4)); w = (CG3D_B(5)); break;
    case CG3D_OP_SPCONV_FWD_TILED: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4) | CG3D_B(6)); w = (CG3D_B(7)); break;
    case CG3D_OP_PAIRS_FWD: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4) | CG3D_B(5) | CG3D_B(7)); w = (CG3D_B(8)); break;
    case CG3D_OP_PAIRS_WGRAD: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4) | CG3D_B(5)); w = (CG3D_B(7)); break;
    case CG3D_OP_LINEAR_FWD: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3)); w = (CG3D_B(4) | CG3D_B(9) | CG3D_B(10)); break;
    case CG3D_OP_BN_SUMS: r = (CG3D_B(1) | CG3D_B(2)); w = (CG3D_B(6)); break;
    case CG3D_OP_BN_APPLY_SUMS: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(7) | CG3D_B(8) | CG3D_B(10) | CG3D_B(11)); w = (CG3D_B(13) | CG3D_B(14) | CG3D_B(15) | CG3D_B(16) | CG3D_B(17) | CG3D_B(18) | CG3D_B(19)); break;
    case CG3D_OP_BN_APPLY: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(6) | CG3D_B(7) | CG3D_B(9) | CG3D_B(10)); w = (CG3D_B(12) | CG3D_B(13)); break;
    case CG3D_OP_BN_BWD_SUMS: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4) | CG3D_B(8) | CG3D_B(9)); w = (CG3D_B(12)); break;
    case CG3D_OP_BN_BWD_APPLY_SUMS: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4) | CG3D_B(8) | CG3D_B(9) | CG3D_B(11) | CG3D_B(12) | CG3D_B(13)); w = (CG3D_B(16) | CG3D_B(17) | CG3D_B(18) | CG3D_B(19) | CG3D_B(20)); break;
    case CG3D_OP_BN_BWD_APPLY: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3) | CG3D_B(4) | CG3D_B(7) | CG3D_B(8) | CG3D_B(10) | CG3D_B(11) | CG3D_B(12) | CG3D_B(13)); w = (CG3D_B(16) | CG3D_B(17) | CG3D_B(18)); break;
    case CG3D_OP_INTERP_MAP: r = (CG3D_B(1) | CG3D_B(4) | CG3D_B(5)); w = (CG3D_B(7) | CG3D_B(8)); break;
    case CG3D_OP_INTERP_FWD: case CG3D_OP_INTERP_BWD: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3)); w = (CG3D_B(4)); break;
    case CG3D_OP_GATHER_ROWS: case CG3D_OP_SCATTER_ADD_ROWS: r = (CG3D_B(1) | CG3D_B(2)); w = (CG3D_B(3)); break;
    case CG3D_OP_SCATTER_MEAN_FWD: r = (CG3D_B(1) | CG3D_B(2)); w = (CG3D_B(4) | CG3D_B(5)); break;
    case CG3D_OP_SCATTER_MEAN_BWD: r = (CG3D_B(1) | CG3D_B(2) | CG3D_B(3)); w = (CG3D_B(5)); break;
    default: return CG3D_ERR_ARG;
    }
    if (rd) *rd = r;
    if (wr) *wr = w;
    return CG3D_OK;
}

/* Hazard keys of the scheduler -- THE SINGLE-BLOCK INVARIANT.  A pointer argument is ordered between lanes by ONE key: the
 * allocation block of its region that holds the pointer's FIRST byte (region-tagged pointers), or its exact address (absolute
 * pointers: weights, statistics tables, tensors of the caller).  The emitter must therefore keep every access inside the block
 * its pointer starts in, and must not hand two lanes two DIFFERENT absolute addresses into one buffer (slices of one tensor at
 * different offsets are not ordered against each other).  cagroup3d_amd/engine.py's Builder allocates one block per matrix
 * and passes block starts only; absolute pointers are whole tensors.  Nothing here can check the extent of an access: an
 * emitter that breaks the invariant gets no edge and a race.  (tests/test_engine_lanes.py walks the tables of the backbone
 * and the class branches with the Python specification of the same rule.) */
typedef struct { int64_t key; int32_t w[CG3D_PROG_MAX_LANES], r[CG3D_PROG_MAX_LANES]; } cg3d_sched_blk;
typedef struct { int64_t after_row, before_row; int32_t rec_lane, wait_lane, ev, next_after, next_before; } cg3d_sched_edge;

static int64_t cg3d_sched_block_of(int64_t a, const int64_t *starts, const int64_t *first) {
    const int64_t tag = (int64_t)((uint64_t)a >> CG3D_PROG_REGION_SHIFT);
    if (!tag) return a;
    if (tag >= CG3D_PROG_REGIONS) return a;
    const int64_t off = a - (tag << CG3D_PROG_REGION_SHIFT);
    int64_t lo = first[tag], hi = first[tag + 1];
    if (lo >= hi) return tag << CG3D_PROG_REGION_SHIFT;
    while (hi - lo > 1) {                                  /* last start <= off (the first block when off lies before all) */
        const int64_t mid = (lo + hi) >> 1;
        if (starts[mid] <= off) lo = mid; else hi = mid;
    }
    return (tag << CG3D_PROG_REGION_SHIFT) | starts[lo];
}
static cg3d_sched_blk *cg3d_sched_find(cg3d_sched_blk *tab, int64_t mask, int64_t key, int make) {
    uint64_t h = (uint64_t)key * 0x9E3779B97F4A7C15ull;
    int64_t i = (int64_t)(h >> 20) & mask;
    for (;;) {
        if (tab[i].key == key) return &tab[i];
        if (tab[i].key == 0) {
            if (!make) return NULL;
            tab[i].key = key;
            return &tab[i];
        }
        i = (i + 1) & mask;
    }
}
static int cg3d_program_schedule_impl(const int64_t *prog, int64_t n, const int64_t *starts, const int64_t *first,
                                      const int64_t *cuts, int32_t ncut, int64_t *out, int64_t cap, int64_t *index,
                                      int64_t *cut_index, int64_t *n_out, int64_t *n_events) {
    enum { S = CG3D_PROG_STRIDE, ML = CG3D_PROG_MAX_LANES };
    if (n < 0 || ncut < 0 || (n > 0 && (!prog || !out || !index)) || (ncut > 0 && (!cuts || !cut_index)) || !first || !n_out || !n_events)
        return CG3D_ERR_ARG;
    int NL = 1;
    for (int64_t i = 0; i < n; i++) {
        const int l = (int)((prog[i * S] >> CG3D_PROG_LANE_SHIFT) & 0xff);
        if (l >= ML) return CG3D_ERR_ARG;
        if (l + 1 > NL) NL = l + 1;
    }
    for (int k = 0; k < ncut; k++)
        if (cuts[k] < 0 || cuts[k] > n || (k > 0 && cuts[k] < cuts[k - 1])) return CG3D_ERR_ARG;      /* ascending */
    if (NL == 1) {                                         /* one lane: nothing to order */
        if (cap < n) return CG3D_ERR_ARG;
        if (n) memcpy(out, prog, (size_t)n * S * sizeof(int64_t));
        for (int64_t i = 0; i < n; i++) index[i] = i;
        for (int k = 0; k < ncut; k++) cut_index[k] = cuts[k];
        *n_out = n;
        *n_events = 0;
        return CG3D_OK;
    }
    int64_t hcap = 64, nptr = 0;                          /* open addressing: at most half full */
    for (int64_t i = 0; i < n; i++)
        for (int c = 1; c < S; c++) nptr += prog[i * S + c] != 0;
    while (hcap < 2 * nptr + 64) hcap <<= 1;
    const int64_t max_edges = n * ML + 2 * (ncut + 2) * ML;
    cg3d_sched_blk *tab = (cg3d_sched_blk *)malloc((size_t)hcap * sizeof(cg3d_sched_blk));
    cg3d_sched_edge *edge = (cg3d_sched_edge *)malloc((size_t)max_edges * sizeof(cg3d_sched_edge));
    /* per row: first / last edge whose RECORD goes behind it, whose WAIT goes in front of it; per lane and sequence number: row */
    int32_t *lists = (int32_t *)malloc((size_t)(n + 1) * 4 * sizeof(int32_t));
    int64_t *where = (int64_t *)malloc((size_t)(n + 1) * ML * sizeof(int64_t));
    int64_t *queue = (int64_t *)malloc((size_t)(n + 2 * max_edges + 8) * sizeof(int64_t));       /* items of a part, lane after lane */
    if (!tab || !edge || !lists || !where || !queue) {
        free(tab); free(edge); free(lists); free(where); free(queue);
        return CG3D_ERR_LAUNCH;
    }
    int32_t *a_head = lists, *a_tail = lists + (n + 1), *b_head = lists + 2 * (n + 1), *b_tail = lists + 3 * (n + 1);
    for (int64_t i = 0; i < 4 * (n + 1); i++) lists[i] = -1;
    int32_t nedge = 0, nev = 0;
    int64_t no = 0;
    int rc = CG3D_OK;
#define CG3D_PUT(row_ptr)                                                                   \
    do {                                                                                    \
        if (no >= cap) { rc = CG3D_ERR_ARG; goto done; }                                    \
        memcpy(out + no * S, (row_ptr), S * sizeof(int64_t));                               \
        no++;                                                                               \
    } while (0)
#define CG3D_PUT_EV(opc, lane, slot)                                                        \
    do {                                                                                    \
        int64_t evrow[S];                                                                   \
        memset(evrow, 0, sizeof(evrow));                                                    \
        evrow[0] = (int64_t)(opc) | ((int64_t)(lane) << CG3D_PROG_LANE_SHIFT);              \
        evrow[1] = (slot);                                                                  \
        evrow[2] = 1;                                                                       \
        CG3D_PUT(evrow);                                                                    \
    } while (0)
    int part = 0;
    int64_t lo = 0;
    for (;;) {
        /* the part [lo, hi) */
        while (part < ncut && cuts[part] <= lo) {           /* cuts at the start of the part (incl. duplicates, 0) */
            cut_index[part] = no;
            part++;
        }
        int64_t hi = part < ncut ? cuts[part] : n;
        if (hi > n) hi = n;
        memset(tab, 0, (size_t)hcap * sizeof(cg3d_sched_blk));
        int64_t cnt[ML];
        int64_t seen[ML][ML];
        int forked[ML], fork_ev = -1;
        const int32_t edge0 = nedge;
        for (int l = 0; l < ML; l++) {
            cnt[l] = 0;
            forked[l] = l == 0;
            for (int m = 0; m < ML; m++) seen[l][m] = 0;
        }
        for (int64_t i = lo; i < hi; i++) {
            const int64_t *row = prog + i * S;
            const int L = (int)((row[0] >> CG3D_PROG_LANE_SHIFT) & 0xff);
            uint32_t rd = 0, wr = 0;
            if (cg3d_program_roles_impl((int32_t)(row[0] & 0xffffffffll), &rd, &wr) != CG3D_OK) { rc = CG3D_ERR_ARG; goto done; }
            int64_t need[ML];
            for (int m = 0; m < ML; m++) need[m] = 0;
            cg3d_sched_blk *br[S], *bw[S];
            int nr = 0, nw = 0;
            for (int c = 1; c < S; c++) {
                if (!((rd | wr) >> c & 1u) || !row[c]) continue;
                const int64_t key = cg3d_sched_block_of(row[c], starts, first);
                cg3d_sched_blk *b = cg3d_sched_find(tab, hcap - 1, key ? key : 1, 1);
                if ((wr >> c) & 1u) {
                    bw[nw++] = b;
                    for (int m = 0; m < NL; m++) {
                        if (b->w[m] > need[m]) need[m] = b->w[m];
                        if (b->r[m] > need[m]) need[m] = b->r[m];
                    }
                } else {
                    br[nr++] = b;
                    for (int m = 0; m < NL; m++)
                        if (b->w[m] > need[m]) need[m] = b->w[m];
                }
            }
            if (!forked[L]) {
                /* the lane's first row of the part: behind everything lane 0 was given before the part (and the zero-fill) */
                if (fork_ev < 0) fork_ev = nev++;
                cg3d_sched_edge *e = &edge[nedge];
                e->after_row = -1; e->before_row = i; e->rec_lane = 0; e->wait_lane = L; e->ev = fork_ev;
                e->next_after = e->next_before = -1;
                if (b_tail[i] < 0) b_head[i] = nedge; else edge[b_tail[i]].next_before = nedge;
                b_tail[i] = nedge++;
                forked[L] = 1;
            }
            for (int m = 0; m < NL; m++) {
                if (m == L || need[m] <= seen[L][m]) continue;
                const int64_t w = where[need[m] * ML + m];
                cg3d_sched_edge *e = &edge[nedge];
                e->after_row = w; e->before_row = i; e->rec_lane = m; e->wait_lane = L; e->ev = nev++;
                e->next_after = e->next_before = -1;
                if (a_tail[w] < 0) a_head[w] = nedge; else edge[a_tail[w]].next_after = nedge;
                a_tail[w] = nedge;
                if (b_tail[i] < 0) b_head[i] = nedge; else edge[b_tail[i]].next_before = nedge;
                b_tail[i] = nedge++;
                seen[L][m] = need[m];
            }
            cnt[L]++;
            where[cnt[L] * ML + L] = i;
            for (int k = 0; k < nr; k++) br[k]->r[L] = (int32_t)cnt[L];
            for (int k = 0; k < nw; k++) { bw[k]->w[L] = (int32_t)cnt[L]; bw[k]->r[L] = (int32_t)cnt[L]; }
        }
        /* the join: lane 0 behind the last row of every other lane */
        int32_t join_ev[ML];
        for (int m = 1; m < NL; m++) {
            join_ev[m] = -1;
            if (cnt[m] > seen[0][m]) {
                const int64_t w = where[cnt[m] * ML + m];
                cg3d_sched_edge *e = &edge[nedge];
                e->after_row = w; e->before_row = -2; e->rec_lane = m; e->wait_lane = 0; e->ev = nev++;
                e->next_after = e->next_before = -1;
                if (a_tail[w] < 0) a_head[w] = nedge; else edge[a_tail[w]].next_after = nedge;
                a_tail[w] = nedge++;
                join_ev[m] = e->ev;
            }
        }
        /* issue order: the lanes' sequences merged round robin.  An item is a row (>= 0), a RECORD (-1 - 2 e) or a WAIT (-2 - 2 e)
         * of edge e; the fork RECORD (edge -1 -> coded with e = max_edges) leads lane 0 */
        {
            int64_t qlen[ML], qpos[ML], *q[ML];
            int64_t total = 0;
            for (int l = 0; l < NL; l++) qlen[l] = 0;
            /* count */
            if (fork_ev >= 0) qlen[0]++;
            for (int64_t i = lo; i < hi; i++) {
                for (int32_t e = b_head[i]; e >= 0; e = edge[e].next_before) qlen[edge[e].wait_lane]++;
                qlen[(prog[i * S] >> CG3D_PROG_LANE_SHIFT) & 0xff]++;
                for (int32_t e = a_head[i]; e >= 0; e = edge[e].next_after) qlen[edge[e].rec_lane]++;
            }
            for (int l = 0; l < NL; l++) { q[l] = queue + total; total += qlen[l]; qpos[l] = 0; qlen[l] = 0; }
            if (fork_ev >= 0) q[0][qlen[0]++] = -1 - 2 * (int64_t)max_edges;
            for (int64_t i = lo; i < hi; i++) {
                for (int32_t e = b_head[i]; e >= 0; e = edge[e].next_before) q[edge[e].wait_lane][qlen[edge[e].wait_lane]++] = -2 - 2 * (int64_t)e;
                const int l = (int)((prog[i * S] >> CG3D_PROG_LANE_SHIFT) & 0xff);
                q[l][qlen[l]++] = i;
                for (int32_t e = a_head[i]; e >= 0; e = edge[e].next_after) q[edge[e].rec_lane][qlen[edge[e].rec_lane]++] = -1 - 2 * (int64_t)e;
            }
            /* placed[ev]: the RECORD of event ev has been put (reuse `where` is not possible: keep a byte map in `lists`' spare? no:
             * events of this part are edge0 .. nedge - 1 plus the fork: one flag per edge, in the edges themselves) */
            for (int32_t e = edge0; e < nedge; e++) edge[e].next_before = 0;      /* from here on: 1 = its RECORD is placed */
            int fork_placed = 0;
            int64_t left = total;
            while (left) {
                int moved = 0;
                for (int l = 0; l < NL; l++) {
                    if (qpos[l] >= qlen[l]) continue;
                    const int64_t it = q[l][qpos[l]];
                    if (it >= 0) {
                        index[it] = no;
                        CG3D_PUT(prog + it * S);
                    } else if ((-it) & 1) {                /* RECORD */
                        const int64_t e = (-1 - it) / 2;
                        if (e == max_edges) { fork_placed = 1; CG3D_PUT_EV(CG3D_OP_EVENT_RECORD, 0, fork_ev); }
                        else { edge[e].next_before = 1; CG3D_PUT_EV(CG3D_OP_EVENT_RECORD, edge[e].rec_lane, edge[e].ev); }
                    } else {                               /* WAIT */
                        const int64_t e = (-2 - it) / 2;
                        const int ready = edge[e].after_row == -1 ? fork_placed : edge[e].next_before;
                        if (!ready) continue;
                        CG3D_PUT_EV(CG3D_OP_EVENT_WAIT, edge[e].wait_lane, edge[e].ev);
                    }
                    qpos[l]++;
                    left--;
                    moved = 1;
                }
                if (!moved) { rc = CG3D_ERR_ARG; goto done; }      /* (cannot happen: emission order is a witness) */
            }
        }
        for (int m = 1; m < NL; m++)
            if (join_ev[m] >= 0) CG3D_PUT_EV(CG3D_OP_EVENT_WAIT, 0, join_ev[m]);
        if (hi >= n) break;
        lo = hi;
    }
    while (part < ncut) cut_index[part++] = no;
    *n_out = no;
    *n_events = nev;
done:
#undef CG3D_PUT
#undef CG3D_PUT_EV
    free(tab); free(edge); free(lists); free(where); free(queue);
    return rc;
}
#undef CG3D_B

/* ---- cg3d_host_segments -------------------------------------------------------------------------------------------------- */
static void cg3d_seg_msort(int64_t *idx, int64_t *tmp, const double *key, int64_t n) {       /* stable, by key */
    for (int64_t w = 1; w < n; w <<= 1) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            const int64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = key[idx[b]] < key[idx[a]] ? idx[b++] : idx[a++];
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
}
static int cg3d_host_segments_impl(const int64_t *off, int32_t K, int32_t G, int64_t maxlen, int32_t xcd_order,
                                   const int64_t *row_bounds, int64_t n_rows, int32_t *out, int64_t cap, int64_t *nseg) {
    if (!off || !out || !nseg || K < 1 || G < 1 || maxlen < 1 || cap < 0) return CG3D_ERR_ARG;
    const int64_t nslot = (int64_t)K * G;
    int64_t n = 0;
    for (int64_t s = 0; s < nslot; s++) {
        const int64_t c = off[s + 1] - off[s];
        if (c < 0) return CG3D_ERR_ARG;
        n += (c + maxlen - 1) / maxlen;
    }
    if (n > cap) return CG3D_ERR_ARG;
    *nseg = n;
    const int reorder = xcd_order && n >= 64 && n <= 4096 && maxlen >= 256;
    int32_t *tab = out;
    double *pos = NULL;
    int64_t *idx = NULL, *tmp = NULL, *place = NULL;
    int32_t *plain = NULL;
    if (reorder) {
        const int64_t run = (n + 7) / 8;
        pos = (double *)malloc((size_t)n * sizeof(double));
        idx = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        tmp = (int64_t *)malloc((size_t)n * sizeof(int64_t));
        place = (int64_t *)malloc((size_t)run * 8 * sizeof(int64_t));
        plain = (int32_t *)malloc((size_t)n * 3 * sizeof(int32_t));
        if (!pos || !idx || !tmp || !place || !plain) { free(pos); free(idx); free(tmp); free(place); free(plain); return CG3D_ERR_LAUNCH; }
        tab = plain;
    }
    int64_t i = 0;
    for (int64_t s = 0; s < nslot; s++) {
        const int64_t c = off[s + 1] - off[s], k = s / G, g = s % G;
        for (int64_t start = off[s]; start < off[s + 1]; start += maxlen, i++) {
            const int64_t cnt = off[s + 1] - start < maxlen ? off[s + 1] - start : maxlen;
            tab[3 * i] = (int32_t)(g * K + k);
            tab[3 * i + 1] = (int32_t)start;
            tab[3 * i + 2] = (int32_t)cnt;
            if (reorder) {
                double p = ((double)(start - off[s]) + 0.5 * (double)cnt) / (double)(c > 1 ? c : 1);
                if (row_bounds) p = ((double)row_bounds[g] + p * ((double)row_bounds[g + 1] - (double)row_bounds[g])) / (n_rows > 1 ? (double)n_rows : 1.0);
                pos[i] = p;
                idx[i] = i;
            }
        }
    }
    if (reorder) {
        const int64_t run = (n + 7) / 8;
        cg3d_seg_msort(idx, tmp, pos, n);
        for (int64_t q = 0; q < run * 8; q++) place[q] = -1;
        for (int64_t r = 0; r < n; r++) place[(r % run) * 8 + r / run] = idx[r];      /* rank r -> run r / run, place r % run in it */
        int64_t o = 0;
        for (int64_t q = 0; q < run * 8; q++)
            if (place[q] >= 0) { memcpy(out + 3 * o, plain + 3 * place[q], 3 * sizeof(int32_t)); o++; }
        free(pos); free(idx); free(tmp); free(place); free(plain);
    }
    return CG3D_OK;
}

/* ---- cg3d_host_bn_chunks ------------------------------------------------------------------------------------------------- */
static int64_t cg3d_bn_chunk_rows(const int64_t *b, int32_t G, int which, int64_t step_rows, int64_t red_min_rows, int64_t red_chunks,
                                  int32_t *rows, int32_t *gco) {
    int64_t n = 0;
    for (int32_t g = 0; g < G; g++) {
        const int64_t ng = b[g + 1] - b[g];
        int64_t step = step_rows;
        if (which == 0) {
            const int64_t per = (ng + red_chunks - 1) / red_chunks;      /* ceil; numpy: -(-ng // red_chunks) */
            step = red_min_rows > per ? red_min_rows : per;
        }
        if (step < 1) step = 1;
        const int64_t nch = (ng + step - 1) / step;
        if (gco) gco[g] = (int32_t)n;
        if (rows)
            for (int64_t j = 0; j < nch; j++) {
                const int64_t r0 = b[g] + j * step;
                rows[3 * (n + j)] = g;
                rows[3 * (n + j) + 1] = (int32_t)r0;
                rows[3 * (n + j) + 2] = (int32_t)(step < b[g + 1] - r0 ? step : b[g + 1] - r0);
            }
        n += nch;
    }
    if (gco) gco[G] = (int32_t)n;
    return n;
}
static int cg3d_host_bn_chunks_impl(const int64_t *bounds, int32_t G, int64_t step_rows, int64_t red_min_rows, int64_t red_chunks,
                                    int32_t *flat, int64_t cap, int64_t *offs, int64_t *sizes, int64_t *nred, int64_t *napp, int64_t *total) {
    if (!bounds || G < 1 || step_rows < 1 || red_chunks < 1 || !flat || !offs || !sizes || !nred || !napp || !total) return CG3D_ERR_ARG;
    for (int32_t g = 0; g < G; g++)
        if (bounds[g + 1] < bounds[g]) return CG3D_ERR_ARG;
    const int64_t nr = cg3d_bn_chunk_rows(bounds, G, 0, step_rows, red_min_rows, red_chunks, NULL, NULL);
    const int64_t na = cg3d_bn_chunk_rows(bounds, G, 1, step_rows, red_min_rows, red_chunks, NULL, NULL);
    sizes[0] = 3 * (nr > 0 ? nr : 1); sizes[1] = 3 * (na > 0 ? na : 1); sizes[2] = G + 1; sizes[3] = G; sizes[4] = G;
    int64_t tot = 0;
    for (int q = 0; q < 5; q++) { offs[q] = tot; tot += (sizes[q] + 3) & ~(int64_t)3; }
    if (tot > cap) return CG3D_ERR_ARG;
    memset(flat, 0, (size_t)tot * sizeof(int32_t));
    cg3d_bn_chunk_rows(bounds, G, 0, step_rows, red_min_rows, red_chunks, flat + offs[0], flat + offs[2]);
    cg3d_bn_chunk_rows(bounds, G, 1, step_rows, red_min_rows, red_chunks, flat + offs[1], NULL);
    for (int32_t g = 0; g < G; g++) {
        const int64_t ng = bounds[g + 1] - bounds[g];
        const double ns = (double)(ng > 1 ? ng : 1), d = ns - 1.0 > 1.0 ? ns - 1.0 : 1.0;
        const float a = (float)ns, u = (float)(ns / d);
        memcpy(flat + offs[3] + g, &a, sizeof(float));
        memcpy(flat + offs[4] + g, &u, sizeof(float));
    }
    *nred = nr; *napp = na; *total = tot;
    return CG3D_OK;
}
static int cg3d_program_run_lanes(const int64_t *prog, int64_t nops, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at) {
    if (nops < 0 || (nops > 0 && !prog) || !streams || nstreams < 1 || nstreams > CG3D_PROG_MAX_LANES) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < nops; i++) {
        const int64_t *row = prog + i * CG3D_PROG_STRIDE;
        const int lane = (int)((row[0] >> CG3D_PROG_LANE_SHIFT) & 0xff);
        const int rc = cg3d_program_dispatch(row, streams[lane < nstreams ? lane : 0]);
        if (rc != CG3D_OK) {
            if (fail_at) *fail_at = i;
            return rc;
        }
    }
    return CG3D_OK;
}
static int cg3d_program_run_bound(const int64_t *prog, int64_t nops, const int64_t *bases, const int64_t *events, int64_t nevents,
                                  void *zero_ptr, int64_t zero_bytes, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at) {
    if (nops < 0 || (nops > 0 && !prog) || !bases || !streams || nstreams < 1 || nstreams > CG3D_PROG_MAX_LANES || nevents < 0 ||
        (nevents > 0 && !events) || zero_bytes < 0)
        return CG3D_ERR_ARG;
    if (zero_bytes > 0) {
        const int rc = CG3D_PROG_MEMSET(zero_ptr, 0, zero_bytes, streams[0]);
        if (rc != CG3D_OK) {
            if (fail_at) *fail_at = -1;
            return rc;
        }
    }
    for (int64_t i = 0; i < nops; i++) {
        int64_t row[CG3D_PROG_STRIDE];
        memcpy(row, prog + i * CG3D_PROG_STRIDE, sizeof(row));
        for (int c = 1; c < CG3D_PROG_STRIDE; c++) {
            const uint64_t v = (uint64_t)row[c], tag = v >> CG3D_PROG_REGION_SHIFT;
            if (tag && tag < CG3D_PROG_REGIONS) row[c] = bases[tag] + (int64_t)(v & ((1ull << CG3D_PROG_REGION_SHIFT) - 1));
        }
        const int op = (int)(row[0] & 0xffffffffll);
        if ((op == CG3D_OP_EVENT_RECORD || op == CG3D_OP_EVENT_WAIT) && row[2] == 1) {
            if (row[1] < 0 || row[1] >= nevents) {
                if (fail_at) *fail_at = i;
                return CG3D_ERR_ARG;
            }
            row[1] = events[row[1]];
            row[2] = 0;
        }
        const int lane = (int)((row[0] >> CG3D_PROG_LANE_SHIFT) & 0xff);
        const int rc = cg3d_program_dispatch(row, streams[lane < nstreams ? lane : 0]);
        if (rc != CG3D_OK) {
            if (fail_at) *fail_at = i;
            return rc;
        }
    }
    return CG3D_OK;
}
#undef CG3D_A_P
#undef CG3D_A_I
#undef CG3D_A_L
#undef CG3D_A_F
#endif /* CG3D_PROGRAM_IMPL */

#endif /* CAGROUP3D_PROGRAM_H */
