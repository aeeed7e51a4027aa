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
#define CG3D_PROG_MAX_LANES 4

/* Runs rows [0, nops) of `prog` in order on `stream`.  Returns CG3D_OK, or the status of the first failing call with its row
 * index in *fail_at (may be NULL); rows after it are not run.  An unknown opcode is CG3D_ERR_ARG. */
int cg3d_run_program(const int64_t *prog, int64_t nops, cg3d_stream_t stream, int64_t *fail_at);

/* The same with one stream per lane (1 <= nstreams <= CG3D_PROG_MAX_LANES).  Nothing is synchronised with the host; the caller's
 * table ends with the waits that bring every lane back to lane 0 (engine.py: _schedule). */
int cg3d_run_program_lanes(const int64_t *prog, int64_t nops, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at);

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
#undef CG3D_A_P
#undef CG3D_A_I
#undef CG3D_A_L
#undef CG3D_A_F
#endif /* CG3D_PROGRAM_IMPL */

#endif /* CAGROUP3D_PROGRAM_H */
