/* oracle_program.c -- cg3d_run_program on the CPU oracle (TEST INFRASTRUCTURE: the checker, never the product): the shared
 * dispatcher of include/cagroup3d_program.h over the oracle's own entry points, so that the host-side engine
 * (cagroup3d_amd/engine.py) can be run and checked on a GPU-less box.  Memset / strided copy are libc; events are dummies. */
#include <string.h>
#include "../include/cagroup3d_hip.h"

static int op_memset(void *dst, int value, int64_t nbytes, cg3d_stream_t s) {
    (void)s;
    if (nbytes < 0 || (nbytes > 0 && !dst)) return CG3D_ERR_ARG;
    if (nbytes > 0) memset(dst, value, (size_t)nbytes);
    return CG3D_OK;
}
static int op_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width, int64_t height, cg3d_stream_t s) {
    (void)s;
    if (width < 0 || height < 0 || dpitch < width || spitch < width) return CG3D_ERR_ARG;
    if (width == 0 || height == 0) return CG3D_OK;
    if (!dst || !src) return CG3D_ERR_ARG;
    for (int64_t r = 0; r < height; r++) memcpy((char *)dst + r * dpitch, (const char *)src + r * spitch, (size_t)width);
    return CG3D_OK;
}
static int op_event_record(int64_t handle, cg3d_stream_t s) {
    (void)s;
    return handle ? CG3D_OK : CG3D_ERR_ARG;
}
static int op_event_wait(int64_t handle, cg3d_stream_t s) {
    (void)s;
    return handle ? CG3D_OK : CG3D_ERR_ARG;
}
#define CG3D_PROG_MEMSET op_memset
#define CG3D_PROG_COPY2D op_copy2d
#define CG3D_PROG_EVENT_RECORD op_event_record
#define CG3D_PROG_EVENT_WAIT op_event_wait
#define CG3D_PROGRAM_IMPL
#include "../include/cagroup3d_program.h"

int cg3d_run_program(const int64_t *prog, int64_t nops, cg3d_stream_t stream, int64_t *fail_at) {
    return cg3d_program_run(prog, nops, stream, fail_at);
}
/* lanes: the oracle runs the rows in table order (always a valid sequential order); the streams are passed on and ignored */
int cg3d_run_program_lanes(const int64_t *prog, int64_t nops, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at) {
    return cg3d_program_run_lanes(prog, nops, streams, nstreams, fail_at);
}
int cg3d_program_schedule(const int64_t *prog, int64_t n, const int64_t *starts, const int64_t *region_first, const int64_t *cuts,
                          int32_t ncut, int64_t *out, int64_t cap, int64_t *index, int64_t *cut_index, int64_t *n_out,
                          int64_t *n_events) {
    return cg3d_program_schedule_impl(prog, n, starts, region_first, cuts, ncut, out, cap, index, cut_index, n_out, n_events);
}
int cg3d_host_segments(const int64_t *off, int32_t K, int32_t G, int64_t maxlen, int32_t xcd_order, const int64_t *row_bounds,
                       int64_t n_rows, int32_t *out, int64_t cap, int64_t *nseg) {
    return cg3d_host_segments_impl(off, K, G, maxlen, xcd_order, row_bounds, n_rows, out, cap, nseg);
}
int cg3d_host_bn_chunks(const int64_t *bounds, int32_t G, int64_t step_rows, int64_t red_min_rows, int64_t red_chunks, int32_t *flat,
                        int64_t cap, int64_t *offs, int64_t *sizes, int64_t *nred, int64_t *napp, int64_t *total) {
    return cg3d_host_bn_chunks_impl(bounds, G, step_rows, red_min_rows, red_chunks, flat, cap, offs, sizes, nred, napp, total);
}
int cg3d_run_program_bound(const int64_t *prog, int64_t nops, const int64_t *bases, const int64_t *events, int64_t nevents,
                           void *zero_ptr, int64_t zero_bytes, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at) {
    return cg3d_program_run_bound(prog, nops, bases, events, nevents, zero_ptr, zero_bytes, streams, nstreams, fail_at);
}
int cg3d_program_roles(int32_t opcode, uint32_t *rd, uint32_t *wr) { return cg3d_program_roles_impl(opcode, rd, wr); }
int cg3d_event_create(int64_t *handle) {
    static int64_t next = 1;
    if (!handle) return CG3D_ERR_ARG;
    *handle = next++;
    return CG3D_OK;
}
int cg3d_event_create_sync(int64_t *handle) { return cg3d_event_create(handle); }
int cg3d_event_destroy(int64_t handle) { return handle ? CG3D_OK : CG3D_ERR_ARG; }
int cg3d_event_elapsed_ms(int64_t start, int64_t stop, float *ms) {
    if (!start || !stop || !ms) return CG3D_ERR_ARG;
    *ms = 0.f;
    return CG3D_OK;
}
