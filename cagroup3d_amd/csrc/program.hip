// program.hip -- cg3d_run_program / cg3d_run_program_lanes: a table of C-ABI calls issued back to back on one stream, or on one
// stream per lane with event edges between them (include/cagroup3d_program.h).
// The host language then pays for ONE foreign call per network pass instead of one per launch; see engine.py.
#include "cg3d_common.h"

static inline int prog_memset(void *dst, int value, int64_t nbytes, cg3d_stream_t s) {
    if (nbytes < 0 || (nbytes > 0 && !dst)) return CG3D_ERR_ARG;
    if (nbytes == 0) return CG3D_OK;
    return hipMemsetAsync(dst, value, (size_t)nbytes, cg3d_hs(s)) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
static inline int prog_copy2d(void *dst, int64_t dpitch, const void *src, int64_t spitch, int64_t width, int64_t height,
                              cg3d_stream_t s) {
    if (width < 0 || height < 0 || dpitch < width || spitch < width) return CG3D_ERR_ARG;
    if (width == 0 || height == 0) return CG3D_OK;
    if (!dst || !src) return CG3D_ERR_ARG;
    return hipMemcpy2DAsync(dst, (size_t)dpitch, src, (size_t)spitch, (size_t)width, (size_t)height, hipMemcpyDeviceToDevice,
                            cg3d_hs(s)) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
static inline int prog_event_record(int64_t handle, cg3d_stream_t s) {
    if (!handle) return CG3D_ERR_ARG;
    return hipEventRecord((hipEvent_t)(intptr_t)handle, cg3d_hs(s)) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
static inline int prog_event_wait(int64_t handle, cg3d_stream_t s) {
    if (!handle) return CG3D_ERR_ARG;
    return hipStreamWaitEvent(cg3d_hs(s), (hipEvent_t)(intptr_t)handle, 0) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
#define CG3D_PROG_MEMSET prog_memset
#define CG3D_PROG_COPY2D prog_copy2d
#define CG3D_PROG_EVENT_RECORD prog_event_record
#define CG3D_PROG_EVENT_WAIT prog_event_wait
#define CG3D_PROGRAM_IMPL
#include "../../include/cagroup3d_program.h"

extern "C" int cg3d_run_program(const int64_t *prog, int64_t nops, cg3d_stream_t stream, int64_t *fail_at) {
    return cg3d_program_run(prog, nops, stream, fail_at);
}
extern "C" int cg3d_run_program_lanes(const int64_t *prog, int64_t nops, const cg3d_stream_t *streams, int32_t nstreams,
                                      int64_t *fail_at) {
    return cg3d_program_run_lanes(prog, nops, streams, nstreams, fail_at);
}
extern "C" int cg3d_program_schedule(const int64_t *prog, int64_t n, const int64_t *starts, const int64_t *region_first,
                                     const int64_t *cuts, int32_t ncut, int64_t *out, int64_t cap, int64_t *index,
                                     int64_t *cut_index, int64_t *n_out, int64_t *n_events) {
    return cg3d_program_schedule_impl(prog, n, starts, region_first, cuts, ncut, out, cap, index, cut_index, n_out, n_events);
}
extern "C" int cg3d_host_segments(const int64_t *off, int32_t K, int32_t G, int64_t maxlen, int32_t xcd_order, const int64_t *row_bounds,
                                  int64_t n_rows, int32_t *out, int64_t cap, int64_t *nseg) {
    return cg3d_host_segments_impl(off, K, G, maxlen, xcd_order, row_bounds, n_rows, out, cap, nseg);
}
extern "C" int cg3d_host_bn_chunks(const int64_t *bounds, int32_t G, int64_t step_rows, int64_t red_min_rows, int64_t red_chunks, int32_t *flat,
                                   int64_t cap, int64_t *offs, int64_t *sizes, int64_t *nred, int64_t *napp, int64_t *total) {
    return cg3d_host_bn_chunks_impl(bounds, G, step_rows, red_min_rows, red_chunks, flat, cap, offs, sizes, nred, napp, total);
}
extern "C" int cg3d_run_program_bound(const int64_t *prog, int64_t nops, const int64_t *bases, const int64_t *events, int64_t nevents,
                                      void *zero_ptr, int64_t zero_bytes, const cg3d_stream_t *streams, int32_t nstreams, int64_t *fail_at) {
    return cg3d_program_run_bound(prog, nops, bases, events, nevents, zero_ptr, zero_bytes, streams, nstreams, fail_at);
}
extern "C" int cg3d_program_roles(int32_t opcode, uint32_t *rd, uint32_t *wr) { return cg3d_program_roles_impl(opcode, rd, wr); }
extern "C" int cg3d_event_create_sync(int64_t *handle) {
    if (!handle) return CG3D_ERR_ARG;
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return CG3D_ERR_LAUNCH;
    *handle = (int64_t)(intptr_t)e;
    return CG3D_OK;
}
extern "C" int cg3d_event_create(int64_t *handle) {
    if (!handle) return CG3D_ERR_ARG;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return CG3D_ERR_LAUNCH;
    *handle = (int64_t)(intptr_t)e;
    return CG3D_OK;
}
extern "C" int cg3d_event_destroy(int64_t handle) {
    if (!handle) return CG3D_ERR_ARG;
    return hipEventDestroy((hipEvent_t)(intptr_t)handle) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
extern "C" int cg3d_event_elapsed_ms(int64_t start, int64_t stop, float *ms) {
    if (!start || !stop || !ms) return CG3D_ERR_ARG;
    if (hipEventSynchronize((hipEvent_t)(intptr_t)stop) != hipSuccess) return CG3D_ERR_LAUNCH;
    return hipEventElapsedTime(ms, (hipEvent_t)(intptr_t)start, (hipEvent_t)(intptr_t)stop) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
