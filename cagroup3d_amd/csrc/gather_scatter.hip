// gather_scatter.hip -- HBM-bound row gathers/scatters for gfx950: trilinear feature
// interpolation (SparseTensor.features_at_coordinates, reference call sites
// pcdet/models/backbones_3d/biresnet.py:182-197,376,389,394) and the "scatter mean" behind
// MinkowskiAvgPooling (biresnet.py:109-127) and ME.SparseTensor(UNWEIGHTED_AVERAGE)
// (pcdet/models/dense_heads/cagroup_head.py:257-271).
// One thread owns 4 consecutive channels of one row (16-byte accesses, rows contiguous across the
// wave); index/weight rows are wave-broadcast reads.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include "cg3d_common.h"

// ---------------------------------------------------------------- interpolation
template <bool VEC>
__global__ void k_interp_fwd(const float *__restrict__ F, const int32_t *__restrict__ idx,
                             const float *__restrict__ w, float *__restrict__ out, int64_t nq, int32_t c,
                             int32_t cq /* channel groups per row */) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t i = t / cq;
    int g = (int)(t % cq);
    if (i >= nq) return;
    if (VEC) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int32_t r = idx[i * 8 + j];
            if (r < 0) continue;
            float wt = w[i * 8 + j];
            float4 f = reinterpret_cast<const float4 *>(F + (int64_t)r * c)[g];
            acc.x += wt * f.x; acc.y += wt * f.y; acc.z += wt * f.z; acc.w += wt * f.w;
        }
        reinterpret_cast<float4 *>(out + i * c)[g] = acc;
    } else {
        float acc = 0.f;
        for (int j = 0; j < 8; j++) {
            int32_t r = idx[i * 8 + j];
            if (r < 0) continue;
            acc += w[i * 8 + j] * F[(int64_t)r * c + g];
        }
        out[i * c + g] = acc;
    }
}
extern "C" int cg3d_interp_fwd(const float *F, const int32_t *idx, const float *w, float *out, int64_t nq, int32_t c,
                               cg3d_stream_t stream) {
    if (nq < 0 || c < 1) return CG3D_ERR_ARG;
    if (nq == 0) return CG3D_OK;
    const bool vec = (c % 4 == 0) && !(((uintptr_t)F | (uintptr_t)out) & 15);
    const int32_t cq = vec ? c / 4 : c;
    const unsigned g = (unsigned)cg3d_divup(nq * cq, 256);
    if (vec) hipLaunchKernelGGL(k_interp_fwd<true>, dim3(g), dim3(256), 0, cg3d_hs(stream), F, idx, w, out, nq, c, cq);
    else hipLaunchKernelGGL(k_interp_fwd<false>, dim3(g), dim3(256), 0, cg3d_hs(stream), F, idx, w, out, nq, c, cq);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// A thread owns one channel of IB_RPT consecutive queries and adds runs of equal corner rows up in registers before they go
// out as atomics, corner position by corner position.  Rows are in (batch, Morton) order since round 2, so consecutive
// queries are the children of one coarse cell and share its 8 corners exactly (up to 8 queries per run at stride ratio 2):
// the kernel runs at the rate of the atomics themselves (84-168 M per launch = 2.2 TB/s of payload), and this divides
// their number.  (In arrival order the same merging measured no gain -- 141 vs 150 us -- there were no runs.)
#define IB_RPT 16
template <bool VEC>     // VEC: idx / w 16-byte aligned (torch allocations are); otherwise the same loads as scalars (offset views)
__global__ void k_interp_bwd(const float *__restrict__ dout, const int32_t *__restrict__ idx,
                             const float *__restrict__ w, float *__restrict__ dF, int64_t nq, int32_t c) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i0 = (t / c) * IB_RPT;
    const int a = (int)(t % c);
    if (i0 >= nq) return;
    const int n = (int)(nq - i0 < IB_RPT ? nq - i0 : IB_RPT);
    int32_t cur[8];
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { cur[j] = -1; acc[j] = 0.f; }
    for (int q = 0; q < n; q++) {
        const int64_t i = i0 + q;
        const float d = dout[i * c + a];
        int32_t rr[8];
        float ww[8];
        if (VEC) {
            const int4 r0 = *reinterpret_cast<const int4 *>(idx + i * 8), r1 = *reinterpret_cast<const int4 *>(idx + i * 8 + 4);
            const float4 w0 = *reinterpret_cast<const float4 *>(w + i * 8), w1 = *reinterpret_cast<const float4 *>(w + i * 8 + 4);
            rr[0] = r0.x; rr[1] = r0.y; rr[2] = r0.z; rr[3] = r0.w; rr[4] = r1.x; rr[5] = r1.y; rr[6] = r1.z; rr[7] = r1.w;
            ww[0] = w0.x; ww[1] = w0.y; ww[2] = w0.z; ww[3] = w0.w; ww[4] = w1.x; ww[5] = w1.y; ww[6] = w1.z; ww[7] = w1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) { rr[j] = idx[i * 8 + j]; ww[j] = w[i * 8 + j]; }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (rr[j] != cur[j]) {
                if (cur[j] >= 0) unsafeAtomicAdd(&dF[(int64_t)cur[j] * c + a], acc[j]);
                cur[j] = rr[j];
                acc[j] = 0.f;
            }
            acc[j] += ww[j] * d;            // (a missing corner, row -1, collects into an accumulator that is never written)
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (cur[j] >= 0) unsafeAtomicAdd(&dF[(int64_t)cur[j] * c + a], acc[j]);
}
extern "C" int cg3d_interp_bwd(const float *dout, const int32_t *idx, const float *w, float *dF, int64_t nq,
                               int32_t c, cg3d_stream_t stream) {
    if (nq < 0 || c < 1) return CG3D_ERR_ARG;
    if (nq == 0) return CG3D_OK;
    const dim3 grid((unsigned)cg3d_divup(cg3d_divup(nq, IB_RPT) * c, 256));
    if ((((uintptr_t)idx | (uintptr_t)w) & 15) == 0)
        hipLaunchKernelGGL(k_interp_bwd<true>, grid, dim3(256), 0, cg3d_hs(stream), dout, idx, w, dF, nq, c);
    else        // an offset view (idx[1:], w[1:]): no alignment precondition on this entry point
        hipLaunchKernelGGL(k_interp_bwd<false>, grid, dim3(256), 0, cg3d_hs(stream), dout, idx, w, dF, nq, c);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ---------------------------------------------------------------- scatter mean
// Thread = one channel of SS_RPT consecutive input rows; for every candidate position j the runs of equal output rows are
// summed in registers first (rows in Morton order: the children of one pooling / quantisation cell are consecutive and
// name the same outputs position by position), one atomic per run instead of one per (input row, output).
#define SS_RPT 8
__global__ void k_scatter_sum(const float *__restrict__ F, const int32_t *__restrict__ map, int32_t J,
                              float *__restrict__ out, float *__restrict__ cnt, int64_t n_in, int32_t c) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i0 = (t / c) * SS_RPT;
    const int a = (int)(t % c);
    if (i0 >= n_in) return;
    const int n = (int)(n_in - i0 < SS_RPT ? n_in - i0 : SS_RPT);
    float f[SS_RPT];
#pragma unroll
    for (int q = 0; q < SS_RPT; q++) f[q] = q < n ? F[(i0 + q) * c + a] : 0.f;
    for (int32_t j = 0; j < J; j++) {
        const int32_t *mj = map + (int64_t)j * n_in + i0;
        int32_t cur = -1;
        float acc = 0.f, k = 0.f;
#pragma unroll
        for (int q = 0; q < SS_RPT; q++) {
            const int32_t m = q < n ? mj[q] : -1;
            if (m != cur) {
                if (cur >= 0) {
                    unsafeAtomicAdd(&out[(int64_t)cur * c + a], acc);
                    if (a == 0) unsafeAtomicAdd(&cnt[cur], k);
                }
                cur = m; acc = 0.f; k = 0.f;
            }
            acc += f[q]; k += 1.f;
        }
        if (cur >= 0) {
            unsafeAtomicAdd(&out[(int64_t)cur * c + a], acc);
            if (a == 0) unsafeAtomicAdd(&cnt[cur], k);
        }
    }
}
__global__ void k_div_rows(float *__restrict__ out, const float *__restrict__ cnt, int64_t n_out, int32_t c) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n_out * c) return;
    float n = cnt[t / c];
    if (n > 0.f) out[t] = out[t] / n;
}
extern "C" int cg3d_scatter_mean_fwd(const float *F, const int32_t *map, int32_t J, float *out, float *cnt,
                                     int64_t n_in, int64_t n_out, int32_t c, cg3d_stream_t stream) {
    if (n_in < 0 || n_out < 0 || c < 1 || J < 1) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (n_out == 0) return CG3D_OK;
    if (hipMemsetAsync(out, 0, n_out * c * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (hipMemsetAsync(cnt, 0, n_out * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (n_in > 0)
        hipLaunchKernelGGL(k_scatter_sum, dim3((unsigned)cg3d_divup(cg3d_divup(n_in, SS_RPT) * c, 256)), dim3(256), 0, s, F, map, J,
                           out, cnt, n_in, c);
    hipLaunchKernelGGL(k_div_rows, dim3((unsigned)cg3d_divup(n_out * c, 256)), dim3(256), 0, s, out, cnt, n_out, c);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

__global__ void k_scatter_mean_bwd(const float *__restrict__ dout, const float *__restrict__ cnt,
                                   const int32_t *__restrict__ map, int32_t J, float *__restrict__ dF,
                                   int64_t n_in, int32_t c) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t i = t / c;
    int a = (int)(t % c);
    if (i >= n_in) return;
    float acc = 0.f;
    for (int32_t j = 0; j < J; j++) {
        int32_t m = map[(int64_t)j * n_in + i];
        if (m < 0) continue;
        acc += dout[(int64_t)m * c + a] / cnt[m];
    }
    dF[i * c + a] = acc;
}
extern "C" int cg3d_scatter_mean_bwd(const float *dout, const float *cnt, const int32_t *map, int32_t J, float *dF,
                                     int64_t n_in, int64_t n_out, int32_t c, cg3d_stream_t stream) {
    (void)n_out;
    if (n_in < 0 || c < 1 || J < 1) return CG3D_ERR_ARG;
    if (n_in == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_scatter_mean_bwd, dim3((unsigned)cg3d_divup(n_in * c, 256)), dim3(256), 0, cg3d_hs(stream),
                       dout, cnt, map, J, dF, n_in, c);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ---------------------------------------------------------------- row gather / scatter-add
template <bool VEC>
__global__ void k_gather_rows(const float *__restrict__ F, const int32_t *__restrict__ idx, float *__restrict__ out,
                              int64_t n, int32_t c, int32_t cq) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t i = t / cq;
    int g = (int)(t % cq);
    if (i >= n) return;
    const int64_t r = idx[i];
    if (VEC) reinterpret_cast<float4 *>(out + i * c)[g] = reinterpret_cast<const float4 *>(F + r * c)[g];
    else out[i * c + g] = F[r * c + g];
}
extern "C" int cg3d_gather_rows(const float *F, const int32_t *idx, float *out, int64_t n, int32_t c,
                                cg3d_stream_t stream) {
    if (n < 0 || c < 1) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    const bool vec = (c % 4 == 0) && !(((uintptr_t)F | (uintptr_t)out) & 15);
    const int32_t cq = vec ? c / 4 : c;
    const unsigned g = (unsigned)cg3d_divup(n * cq, 256);
    if (vec) hipLaunchKernelGGL(k_gather_rows<true>, dim3(g), dim3(256), 0, cg3d_hs(stream), F, idx, out, n, c, cq);
    else hipLaunchKernelGGL(k_gather_rows<false>, dim3(g), dim3(256), 0, cg3d_hs(stream), F, idx, out, n, c, cq);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
// A thread owns a channel group of RPT consecutive rows and adds runs of equal destination rows in registers before
// it touches memory: the RoI head's grid gather sends the 343 grid points of every degenerate (zero-padded) RoI to ONE
// voxel row (measured: 21 952 rows onto one destination, 830 us of same-address atomics per step) -- with the runs
// combined the kernel is back at its stream rate.
template <int RPT>
__global__ void k_scatter_add_rows(const float *__restrict__ dout, const int32_t *__restrict__ idx,
                                   float *__restrict__ dF, int64_t n, int32_t c) {
    // one channel per lane (a wave's atomics land on 64 consecutive floats: 2 cache lines per instruction; 16 bytes per
    // lane would spread one instruction over 8 lines and was 3x slower)
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t chunk = t / c;
    const int a = (int)(t - chunk * c);
    const int64_t r0 = chunk * RPT;
    if (r0 >= n) return;
    // every load of the chunk is issued before the first add
    int32_t id[RPT];
    float v[RPT];
#pragma unroll
    for (int u = 0; u < RPT; u++) {
        const int64_t r = r0 + u < n ? r0 + u : n - 1;
        id[u] = r0 + u < n ? idx[r] : -1;
        v[u] = dout[r * c + a];
    }
    int32_t prev = id[0];
    float acc = v[0];
#pragma unroll
    for (int u = 1; u <= RPT; u++) {
        const int32_t nid = u < RPT ? id[u] : -1;
        if (u < RPT && nid == prev) {
            acc += v[u];
        } else {
            if (prev >= 0) unsafeAtomicAdd(&dF[(int64_t)prev * c + a], acc);
            if (u < RPT) { acc = v[u]; prev = nid; }
        }
    }
}
extern "C" int cg3d_scatter_add_rows(const float *dout, const int32_t *idx, float *dF, int64_t n, int32_t c,
                                     cg3d_stream_t stream) {
    if (n < 0 || c < 1) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    constexpr int RPT = 16;     // measured on the RoI-head gather: 966 / 285 / 168 / 107 us at 1 / 4 / 8 / 16 rows per thread
    hipLaunchKernelGGL((k_scatter_add_rows<RPT>), dim3((unsigned)cg3d_divup(cg3d_divup(n, RPT) * c, 256)), dim3(256), 0,
                       cg3d_hs(stream), dout, idx, dF, n, c);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
