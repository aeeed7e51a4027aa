// bn_act.hip -- grouped BatchNorm (+ residual) (+ ReLU / ELU) for sparse-tensor feature matrices.
//
// Replaces the chains ME.MinkowskiBatchNorm -> MinkowskiReLU/ELU (-> `out += residual`) of
// pcdet/models/backbones_3d/biresnet.py:33-50,78-103 and dense_heads/cagroup_head.py:117-127, which the
// reference runs as 3-4 separate elementwise launches per site; with row groups all 18 class branches
// normalise in ONE launch.  HBM-bound: X is read twice in the forward (statistics, apply) and dY/X/Y once
// each per backward kernel; every access is a 16-byte vector per lane, a row (C*4 bytes) is covered by
// C/4 consecutive lanes, so each wave reads whole 128-byte lines.
// Statistics: fp32 partial sums per thread, fp64 across threads/workgroups (global fp64 atomics).
#include "cg3d_common.h"

#define BN_ROWS_PER_CHUNK 256   // rows a workgroup walks; the host builds the chunk table with this bound

__device__ static inline float act_fwd(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    return v;
}
// derivative expressed through the OUTPUT y (ELU: y <= 0 -> dy/dv = y + 1)
__device__ static inline float act_bwd(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// thread t of the block owns channel quad (t % cq) and walks rows (t / cq), +rpb, ...
template <bool BWD>
__global__ __launch_bounds__(256) void k_bn_reduce(const float *__restrict__ A, const float *__restrict__ X,
                                                   const float *__restrict__ Yv, const int32_t *__restrict__ chunks,
                                                   int32_t G, int32_t c, const float *__restrict__ mean,
                                                   const float *__restrict__ invstd, int act, double *__restrict__ sums) {
    __shared__ float4 red0[256], red1[256];
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const int tpr = cq < 256 ? cq : 256;          // threads per row
    const int rpb = 256 / tpr;                    // rows per block pass
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    for (int q = tq; q < cq; q += tpr) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        float4 mu = s0, is = s0;
        if (BWD) { mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q]; is = reinterpret_cast<const float4 *>(invstd + (int64_t)g * c)[q]; }
        if (tr < rpb) {
            for (int r = tr; r < nr; r += rpb) {
                const int64_t off = (int64_t)(r0 + r) * cq + q;
                if (!BWD) {
                    float4 v = reinterpret_cast<const float4 *>(A)[off];
                    s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                    s1.x += v.x * v.x; s1.y += v.y * v.y; s1.z += v.z * v.z; s1.w += v.w * v.w;
                } else {
                    float4 d = reinterpret_cast<const float4 *>(A)[off];
                    float4 x = reinterpret_cast<const float4 *>(X)[off];
                    if (act) {
                        float4 y = reinterpret_cast<const float4 *>(Yv)[off];
                        d.x *= act_bwd(y.x, act); d.y *= act_bwd(y.y, act); d.z *= act_bwd(y.z, act); d.w *= act_bwd(y.w, act);
                    }
                    s0.x += d.x; s0.y += d.y; s0.z += d.z; s0.w += d.w;
                    s1.x += d.x * (x.x - mu.x) * is.x; s1.y += d.y * (x.y - mu.y) * is.y;
                    s1.z += d.z * (x.z - mu.z) * is.z; s1.w += d.w * (x.w - mu.w) * is.w;
                }
            }
        }
        red0[threadIdx.x] = s0; red1[threadIdx.x] = s1;
        __syncthreads();
        if (tr == 0) {
            for (int j = 1; j < rpb; j++) {
                float4 a = red0[j * tpr + tq], b = red1[j * tpr + tq];
                s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
                s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            }
            double *d0 = sums + ((int64_t)g * c + q * 4);
            double *d1 = sums + ((int64_t)(G + g) * c + q * 4);
            atomicAdd(d0, (double)s0.x); atomicAdd(d0 + 1, (double)s0.y); atomicAdd(d0 + 2, (double)s0.z); atomicAdd(d0 + 3, (double)s0.w);
            atomicAdd(d1, (double)s1.x); atomicAdd(d1 + 1, (double)s1.y); atomicAdd(d1 + 2, (double)s1.z); atomicAdd(d1 + 3, (double)s1.w);
        }
        __syncthreads();
    }
}

extern "C" int cg3d_bn_stats(const float *X, const int32_t *chunks, int64_t nchunk, int32_t G, int32_t c, double *sums,
                             cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || ((uintptr_t)X & 15)) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * G * c, s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (nchunk == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_bn_reduce<false>, dim3((unsigned)nchunk), dim3(256), 0, s, X, nullptr, nullptr, chunks, G, c,
                       nullptr, nullptr, 0, sums);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_bn_bwd_reduce(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                                  int32_t G, int32_t c, const float *mean, const float *invstd, int32_t act,
                                  double *sums, cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || (((uintptr_t)X | (uintptr_t)dY | (uintptr_t)Y) & 15)) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * 2 * G * c, s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (nchunk == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_bn_reduce<true>, dim3((unsigned)nchunk), dim3(256), 0, s, dY, X, Y, chunks, G, c, mean, invstd,
                       act, sums);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ X, const float *__restrict__ R,
                                                  const int32_t *__restrict__ chunks, int32_t c,
                                                  const float *__restrict__ mean, const float *__restrict__ invstd,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, int act,
                                                  float *__restrict__ Y) {
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const int64_t total = (int64_t)nr * cq;
    for (int64_t t = threadIdx.x; t < total; t += 256) {
        const int q = (int)(t % cq);
        const int64_t off = (int64_t)r0 * cq + t;
        const float4 mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q];
        const float4 is = reinterpret_cast<const float4 *>(invstd + (int64_t)g * c)[q];
        const float4 ga = reinterpret_cast<const float4 *>(gamma + (int64_t)g * c)[q];
        const float4 be = reinterpret_cast<const float4 *>(beta + (int64_t)g * c)[q];
        float4 x = reinterpret_cast<const float4 *>(X)[off];
        float4 y;
        y.x = (x.x - mu.x) * is.x * ga.x + be.x; y.y = (x.y - mu.y) * is.y * ga.y + be.y;
        y.z = (x.z - mu.z) * is.z * ga.z + be.z; y.w = (x.w - mu.w) * is.w * ga.w + be.w;
        if (R) {
            float4 r = reinterpret_cast<const float4 *>(R)[off];
            y.x += r.x; y.y += r.y; y.z += r.z; y.w += r.w;
        }
        y.x = act_fwd(y.x, act); y.y = act_fwd(y.y, act); y.z = act_fwd(y.z, act); y.w = act_fwd(y.w, act);
        reinterpret_cast<float4 *>(Y)[off] = y;
    }
}
extern "C" int cg3d_bn_apply(const float *X, const float *residual, const int32_t *chunks, int64_t nchunk, int32_t c,
                             const float *mean, const float *invstd, const float *gamma, const float *beta, int32_t act,
                             float *Y, cg3d_stream_t stream) {
    if (nchunk < 0 || c < 4 || (c & 3) || (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)residual) & 15)) return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), X, residual, chunks, c, mean,
                       invstd, gamma, beta, act, Y);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ dY, const float *__restrict__ X,
                                                      const float *__restrict__ Yv, const int32_t *__restrict__ chunks,
                                                      int32_t c, const float *__restrict__ mean,
                                                      const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                      const double *__restrict__ sums, const float *__restrict__ group_n,
                                                      int32_t G, int act, int use_batch, float *__restrict__ dX,
                                                      float *__restrict__ dR) {
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const int64_t total = (int64_t)nr * cq;
    const float inv_n = use_batch ? 1.f / group_n[g] : 0.f;
    for (int64_t t = threadIdx.x; t < total; t += 256) {
        const int q = (int)(t % cq);
        const int64_t off = (int64_t)r0 * cq + t;
        const float4 mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q];
        const float4 is = reinterpret_cast<const float4 *>(invstd + (int64_t)g * c)[q];
        const float4 ga = reinterpret_cast<const float4 *>(gamma + (int64_t)g * c)[q];
        const double *s0 = sums + ((int64_t)g * c + q * 4), *s1 = sums + ((int64_t)(G + g) * c + q * 4);
        float4 d = reinterpret_cast<const float4 *>(dY)[off];
        float4 x = reinterpret_cast<const float4 *>(X)[off];
        if (act) {
            float4 y = reinterpret_cast<const float4 *>(Yv)[off];
            d.x *= act_bwd(y.x, act); d.y *= act_bwd(y.y, act); d.z *= act_bwd(y.z, act); d.w *= act_bwd(y.w, act);
        }
        if (dR) reinterpret_cast<float4 *>(dR)[off] = d;
        float4 o;
        o.x = ga.x * is.x * (d.x - ((float)s0[0] + (x.x - mu.x) * is.x * (float)s1[0]) * inv_n);
        o.y = ga.y * is.y * (d.y - ((float)s0[1] + (x.y - mu.y) * is.y * (float)s1[1]) * inv_n);
        o.z = ga.z * is.z * (d.z - ((float)s0[2] + (x.z - mu.z) * is.z * (float)s1[2]) * inv_n);
        o.w = ga.w * is.w * (d.w - ((float)s0[3] + (x.w - mu.w) * is.w * (float)s1[3]) * inv_n);
        reinterpret_cast<float4 *>(dX)[off] = o;
    }
}
extern "C" int cg3d_bn_bwd_apply(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                                 int32_t c, const float *mean, const float *invstd, const float *gamma, const double *sums,
                                 const float *group_n, int32_t G, int32_t act, int32_t use_batch_stats, float *dX,
                                 float *dRes, cg3d_stream_t stream) {
    if (nchunk < 0 || c < 4 || (c & 3) || (((uintptr_t)X | (uintptr_t)dY | (uintptr_t)Y | (uintptr_t)dX | (uintptr_t)dRes) & 15))
        return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), dY, X, Y, chunks, c, mean,
                       invstd, gamma, sums, group_n, G, act, use_batch_stats, dX, dRes);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
