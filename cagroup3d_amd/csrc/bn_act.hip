// bn_act.hip -- grouped BatchNorm (+ residual) (+ ReLU / ELU) for sparse-tensor feature matrices.
//
// Replaces the chains ME.MinkowskiBatchNorm -> MinkowskiReLU/ELU (-> `out += residual`) of
// pcdet/models/backbones_3d/biresnet.py:33-50,78-103 and dense_heads/cagroup_head.py:117-127, which the
// reference runs as 3-4 separate elementwise launches per site; with row groups all 18 class branches
// normalise in ONE launch.  HBM-bound: X is read twice in the forward (statistics, apply) and dY/X/Y once
// each per backward kernel; every access is a 16-byte vector per lane, a row (C*4 bytes) is covered by
// C/4 consecutive lanes, so each wave reads whole 128-byte lines.
// Statistics: fp32 partial sums per <=256-row chunk (plain stores, no atomics), then a finalise kernel
// reduces the chunks of every group in fp64.
#include "cg3d_common.h"

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

// thread t of the block owns channel quad (t % tpr) and walks rows (t / tpr), +rpb, ...
template <bool BWD>
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ A, const float *__restrict__ X,
                                                    const float *__restrict__ Yv, const int32_t *__restrict__ chunks,
                                                    int32_t c, const float *__restrict__ mean,
                                                    const float *__restrict__ var, float eps, int act,
                                                    float *__restrict__ ws) {
    __shared__ float4 red0[256], red1[256];
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const int tpr = cq < 256 ? cq : 256;          // threads per row
    const int rpb = 256 / tpr;                    // rows per block pass
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    float *w0 = ws + (int64_t)blockIdx.x * 2 * c, *w1 = w0 + c;
    for (int q = tq; q < cq; q += tpr) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, mu = s0, is = s0;
        if (BWD) {
            mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q];
            float4 v = reinterpret_cast<const float4 *>(var + (int64_t)g * c)[q];
            is = make_float4(rsqrtf(v.x + eps), rsqrtf(v.y + eps), rsqrtf(v.z + eps), rsqrtf(v.w + eps));
        }
        if (tr < rpb) {
            // four rows per trip, all loads issued before the first use: a thread walks only nr / rpb rows, so with
            // one load in flight the kernel was bound by memory LATENCY, not bandwidth
            for (int r = tr; r < nr; r += 4 * rpb) {
                float4 a[4], x[4], y[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int rr = r + u * rpb;
                    ok[u] = rr < nr;
                    const int64_t off = (int64_t)(r0 + (ok[u] ? rr : r)) * cq + q;      // clamped, unconditional load
                    a[u] = reinterpret_cast<const float4 *>(A)[off];
                    if (BWD) {
                        x[u] = reinterpret_cast<const float4 *>(X)[off];
                        if (act) y[u] = reinterpret_cast<const float4 *>(Yv)[off];
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (!ok[u]) continue;
                    if (!BWD) {
                        const float4 v = a[u];
                        s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                        s1.x += v.x * v.x; s1.y += v.y * v.y; s1.z += v.z * v.z; s1.w += v.w * v.w;
                    } else {
                        float4 d = a[u];
                        if (act) {
                            d.x *= act_bwd(y[u].x, act); d.y *= act_bwd(y[u].y, act);
                            d.z *= act_bwd(y[u].z, act); d.w *= act_bwd(y[u].w, act);
                        }
                        s0.x += d.x; s0.y += d.y; s0.z += d.z; s0.w += d.w;
                        s1.x += d.x * (x[u].x - mu.x) * is.x; s1.y += d.y * (x[u].y - mu.y) * is.y;
                        s1.z += d.z * (x[u].z - mu.z) * is.z; s1.w += d.w * (x[u].w - mu.w) * is.w;
                    }
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
            reinterpret_cast<float4 *>(w0)[q] = s0;
            reinterpret_cast<float4 *>(w1)[q] = s1;
        }
        __syncthreads();
    }
}

// fp64 reduction over the group's chunks: a 1024-thread block owns 16 channels of one group; its 64 chunk-lanes stride
// the group's chunks (four loads in flight each), then combine through LDS.  (Round 3: 64 lanes instead of 16 -- the tile
// convolution now leaves one partial per TILE, up to 1 500 of them, and with 16 lanes every thread walked ~100 chunks
// one L2 round trip after the other: 11 us per launch, 130 launches per step.)
// MODE 0: out0 = mean, out1 = biased variance (row count from the chunk table)
// MODE 1: out0 = sum0 (dbeta), out1 = sum1 (dgamma)
#define BN_FL 64
template <int MODE>
__global__ __launch_bounds__(16 * BN_FL) void k_bn_finalize(const float *__restrict__ ws, const int32_t *__restrict__ chunks,
                                                            const int32_t *__restrict__ gco, int32_t G, int32_t c,
                                                            float *__restrict__ out0, float *__restrict__ out1,
                                                            float *__restrict__ run0, float *__restrict__ run1,
                                                            long long *__restrict__ nbt, float momentum,
                                                            long long rows_arg, int nchunk_arg) {
    __shared__ double r0[16 * BN_FL], r1[16 * BN_FL];
    __shared__ long long rr[16 * BN_FL];
    const int cblocks = (c + 15) / 16;
    const int g = blockIdx.x / cblocks, a = (blockIdx.x % cblocks) * 16 + (threadIdx.x & 15), lanek = threadIdx.x >> 4;
    double s0 = 0.0, s1 = 0.0;
    long long rows = 0;
    if (a < c) {
        // four chunks per trip, every load issued before the first add
        // (no chunk table: one group, chunks [0, nchunk_arg), rows_arg rows -- partial sums produced by another kernel)
        int k = (gco ? gco[g] : 0) + lanek;
        const int kend = gco ? gco[g + 1] : nchunk_arg;
        for (; k + 3 * BN_FL < kend; k += 4 * BN_FL) {
            float p0[4], p1[4];
            int pr[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                p0[u] = ws[(int64_t)(k + BN_FL * u) * 2 * c + a];
                p1[u] = ws[(int64_t)(k + BN_FL * u) * 2 * c + c + a];
                pr[u] = (MODE == 0 && chunks) ? chunks[(k + BN_FL * u) * 3 + 2] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { s0 += (double)p0[u]; s1 += (double)p1[u]; rows += pr[u]; }
        }
        for (; k < kend; k += BN_FL) {
            s0 += (double)ws[(int64_t)k * 2 * c + a];
            s1 += (double)ws[(int64_t)k * 2 * c + c + a];
            if (MODE == 0 && chunks) rows += chunks[k * 3 + 2];
        }
    }
    r0[threadIdx.x] = s0; r1[threadIdx.x] = s1; rr[threadIdx.x] = rows;
    __syncthreads();
    // tree over the chunk lanes (64 -> 1), 16 channels side by side
    for (int half = BN_FL / 2; half >= 1; half >>= 1) {
        if (lanek < half) {
            r0[threadIdx.x] += r0[threadIdx.x + half * 16];
            r1[threadIdx.x] += r1[threadIdx.x + half * 16];
            rr[threadIdx.x] += rr[threadIdx.x + half * 16];
        }
        __syncthreads();
    }
    if (lanek == 0 && a < c) {
        s0 = r0[threadIdx.x]; s1 = r1[threadIdx.x]; rows = rr[threadIdx.x];
        const int64_t t = (int64_t)g * c + a;
        if (MODE == 0) {
            if (!chunks) rows = rows_arg;
            const double n = rows > 0 ? (double)rows : 1.0;
            const double m = s0 / n;
            double v = s1 / n - m * m;
            out0[t] = (float)m;
            out1[t] = (float)(v > 0.0 ? v : 0.0);
            if (run0 && run1) {     // running statistics, as nn.BatchNorm1d updates them (unbiased variance)
                const float unb = (float)(n / (n > 1.0 ? n - 1.0 : 1.0));
                run0[t] = (1.f - momentum) * run0[t] + momentum * (float)m;
                run1[t] = (1.f - momentum) * run1[t] + momentum * ((float)(v > 0.0 ? v : 0.0) * unb);
            }
            if (nbt && a == 0) nbt[g] += 1;
        } else {
            out0[t] = (float)s0;
            out1[t] = (float)s1;
        }
    }
}

static bool bad(const void *p) { return ((uintptr_t)p & 15) != 0; }

extern "C" int cg3d_bn_stats(const float *X, const int32_t *chunks, int64_t nchunk, const int32_t *group_chunk_off,
                             int32_t G, int32_t c, float *ws, float *mean, float *var, float *running_mean,
                             float *running_var, int64_t *num_batches_tracked, float momentum, cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || bad(X) || bad(ws)) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (nchunk > 0)
        hipLaunchKernelGGL(k_bn_partial<false>, dim3((unsigned)nchunk), dim3(256), 0, s, X, nullptr, nullptr, chunks, c,
                           nullptr, nullptr, 0.f, 0, ws);
    hipLaunchKernelGGL(k_bn_finalize<0>, dim3((unsigned)(G * ((c + 15) / 16))), dim3(16 * BN_FL), 0, s, ws, chunks,
                       group_chunk_off, G, c, mean, var, running_mean, running_var,
                       reinterpret_cast<long long *>(num_batches_tracked), momentum, 0ll, 0);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_bn_stats_from_partials(const float *ws, int64_t nchunk, int64_t rows, int32_t c, float *mean, float *var,
                                           float *running_mean, float *running_var, int64_t *num_batches_tracked,
                                           float momentum, cg3d_stream_t stream) {
    if (nchunk < 1 || nchunk > 0x7fffffffll || rows < 0 || c < 1 || !ws) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_bn_finalize<0>, dim3((unsigned)((c + 15) / 16)), dim3(16 * BN_FL), 0, cg3d_hs(stream), ws, nullptr, nullptr, 1, c,
                       mean, var, running_mean, running_var, reinterpret_cast<long long *>(num_batches_tracked), momentum,
                       (long long)rows, (int)nchunk);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_bn_bwd_reduce(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                                  const int32_t *group_chunk_off, int32_t G, int32_t c, const float *mean,
                                  const float *var, float eps, int32_t act, float *ws, float *dbeta, float *dgamma,
                                  cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || bad(X) || bad(dY) || bad(Y) || bad(ws)) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (nchunk > 0)
        hipLaunchKernelGGL(k_bn_partial<true>, dim3((unsigned)nchunk), dim3(256), 0, s, dY, X, Y, chunks, c, mean, var, eps,
                           act, ws);
    hipLaunchKernelGGL(k_bn_finalize<1>, dim3((unsigned)(G * ((c + 15) / 16))), dim3(16 * BN_FL), 0, s, ws, chunks,
                       group_chunk_off, G, c, dbeta, dgamma, nullptr, nullptr, nullptr, 0.f, 0ll, 0);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// four fp32 -> four bf16, round-to-nearest-even (v_cvt_pk_bf16_f32): the copy the next convolution gathers from
typedef float bn_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ static inline uint2 bn_pack4bf(float4 v) {
    const bn_f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(a, bn_bf16x2)),
                      __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bn_bf16x2)));
}

__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ X, const float *__restrict__ R,
                                                  const int32_t *__restrict__ chunks, int32_t c,
                                                  const float *__restrict__ mean, const float *__restrict__ var, float eps,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, int act,
                                                  float *__restrict__ Y, uint2 *__restrict__ Y16) {
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    // thread = (channel quad tq, row lane tr): the per-channel constants are loaded and inverted ONCE per thread, rows
    // are walked four at a time with all loads issued before the first use (memory-level parallelism)
    const int tpr = cq < 256 ? cq : 256, rpb = 256 / tpr;
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    if (tr >= rpb) return;
    for (int q = tq; q < cq; q += tpr) {
        const float4 mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q];
        const float4 vv = reinterpret_cast<const float4 *>(var + (int64_t)g * c)[q];
        const float4 ga = reinterpret_cast<const float4 *>(gamma + (int64_t)g * c)[q];
        const float4 be = reinterpret_cast<const float4 *>(beta + (int64_t)g * c)[q];
        const float4 is = make_float4(rsqrtf(vv.x + eps), rsqrtf(vv.y + eps), rsqrtf(vv.z + eps), rsqrtf(vv.w + eps));
        for (int r = tr; r < nr; r += 4 * rpb) {
            float4 xv[4], rv[4];
            int64_t off[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int rr = r + u * rpb;
                ok[u] = rr < nr;
                off[u] = (int64_t)(r0 + (ok[u] ? rr : r)) * cq + q;
                xv[u] = reinterpret_cast<const float4 *>(X)[off[u]];
                if (R) rv[u] = reinterpret_cast<const float4 *>(R)[off[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!ok[u]) continue;
                const float4 x = xv[u];
                float4 y;
                y.x = (x.x - mu.x) * is.x * ga.x + be.x; y.y = (x.y - mu.y) * is.y * ga.y + be.y;
                y.z = (x.z - mu.z) * is.z * ga.z + be.z; y.w = (x.w - mu.w) * is.w * ga.w + be.w;
                if (R) { y.x += rv[u].x; y.y += rv[u].y; y.z += rv[u].z; y.w += rv[u].w; }
                y.x = act_fwd(y.x, act); y.y = act_fwd(y.y, act); y.z = act_fwd(y.z, act); y.w = act_fwd(y.w, act);
                reinterpret_cast<float4 *>(Y)[off[u]] = y;
                if (Y16) Y16[off[u]] = bn_pack4bf(y);
            }
        }
    }
}
extern "C" int cg3d_bn_apply(const float *X, const float *residual, const int32_t *chunks, int64_t nchunk, int32_t c,
                             const float *mean, const float *var, float eps, const float *gamma, const float *beta,
                             int32_t act, float *Y, uint16_t *Y16, cg3d_stream_t stream) {
    if (nchunk < 0 || c < 4 || (c & 3) || bad(X) || bad(Y) || bad(residual) || ((uintptr_t)Y16 & 7)) return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_bn_apply, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), X, residual, chunks, c, mean, var,
                       eps, gamma, beta, act, Y, reinterpret_cast<uint2 *>(Y16));
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ dY, const float *__restrict__ X,
                                                      const float *__restrict__ Yv, const int32_t *__restrict__ chunks,
                                                      int32_t c, const float *__restrict__ mean,
                                                      const float *__restrict__ var, float eps,
                                                      const float *__restrict__ gamma, const float *__restrict__ dbeta,
                                                      const float *__restrict__ dgamma, const float *__restrict__ group_n,
                                                      int act, int use_batch, float *__restrict__ dX,
                                                      uint2 *__restrict__ dX16, float *__restrict__ dR) {
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const float inv_n = use_batch ? 1.f / group_n[g] : 0.f;
    const int tpr = cq < 256 ? cq : 256, rpb = 256 / tpr;
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    if (tr >= rpb) return;
    for (int q = tq; q < cq; q += tpr) {
        const float4 mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q];
        const float4 vv = reinterpret_cast<const float4 *>(var + (int64_t)g * c)[q];
        const float4 ga = reinterpret_cast<const float4 *>(gamma + (int64_t)g * c)[q];
        const float4 sb = reinterpret_cast<const float4 *>(dbeta + (int64_t)g * c)[q];
        const float4 sg = reinterpret_cast<const float4 *>(dgamma + (int64_t)g * c)[q];
        const float4 is = make_float4(rsqrtf(vv.x + eps), rsqrtf(vv.y + eps), rsqrtf(vv.z + eps), rsqrtf(vv.w + eps));
        for (int r = tr; r < nr; r += 4 * rpb) {
            float4 dv[4], xv[4], yv[4];
            int64_t off[4];
            bool ok[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int rr = r + u * rpb;
                ok[u] = rr < nr;
                off[u] = (int64_t)(r0 + (ok[u] ? rr : r)) * cq + q;
                dv[u] = reinterpret_cast<const float4 *>(dY)[off[u]];
                xv[u] = reinterpret_cast<const float4 *>(X)[off[u]];
                if (act) yv[u] = reinterpret_cast<const float4 *>(Yv)[off[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (!ok[u]) continue;
                float4 d = dv[u];
                const float4 x = xv[u];
                if (act) {
                    const float4 y = yv[u];
                    d.x *= act_bwd(y.x, act); d.y *= act_bwd(y.y, act); d.z *= act_bwd(y.z, act); d.w *= act_bwd(y.w, act);
                }
                if (dR) reinterpret_cast<float4 *>(dR)[off[u]] = d;
                float4 o;
                o.x = ga.x * is.x * (d.x - (sb.x + (x.x - mu.x) * is.x * sg.x) * inv_n);
                o.y = ga.y * is.y * (d.y - (sb.y + (x.y - mu.y) * is.y * sg.y) * inv_n);
                o.z = ga.z * is.z * (d.z - (sb.z + (x.z - mu.z) * is.z * sg.z) * inv_n);
                o.w = ga.w * is.w * (d.w - (sb.w + (x.w - mu.w) * is.w * sg.w) * inv_n);
                reinterpret_cast<float4 *>(dX)[off[u]] = o;
                if (dX16) dX16[off[u]] = bn_pack4bf(o);
            }
        }
    }
}
extern "C" int cg3d_bn_bwd_apply(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                                 int32_t c, const float *mean, const float *var, float eps, const float *gamma,
                                 const float *dbeta, const float *dgamma, const float *group_n, int32_t act,
                                 int32_t use_batch_stats, float *dX, uint16_t *dX16, float *dRes, cg3d_stream_t stream) {
    if (nchunk < 0 || c < 4 || (c & 3) || bad(X) || bad(dY) || bad(Y) || bad(dX) || bad(dRes) || ((uintptr_t)dX16 & 7))
        return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), dY, X, Y, chunks, c, mean, var,
                       eps, gamma, dbeta, dgamma, group_n, act, use_batch_stats, dX, reinterpret_cast<uint2 *>(dX16), dRes);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
