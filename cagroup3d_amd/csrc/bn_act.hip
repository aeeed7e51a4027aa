// bn_act.hip -- grouped BatchNorm (+ residual) (+ ReLU / ELU) for sparse-tensor feature matrices.
//
// Replaces the chains ME.MinkowskiBatchNorm -> MinkowskiReLU/ELU (-> `out += residual`) of
// pcdet/models/backbones_3d/biresnet.py:33-50,78-103 and dense_heads/cagroup_head.py:117-127, which the
// reference runs as 3-4 separate elementwise launches per site; with row groups all 18 class branches
// normalise in ONE launch.  HBM-bound: X is read twice in the forward (statistics, apply) and dY/X/Y once
// each per backward kernel; every access is a 16-byte vector per lane, a row (C*4 bytes) is covered by
// C/4 consecutive lanes, so each wave reads whole 128-byte lines.
// Statistics (round 3): every workgroup adds its chunk's partial sums into ONE zero-filled fp32 table per layer,
// sums[CG3D_BN_SLOTS][2][G][C] (sum, sum of squares -- or sum dz, sum dz * xhat in the backward), with fp32 atomics into
// slot (workgroup index % CG3D_BN_SLOTS); the kernels that need mean / variance / dbeta / dgamma add the slots up and derive
// them themselves (in fp64, a few channels per thread).  The per-chunk partials + fp64 finalise kernel of rounds 1-2 cost
// 130 launches of ~10 us per step for a few hundred KB of work.  (One slot was measured first: the ~1 200 workgroups of a
// layer then queue on the same 2 C addresses at the L2 atomic unit -- 17 -> 70 us per launch.)
// Determinism: the atomic chains make the slot sums (hence mean / var / running statistics / dgamma / dbeta) differ in the
// last bits from run to run, like every atomic reduction of this library; the slots are added up in fp64.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
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
                                                    float *__restrict__ sums, int G) {
    __shared__ float4 red0[256], red1[256];
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const int tpr = cq < 256 ? cq : 256;          // threads per row
    const int rpb = 256 / tpr;                    // rows per block pass
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const int slot = blockIdx.x % CG3D_BN_SLOTS;
    float *w0 = sums + ((int64_t)(slot * 2) * G + g) * c, *w1 = sums + ((int64_t)(slot * 2 + 1) * G + g) * c;    // rows g of sums[slot][0] / [1]
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
            unsafeAtomicAdd(w0 + 4 * q, s0.x); unsafeAtomicAdd(w0 + 4 * q + 1, s0.y);
            unsafeAtomicAdd(w0 + 4 * q + 2, s0.z); unsafeAtomicAdd(w0 + 4 * q + 3, s0.w);
            unsafeAtomicAdd(w1 + 4 * q, s1.x); unsafeAtomicAdd(w1 + 4 * q + 1, s1.y);
            unsafeAtomicAdd(w1 + 4 * q + 2, s1.z); unsafeAtomicAdd(w1 + 4 * q + 3, s1.w);
        }
        __syncthreads();
    }
}

static bool bad(const void *p) { return ((uintptr_t)p & 15) != 0; }
// bf16 storage: a thread column per channel oct, power-of-two column counts (the reductions meet by xor-shuffles), <= 1024 channels
static bool bn16_ok(int32_t c) { return c >= 64 && c <= 1024 && !(c & (c - 1)); }
__global__ void k_bn16_partial_bwd(const uint16_t *__restrict__ dY, const uint16_t *__restrict__ X, const uint16_t *__restrict__ Yv,
                                   const int32_t *__restrict__ chunks, int32_t c, const float *__restrict__ mean,
                                   const float *__restrict__ var, float eps, int act, float *__restrict__ sums, int G);

// sums[slot][0][g][:] += sum over the rows of group g of x, sums[slot][1][g][:] += sum of x^2 (the caller zero-fills `sums`)
extern "C" int cg3d_bn_sums(const float *X, const int32_t *chunks, int64_t nchunk, int32_t G, int32_t c, float *sums,
                            cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || bad(X) || !sums) return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_bn_partial<false>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), X, nullptr, nullptr, chunks, c,
                       nullptr, nullptr, 0.f, 0, sums, G);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
// dsums[slot][0][g][:] += sum dz (dbeta), dsums[slot][1][g][:] += sum dz * xhat (dgamma); dz = dy * act'(y)
extern "C" int cg3d_bn_bwd_sums(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk, int32_t G,
                                int32_t c, const float *mean, const float *var, float eps, int32_t act, float *dsums,
                                cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || bad(X) || bad(dY) || bad(Y) || !dsums) return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    if (act & CG3D_BN_STORE_BF16) {          // dY, X, Y are bf16 rows
        if (!bn16_ok(c)) return CG3D_ERR_ARG;
        hipLaunchKernelGGL(k_bn16_partial_bwd, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), reinterpret_cast<const uint16_t *>(dY),
                           reinterpret_cast<const uint16_t *>(X), reinterpret_cast<const uint16_t *>(Y), chunks, c, mean, var, eps,
                           act & ~CG3D_BN_STORE_BF16, dsums, G);
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    hipLaunchKernelGGL(k_bn_partial<true>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), dY, X, Y, chunks, c, mean, var, eps,
                       act, dsums, G);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// The slot table of a layer, added up ONCE per workgroup.  Every thread used to add the 16 slots of its own channels up
// itself: 0.5-1 KB of table loads per thread, 128-256 KB per workgroup out of the L1 for an 8 KB table -- more bytes than the
// rows the workgroup then moves (a 128-row chunk of 64 channels is 16-32 KB), and 1 200 workgroups per launch do it.  Here
// thread t adds up channel t (t + 256, ...) of both statistics in fp64, in slot order as before (same bits), and leaves the two
// per-channel results in LDS; the row loop reads its channels' constants from there.
#define BN_LDS_C 1024
__device__ static inline void bn_slot_pair_to_lds(const float *__restrict__ table, int G, int g, int c, float *s0, float *s1,
                                                  bool write_out, float *__restrict__ out0, float *__restrict__ out1) {
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int sl = 0; sl < CG3D_BN_SLOTS; sl++) {
            a0 += table[((int64_t)(sl * 2) * G + g) * c + ch];
            a1 += table[((int64_t)(sl * 2 + 1) * G + g) * c + ch];
        }
        s0[ch] = (float)a0;
        s1[ch] = (float)a1;
        if (write_out) { out0[(int64_t)g * c + ch] = (float)a0; out1[(int64_t)g * c + ch] = (float)a1; }
    }
}
// forward: mean / biased variance of channel ch from the slot sums (fp64), written to LDS -- and, by the first chunk of a
// group, to mean / var and the running statistics
__device__ static inline void bn_mean_var_to_lds(const float *__restrict__ sums, const float *__restrict__ group_n, int G, int g, int c,
                                                 float *s_mu, float *s_var, bool first_of_group, float *__restrict__ mean,
                                                 float *__restrict__ var, float *__restrict__ run_mean, float *__restrict__ run_var,
                                                 long long *__restrict__ nbt, float momentum) {
    const double n = group_n[g] > 0.f ? (double)group_n[g] : 1.0;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int sl = 0; sl < CG3D_BN_SLOTS; sl++) {
            a0 += sums[((int64_t)(sl * 2) * G + g) * c + ch];
            a1 += sums[((int64_t)(sl * 2 + 1) * G + g) * c + ch];
        }
        const double m = a0 / n, v = a1 / n - m * m;
        const float mu = (float)m, vv = (float)(v > 0 ? v : 0);
        s_mu[ch] = mu;
        s_var[ch] = vv;
        if (first_of_group) {
            mean[(int64_t)g * c + ch] = mu;
            var[(int64_t)g * c + ch] = vv;
            if (run_mean && run_var) {
                const float unb = (float)(n / (n > 1.0 ? n - 1.0 : 1.0));
                float *rm = run_mean + (int64_t)g * c + ch, *rv = run_var + (int64_t)g * c + ch;
                *rm = (1.f - momentum) * *rm + momentum * mu;
                *rv = (1.f - momentum) * *rv + momentum * (vv * unb);
            }
            if (nbt && ch == 0) nbt[g] += 1;
        }
    }
}

// four fp32 -> four bf16, round-to-nearest-even (v_cvt_pk_bf16_f32): the copy the next convolution gathers from
typedef float bn_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bn_bf16x2 __attribute__((ext_vector_type(2)));
__device__ static inline uint2 bn_pack4bf(float4 v) {
    const bn_f32x2 a = {v.x, v.y}, b = {v.z, v.w};
    return make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(a, bn_bf16x2)),
                      __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bn_bf16x2)));
}

// ------------------------------------------------------------------------------------------------ bf16 storage
// CG3D_BN_STORE_BF16: every row matrix of the call is stored as bf16.  Same structure as the kernels above with EIGHT channels
// per thread (one 16-byte access = 8 bf16): a row of C channels is covered by C / 8 lanes, the per-channel constants live in
// registers, rows are walked four at a time with all loads issued before the first use.  Operands are widened to fp32 (a
// shift), everything is computed in fp32 as above, results are rounded to nearest even on the store (v_cvt_pk_bf16_f32).
struct bn_v8 { float v[8]; };
__device__ static inline bn_v8 bn_ld8(const uint16_t *base, int64_t off8) {
    const uint4 u = reinterpret_cast<const uint4 *>(base)[off8];
    bn_v8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
__device__ static inline bn_v8 bn_unpack8(const uint4 u) {
    bn_v8 r;
    r.v[0] = __uint_as_float(u.x << 16); r.v[1] = __uint_as_float(u.x & 0xffff0000u);
    r.v[2] = __uint_as_float(u.y << 16); r.v[3] = __uint_as_float(u.y & 0xffff0000u);
    r.v[4] = __uint_as_float(u.z << 16); r.v[5] = __uint_as_float(u.z & 0xffff0000u);
    r.v[6] = __uint_as_float(u.w << 16); r.v[7] = __uint_as_float(u.w & 0xffff0000u);
    return r;
}
__device__ static inline uint4 bn_pack8(const bn_v8 &a) {
    const uint2 lo = bn_pack4bf(make_float4(a.v[0], a.v[1], a.v[2], a.v[3])), hi = bn_pack4bf(make_float4(a.v[4], a.v[5], a.v[6], a.v[7]));
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
__device__ static inline void bn_st8(uint16_t *base, int64_t off8, const bn_v8 &a) {
    const uint2 lo = bn_pack4bf(make_float4(a.v[0], a.v[1], a.v[2], a.v[3])), hi = bn_pack4bf(make_float4(a.v[4], a.v[5], a.v[6], a.v[7]));
    reinterpret_cast<uint4 *>(base)[off8] = make_uint4(lo.x, lo.y, hi.x, hi.y);
}
__device__ static inline bn_v8 bn_ldf8(const float *p) {          // 8 consecutive per-channel constants
    const float4 a = reinterpret_cast<const float4 *>(p)[0], b = reinterpret_cast<const float4 *>(p)[1];
    bn_v8 r = {{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
    return r;
}
__device__ static inline void bn_stf8(float *p, const bn_v8 &a) {
    reinterpret_cast<float4 *>(p)[0] = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    reinterpret_cast<float4 *>(p)[1] = make_float4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
// sum over the CG3D_BN_SLOTS slots of one statistics row (fp64), 8 channels
__device__ static inline void bn_slot_sums8(const float *table, int G, int g, int c, int o8, int which, double out[8]) {
#pragma unroll
    for (int e = 0; e < 8; e++) out[e] = 0.0;
#pragma unroll
    for (int sl = 0; sl < CG3D_BN_SLOTS; sl++) {
        const bn_v8 u = bn_ldf8(table + ((int64_t)(sl * 2 + which) * G + g) * c + 8 * o8);
#pragma unroll
        for (int e = 0; e < 8; e++) out[e] += u.v[e];
    }
}

__global__ __launch_bounds__(256) void k_bn16_partial_bwd(const uint16_t *__restrict__ dY, const uint16_t *__restrict__ X,
                                                          const uint16_t *__restrict__ Yv, const int32_t *__restrict__ chunks,
                                                          int32_t c, const float *__restrict__ mean, const float *__restrict__ var,
                                                          float eps, int act, float *__restrict__ sums, int G) {
    // c / 8 <= 256 channel octs (the entry point checks): one oct per thread column, no loop over octs
    __shared__ float red[4][2][64][9];                   // [wave][statistic][oct of the wave][8 (+1: bank spread)]
    const int co = c >> 3;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const int tpr = co < 256 ? co : 256, rpb = 256 / tpr;          // tpr: a power of two (8 .. 256)
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.x % CG3D_BN_SLOTS;
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; e++) s0[e] = s1[e] = 0.f;
    {
        const int q = tq;
        // rows first: their addresses do not depend on the per-channel constants, whose loads then hide behind them
        constexpr int U = 4;
        uint4 a[U], x[U], y[U];
        bool ok[U];
        auto load = [&](int r) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int rr = r + u * rpb;
                ok[u] = rr < nr;
                const int64_t off = (int64_t)(r0 + (ok[u] ? rr : (r < nr ? r : 0))) * co + q;      // clamped, unconditional load
                a[u] = reinterpret_cast<const uint4 *>(dY)[off];
                x[u] = reinterpret_cast<const uint4 *>(X)[off];
                if (act) y[u] = reinterpret_cast<const uint4 *>(Yv)[off];
            }
        };
        load(tr);
        const bn_v8 mu = bn_ldf8(mean + (int64_t)g * c + 8 * q), vv = bn_ldf8(var + (int64_t)g * c + 8 * q);
        float is[8];
#pragma unroll
        for (int e = 0; e < 8; e++) is[e] = rsqrtf(vv.v[e] + eps);
        for (int r = tr; r < nr; r += U * rpb) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (!ok[u]) continue;
                const bn_v8 av = bn_unpack8(a[u]), xv = bn_unpack8(x[u]);
                bn_v8 yv;
                if (act) yv = bn_unpack8(y[u]);
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float d = av.v[e];
                    if (act) d *= act_bwd(yv.v[e], act);
                    s0[e] += d;
                    s1[e] += d * (xv.v[e] - mu.v[e]) * is[e];
                }
            }
            if (r + U * rpb < nr) load(r + U * rpb);
        }
    }
    // lanes of a wave that own the same oct (stride tpr) meet by xor-shuffles; the waves meet in LDS
    for (int off = tpr; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; e++) { s0[e] += __shfl_xor(s0[e], off); s1[e] += __shfl_xor(s1[e], off); }
    const int wo = tpr < 64 ? tpr : 64;                  // octs a wave covers
    if (lane < wo) {
#pragma unroll
        for (int e = 0; e < 8; e++) { red[wave][0][lane][e] = s0[e]; red[wave][1][lane][e] = s1[e]; }
    }
    __syncthreads();
    if ((int)threadIdx.x < tpr) {
        const int t = threadIdx.x;
        float f0[8], f1[8];
#pragma unroll
        for (int e = 0; e < 8; e++) f0[e] = f1[e] = 0.f;
        for (int w = 0; w < 4; w++) {
            // wave w covers the octs (64 w) % tpr ... + wo - 1
            const int first = (64 * w) % tpr;
            if (t >= first && t < first + wo) {
#pragma unroll
                for (int e = 0; e < 8; e++) { f0[e] += red[w][0][t - first][e]; f1[e] += red[w][1][t - first][e]; }
            }
        }
        float *w0 = sums + ((int64_t)(slot * 2) * G + g) * c + 8 * t, *w1 = sums + ((int64_t)(slot * 2 + 1) * G + g) * c + 8 * t;
#ifdef CG3D_BN_DBG_NOATOM          // dev build only (tools/_r05_y.sh): plain stores, wrong sums -- what the atomic chains cost
#pragma unroll
        for (int e = 0; e < 8; e++) { w0[e] = f0[e]; w1[e] = f1[e]; }
#else
#pragma unroll
        for (int e = 0; e < 8; e++) { unsafeAtomicAdd(w0 + e, f0[e]); unsafeAtomicAdd(w1 + e, f1[e]); }
#endif
    }
}

template <bool SUMS>
__global__ __launch_bounds__(256) void k_bn16_apply(const uint16_t *__restrict__ X, const uint16_t *__restrict__ R,
                                                    const int32_t *__restrict__ chunks, int32_t c, float *__restrict__ mean,
                                                    float *__restrict__ var, float eps, const float *__restrict__ gamma,
                                                    const float *__restrict__ beta, int act, uint16_t *__restrict__ Y,
                                                    const float *__restrict__ sums, const float *__restrict__ group_n, int G,
                                                    float *__restrict__ run_mean, float *__restrict__ run_var,
                                                    long long *__restrict__ nbt, float momentum) {
    // c / 8 <= 256 octs, c <= BN_LDS_C with SUMS (the entry points check): one oct per thread column
    const int co = c >> 3;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const bool first_of_group = SUMS && (blockIdx.x == 0 || chunks[(blockIdx.x - 1) * 3] != g);
    const int tpr = co < 256 ? co : 256, rpb = 256 / tpr;
    const int q = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const bool active = tr < rpb;
    __shared__ __attribute__((aligned(16))) float s_mu[SUMS ? BN_LDS_C : 8], s_var[SUMS ? BN_LDS_C : 8];
    // the first rows are requested BEFORE the statistics are derived: their addresses do not depend on them, and a chunk is
    // one or two trips long -- the table loads, the fp64 sums and the barrier hide behind the rows' round trip
    uint4 xv[4], rv[4];                                  // (packed: 8 bf16 per register quad, widened where they are used)
    int64_t off[4];
    bool ok[4];
    auto load = [&](int r) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int rr = r + u * rpb;
            ok[u] = active && rr < nr;
            off[u] = (int64_t)(r0 + (ok[u] ? rr : 0)) * co + q;
            xv[u] = reinterpret_cast<const uint4 *>(X)[off[u]];
            if (R) rv[u] = reinterpret_cast<const uint4 *>(R)[off[u]];
        }
    };
    load(tr);
    bn_v8 mu, vv;
    if (SUMS) {
        bn_mean_var_to_lds(sums, group_n, G, g, c, s_mu, s_var, first_of_group, mean, var, run_mean, run_var, nbt, momentum);
        __syncthreads();
        mu = bn_ldf8(&s_mu[8 * q]);
        vv = bn_ldf8(&s_var[8 * q]);
    } else {
        mu = bn_ldf8(mean + (int64_t)g * c + 8 * q);
        vv = bn_ldf8(var + (int64_t)g * c + 8 * q);
    }
    if (!active) return;
    const bn_v8 ga = bn_ldf8(gamma + (int64_t)g * c + 8 * q), be = bn_ldf8(beta + (int64_t)g * c + 8 * q);
    float sc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) sc[e] = rsqrtf(vv.v[e] + eps);
    for (int r = tr; r < nr; r += 4 * rpb) {
        uint4 yv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const bn_v8 x = bn_unpack8(xv[u]);
            bn_v8 rr, y;
            if (R) rr = bn_unpack8(rv[u]);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                float t = (x.v[e] - mu.v[e]) * sc[e] * ga.v[e] + be.v[e];
                if (R) t += rr.v[e];
                y.v[e] = act_fwd(t, act);
            }
            yv[u] = bn_pack8(y);
        }
        bool okc[4];
        int64_t offc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { okc[u] = ok[u]; offc[u] = off[u]; }
        if (r + 4 * rpb < nr) load(r + 4 * rpb);         // the next trip's rows, ahead of this trip's stores
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (okc[u]) reinterpret_cast<uint4 *>(Y)[offc[u]] = yv[u];
    }
}

template <bool SUMS>
__global__ __launch_bounds__(256) void k_bn16_bwd_apply(const uint16_t *__restrict__ dY, const uint16_t *__restrict__ X,
                                                        const uint16_t *__restrict__ Yv, const int32_t *__restrict__ chunks,
                                                        int32_t c, const float *__restrict__ mean, const float *__restrict__ var,
                                                        float eps, const float *__restrict__ gamma, float *__restrict__ dbeta,
                                                        float *__restrict__ dgamma, const float *__restrict__ group_n, int act,
                                                        int use_batch, uint16_t *__restrict__ dX, uint16_t *__restrict__ dR,
                                                        const float *__restrict__ dsums, int G) {
    const int co = c >> 3;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const bool first_of_group = SUMS && (blockIdx.x == 0 || chunks[(blockIdx.x - 1) * 3] != g);
    const float inv_n = use_batch ? 1.f / group_n[g] : 0.f;
    const int tpr = co < 256 ? co : 256, rpb = 256 / tpr;
    const int q = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const bool active = tr < rpb;
    __shared__ __attribute__((aligned(16))) float s_b[SUMS ? BN_LDS_C : 8], s_g[SUMS ? BN_LDS_C : 8];
    constexpr int U = 2;                                 // rows per trip (three input streams: 6 x 16 bytes in flight per thread)
    uint4 dv[U], xv[U], yv[U];
    int64_t off[U];
    bool ok[U];
    auto load = [&](int r) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int rr = r + u * rpb;
            ok[u] = active && rr < nr;
            off[u] = (int64_t)(r0 + (ok[u] ? rr : 0)) * co + q;
            dv[u] = reinterpret_cast<const uint4 *>(dY)[off[u]];
            xv[u] = reinterpret_cast<const uint4 *>(X)[off[u]];
            if (act) yv[u] = reinterpret_cast<const uint4 *>(Yv)[off[u]];
        }
    };
    load(tr);                                            // (before the slot sums: see k_bn16_apply)
    bn_v8 sb, sg;
    if (SUMS) {
        bn_slot_pair_to_lds(dsums, G, g, c, s_b, s_g, first_of_group, dbeta, dgamma);
        __syncthreads();
        sb = bn_ldf8(&s_b[8 * q]);
        sg = bn_ldf8(&s_g[8 * q]);
    } else {
        sb = bn_ldf8(dbeta + (int64_t)g * c + 8 * q);
        sg = bn_ldf8(dgamma + (int64_t)g * c + 8 * q);
    }
    if (!active) return;
    const bn_v8 mu = bn_ldf8(mean + (int64_t)g * c + 8 * q), vv = bn_ldf8(var + (int64_t)g * c + 8 * q);
    const bn_v8 ga = bn_ldf8(gamma + (int64_t)g * c + 8 * q);
    float is[8];
#pragma unroll
    for (int e = 0; e < 8; e++) is[e] = rsqrtf(vv.v[e] + eps);
    for (int r = tr; r < nr; r += U * rpb) {
        uint4 dp[U], op[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bn_v8 dyv = bn_unpack8(dv[u]), x = bn_unpack8(xv[u]);
            bn_v8 y, d, o;
            if (act) y = bn_unpack8(yv[u]);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                d.v[e] = dyv.v[e] * (act ? act_bwd(y.v[e], act) : 1.f);
                o.v[e] = ga.v[e] * is[e] * (d.v[e] - (sb.v[e] + (x.v[e] - mu.v[e]) * is[e] * sg.v[e]) * inv_n);
            }
            dp[u] = bn_pack8(d);
            op[u] = bn_pack8(o);
        }
        bool okc[U];
        int64_t offc[U];
#pragma unroll
        for (int u = 0; u < U; u++) { okc[u] = ok[u]; offc[u] = off[u]; }
        if (r + U * rpb < nr) load(r + U * rpb);
#pragma unroll
        for (int u = 0; u < U; u++)
            if (okc[u]) {
                if (dR) reinterpret_cast<uint4 *>(dR)[offc[u]] = dp[u];
                reinterpret_cast<uint4 *>(dX)[offc[u]] = op[u];
            }
    }
}

// SUMS: mean / variance are not inputs but derived from the zero-based statistics table sums[2][G][C] of the layer
// (cg3d_bn_sums, or the producing convolution's epilogue) and the group's row count -- in fp64, per thread, for its own
// channel quad; the first chunk of every group also writes them to mean / var (the backward pass reads them) and updates
// the running statistics like nn.BatchNorm1d (momentum, unbiased variance).
template <bool SUMS>
__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ X, const float *__restrict__ R,
                                                  const int32_t *__restrict__ chunks, int32_t c,
                                                  float *__restrict__ mean, float *__restrict__ var, float eps,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, int act,
                                                  float *__restrict__ Y, uint2 *__restrict__ Y16,
                                                  const float *__restrict__ sums, const float *__restrict__ group_n, int G,
                                                  float *__restrict__ run_mean, float *__restrict__ run_var,
                                                  long long *__restrict__ nbt, float momentum) {
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const bool first_of_group = SUMS && (blockIdx.x == 0 || chunks[(blockIdx.x - 1) * 3] != g);
    // thread = (channel quad tq, row lane tr): the per-channel constants are loaded and inverted ONCE per thread, rows
    // are walked four at a time with all loads issued before the first use (memory-level parallelism)
    const int tpr = cq < 256 ? cq : 256, rpb = 256 / tpr;
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    __shared__ float s_mu[SUMS ? BN_LDS_C : 4], s_var[SUMS ? BN_LDS_C : 4];
    const bool lds = SUMS && c <= BN_LDS_C;
    if (lds) {
        bn_mean_var_to_lds(sums, group_n, G, g, c, s_mu, s_var, first_of_group, mean, var, run_mean, run_var, nbt, momentum);
        __syncthreads();
    }
    if (tr >= rpb) return;
    for (int q = tq; q < cq; q += tpr) {
        float4 mu, vv;
        if (lds) {
            mu = *reinterpret_cast<const float4 *>(&s_mu[4 * q]);
            vv = *reinterpret_cast<const float4 *>(&s_var[4 * q]);
        } else if (SUMS) {
            // (more than BN_LDS_C channels: every thread adds the slots of its own channels up)
            // the 16 slot sums are added up in fp64 (each slot is an fp32 atomic chain over ~1/16 of the rows; the slot sums
            // themselves are large and nearly equal, so their sum is where fp32 would lose the low bits E[x^2] - mean^2 needs)
            double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
#pragma unroll
            for (int sl = 0; sl < CG3D_BN_SLOTS; sl++) {
                const float4 u0 = reinterpret_cast<const float4 *>(sums + ((int64_t)(sl * 2) * G + g) * c)[q];
                const float4 u1 = reinterpret_cast<const float4 *>(sums + ((int64_t)(sl * 2 + 1) * G + g) * c)[q];
                a0[0] += u0.x; a0[1] += u0.y; a0[2] += u0.z; a0[3] += u0.w;
                a1[0] += u1.x; a1[1] += u1.y; a1[2] += u1.z; a1[3] += u1.w;
            }
            const double n = group_n[g] > 0.f ? (double)group_n[g] : 1.0;
            const double m0 = a0[0] / n, m1 = a0[1] / n, m2 = a0[2] / n, m3 = a0[3] / n;
            const double v0 = a1[0] / n - m0 * m0, v1 = a1[1] / n - m1 * m1, v2 = a1[2] / n - m2 * m2, v3 = a1[3] / n - m3 * m3;
            mu = make_float4((float)m0, (float)m1, (float)m2, (float)m3);
            vv = make_float4((float)(v0 > 0 ? v0 : 0), (float)(v1 > 0 ? v1 : 0), (float)(v2 > 0 ? v2 : 0), (float)(v3 > 0 ? v3 : 0));
            if (first_of_group && tr == 0) {
                reinterpret_cast<float4 *>(mean + (int64_t)g * c)[q] = mu;
                reinterpret_cast<float4 *>(var + (int64_t)g * c)[q] = vv;
                if (run_mean && run_var) {
                    const float unb = (float)(n / (n > 1.0 ? n - 1.0 : 1.0));
                    float4 *rm = reinterpret_cast<float4 *>(run_mean + (int64_t)g * c) + q, *rvp = reinterpret_cast<float4 *>(run_var + (int64_t)g * c) + q;
                    float4 a = *rm, b = *rvp;
                    a.x = (1.f - momentum) * a.x + momentum * mu.x; a.y = (1.f - momentum) * a.y + momentum * mu.y;
                    a.z = (1.f - momentum) * a.z + momentum * mu.z; a.w = (1.f - momentum) * a.w + momentum * mu.w;
                    b.x = (1.f - momentum) * b.x + momentum * (vv.x * unb); b.y = (1.f - momentum) * b.y + momentum * (vv.y * unb);
                    b.z = (1.f - momentum) * b.z + momentum * (vv.z * unb); b.w = (1.f - momentum) * b.w + momentum * (vv.w * unb);
                    *rm = a; *rvp = b;
                }
                if (nbt && q == 0) nbt[g] += 1;
            }
        } else {
            mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q];
            vv = reinterpret_cast<const float4 *>(var + (int64_t)g * c)[q];
        }
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
    if (act & CG3D_BN_STORE_BF16) {          // X, residual, Y are bf16 rows
        if (!bn16_ok(c) || Y16) return CG3D_ERR_ARG;
        hipLaunchKernelGGL(k_bn16_apply<false>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), reinterpret_cast<const uint16_t *>(X),
                           reinterpret_cast<const uint16_t *>(residual), chunks, c, const_cast<float *>(mean), const_cast<float *>(var), eps,
                           gamma, beta, act & ~CG3D_BN_STORE_BF16, reinterpret_cast<uint16_t *>(Y), nullptr, nullptr, 1, nullptr, nullptr,
                           nullptr, 0.f);
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    hipLaunchKernelGGL(k_bn_apply<false>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), X, residual, chunks, c,
                       const_cast<float *>(mean), const_cast<float *>(var), eps, gamma, beta, act, Y, reinterpret_cast<uint2 *>(Y16),
                       nullptr, nullptr, 1, nullptr, nullptr, nullptr, 0.f);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_bn_apply_sums(const float *X, const float *residual, const int32_t *chunks, int64_t nchunk, int32_t G,
                                  int32_t c, const float *sums, const float *group_n, float eps, const float *gamma,
                                  const float *beta, int32_t act, float *Y, uint16_t *Y16, float *mean, float *var,
                                  float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum,
                                  cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || bad(X) || bad(Y) || bad(residual) || ((uintptr_t)Y16 & 7) || bad(sums) || bad(mean) ||
        bad(var) || bad(running_mean) || bad(running_var) || !sums || !group_n || !mean || !var)
        return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    if (act & CG3D_BN_STORE_BF16) {
        if (!bn16_ok(c) || Y16) return CG3D_ERR_ARG;
        hipLaunchKernelGGL(k_bn16_apply<true>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), reinterpret_cast<const uint16_t *>(X),
                           reinterpret_cast<const uint16_t *>(residual), chunks, c, mean, var, eps, gamma, beta, act & ~CG3D_BN_STORE_BF16,
                           reinterpret_cast<uint16_t *>(Y), sums, group_n, G, running_mean, running_var,
                           reinterpret_cast<long long *>(num_batches_tracked), momentum);
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    hipLaunchKernelGGL(k_bn_apply<true>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), X, residual, chunks, c, mean, var, eps,
                       gamma, beta, act, Y, reinterpret_cast<uint2 *>(Y16), sums, group_n, G, running_mean, running_var,
                       reinterpret_cast<long long *>(num_batches_tracked), momentum);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// SUMS: dbeta / dgamma are not inputs but the slot sums of the backward statistics table dsums[CG3D_BN_SLOTS][2][G][C]
// (cg3d_bn_bwd_sums); the first chunk of every group writes them to dbeta / dgamma (the parameters' gradients).
template <bool SUMS>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ dY, const float *__restrict__ X,
                                                      const float *__restrict__ Yv, const int32_t *__restrict__ chunks,
                                                      int32_t c, const float *__restrict__ mean,
                                                      const float *__restrict__ var, float eps,
                                                      const float *__restrict__ gamma, float *__restrict__ dbeta,
                                                      float *__restrict__ dgamma, const float *__restrict__ group_n,
                                                      int act, int use_batch, float *__restrict__ dX,
                                                      uint2 *__restrict__ dX16, float *__restrict__ dR,
                                                      const float *__restrict__ dsums, int G) {
    const int cq = c >> 2;
    const int g = chunks[blockIdx.x * 3], r0 = chunks[blockIdx.x * 3 + 1], nr = chunks[blockIdx.x * 3 + 2];
    const bool first_of_group = SUMS && (blockIdx.x == 0 || chunks[(blockIdx.x - 1) * 3] != g);
    const float inv_n = use_batch ? 1.f / group_n[g] : 0.f;
    const int tpr = cq < 256 ? cq : 256, rpb = 256 / tpr;
    const int tq = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    __shared__ float s_b[SUMS ? BN_LDS_C : 4], s_g[SUMS ? BN_LDS_C : 4];
    const bool lds = SUMS && c <= BN_LDS_C;
    if (lds) {
        bn_slot_pair_to_lds(dsums, G, g, c, s_b, s_g, first_of_group, dbeta, dgamma);
        __syncthreads();
    }
    if (tr >= rpb) return;
    for (int q = tq; q < cq; q += tpr) {
        const float4 mu = reinterpret_cast<const float4 *>(mean + (int64_t)g * c)[q];
        const float4 vv = reinterpret_cast<const float4 *>(var + (int64_t)g * c)[q];
        const float4 ga = reinterpret_cast<const float4 *>(gamma + (int64_t)g * c)[q];
        float4 sb, sg;
        if (lds) {
            sb = *reinterpret_cast<const float4 *>(&s_b[4 * q]);
            sg = *reinterpret_cast<const float4 *>(&s_g[4 * q]);
        } else if (SUMS) {
            double b0[4] = {0, 0, 0, 0}, b1[4] = {0, 0, 0, 0};            // slot sums added in fp64, as in the forward
#pragma unroll
            for (int sl = 0; sl < CG3D_BN_SLOTS; sl++) {
                const float4 u0 = reinterpret_cast<const float4 *>(dsums + ((int64_t)(sl * 2) * G + g) * c)[q];
                const float4 u1 = reinterpret_cast<const float4 *>(dsums + ((int64_t)(sl * 2 + 1) * G + g) * c)[q];
                b0[0] += u0.x; b0[1] += u0.y; b0[2] += u0.z; b0[3] += u0.w;
                b1[0] += u1.x; b1[1] += u1.y; b1[2] += u1.z; b1[3] += u1.w;
            }
            sb = make_float4((float)b0[0], (float)b0[1], (float)b0[2], (float)b0[3]);
            sg = make_float4((float)b1[0], (float)b1[1], (float)b1[2], (float)b1[3]);
            if (first_of_group && tr == 0) {
                reinterpret_cast<float4 *>(dbeta + (int64_t)g * c)[q] = sb;
                reinterpret_cast<float4 *>(dgamma + (int64_t)g * c)[q] = sg;
            }
        } else {
            sb = reinterpret_cast<const float4 *>(dbeta + (int64_t)g * c)[q];
            sg = reinterpret_cast<const float4 *>(dgamma + (int64_t)g * c)[q];
        }
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
    if (act & CG3D_BN_STORE_BF16) {          // dY, X, Y, dX, dRes are bf16 rows
        if (!bn16_ok(c) || dX16) return CG3D_ERR_ARG;
        hipLaunchKernelGGL(k_bn16_bwd_apply<false>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), reinterpret_cast<const uint16_t *>(dY),
                           reinterpret_cast<const uint16_t *>(X), reinterpret_cast<const uint16_t *>(Y), chunks, c, mean, var, eps, gamma,
                           const_cast<float *>(dbeta), const_cast<float *>(dgamma), group_n, act & ~CG3D_BN_STORE_BF16, use_batch_stats,
                           reinterpret_cast<uint16_t *>(dX), reinterpret_cast<uint16_t *>(dRes), nullptr, 1);
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    hipLaunchKernelGGL(k_bn_bwd_apply<false>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), dY, X, Y, chunks, c, mean, var,
                       eps, gamma, const_cast<float *>(dbeta), const_cast<float *>(dgamma), group_n, act, use_batch_stats, dX,
                       reinterpret_cast<uint2 *>(dX16), dRes, nullptr, 1);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_bn_bwd_apply_sums(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                                      int32_t G, int32_t c, const float *mean, const float *var, float eps, const float *gamma,
                                      const float *dsums, const float *group_n, int32_t act, int32_t use_batch_stats,
                                      float *dX, uint16_t *dX16, float *dRes, float *dbeta, float *dgamma,
                                      cg3d_stream_t stream) {
    if (nchunk < 0 || G < 1 || c < 4 || (c & 3) || bad(X) || bad(dY) || bad(Y) || bad(dX) || bad(dRes) || ((uintptr_t)dX16 & 7) ||
        bad(dsums) || bad(dbeta) || bad(dgamma) || !dsums || !dbeta || !dgamma)
        return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    if (act & CG3D_BN_STORE_BF16) {
        if (!bn16_ok(c) || dX16) return CG3D_ERR_ARG;
        hipLaunchKernelGGL(k_bn16_bwd_apply<true>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), reinterpret_cast<const uint16_t *>(dY),
                           reinterpret_cast<const uint16_t *>(X), reinterpret_cast<const uint16_t *>(Y), chunks, c, mean, var, eps, gamma,
                           dbeta, dgamma, group_n, act & ~CG3D_BN_STORE_BF16, use_batch_stats, reinterpret_cast<uint16_t *>(dX),
                           reinterpret_cast<uint16_t *>(dRes), dsums, G);
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    hipLaunchKernelGGL(k_bn_bwd_apply<true>, dim3((unsigned)nchunk), dim3(256), 0, cg3d_hs(stream), dY, X, Y, chunks, c, mean, var, eps,
                       gamma, dbeta, dgamma, group_n, act, use_batch_stats, dX, reinterpret_cast<uint2 *>(dX16), dRes, dsums, G);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

