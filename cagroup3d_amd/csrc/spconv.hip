// spconv.hip -- sparse 3D convolution for gfx950 as an output-stationary implicit GEMM on the
// matrix cores: gather rows through the kernel map -> per-offset [rows x Cin] x [Cin x Cout]
// contraction on MFMA -> direct store (no scatter, no atomics in the forward / data-gradient).
//
// Replaces MinkowskiEngine's ConvolutionForwardGPU / ConvolutionBackwardGPU (gather -> GEMM ->
// scatter-add per offset; un-vendored) for every MinkowskiConvolution /
// MinkowskiConvolutionTranspose / MinkowskiGenerativeConvolutionTranspose call site in
// pcdet/models/backbones_3d/biresnet.py, dense_heads/cagroup_head.py and
// roi_heads/cagroup_roi_head.py:69.
//
// Tiling (wave64, v_mfma_f32_32x32x2_f32 = exact fp32 products, fp32 accumulate):
//   workgroup = 4 waves = 128 output rows x CT output channels; wave = 32 rows x CT.
//   The MFMA's two k-slices are mapped to the two HALVES of a 64-channel chunk, so lane (r, h)
//   reads 32 CONTIGUOUS input channels of its gathered row straight into registers (8 x 16-byte
//   loads, no LDS round trip, no transposition), and the matching B fragment W[k][c][n0 + r] is a
//   coalesced 128-byte row read by the 32 lanes of a half-wave (L1/L2 resident: every workgroup
//   streams the same W[k]).
//   Offsets with no valid neighbour in the wave's 32 rows are skipped with one wave ballot.
// The weight gradient consumes rows two at a time (k-slices = two rows), so its ballot skips at
// 2-row granularity -- almost all padding work vanishes on the sparse high-resolution maps.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdlib.h>
#include "cg3d_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------- forward / data gradient
// KH = channels handled per k-slice (32 for Cin >= 64 chunks, 2 for Cin <= 4)
template <int NT, int KH, bool VEC4>
__global__ __launch_bounds__(256) void k_spconv_fwd(const float *__restrict__ X, const float *__restrict__ W,
                                                    const int32_t *__restrict__ nbr,
                                                    const float *__restrict__ bias, float *__restrict__ Y,
                                                    int64_t n_out, int32_t K, int32_t cin, int32_t cout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + r;
    const int n0 = blockIdx.y * (NT * 32);
    const bool row_ok = row < n_out;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    bool col_ok[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) col_ok[t] = (n0 + t * 32 + r) < cout;

    for (int32_t k = 0; k < K; k++) {
        int32_t idx = row_ok ? nbr[(int64_t)k * n_out + row] : -1;
        if (!__any(idx >= 0)) continue;  // wave-uniform skip
        const float *xrow = X + (int64_t)(idx < 0 ? 0 : idx) * cin;
        const float *wk = W + (int64_t)k * cin * cout;
        for (int32_t c0 = 0; c0 < cin; c0 += 2 * KH) {
            const int cb = c0 + h * KH;  // first channel of this lane's k-slice
            float a[KH];
            if (VEC4) {
#pragma unroll
                for (int t = 0; t < KH; t += 4) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx >= 0 && cb + t < cin) v = *reinterpret_cast<const float4 *>(xrow + cb + t);
                    a[t] = v.x; a[t + 1] = v.y; a[t + 2] = v.z; a[t + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < KH; t++) a[t] = (idx >= 0 && cb + t < cin) ? xrow[cb + t] : 0.f;
            }
#pragma unroll
            for (int t = 0; t < KH; t++) {
                const bool c_ok = (cb + t) < cin;
                const float *wr = wk + (int64_t)(cb + t) * cout + n0 + r;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    float b = (c_ok && col_ok[nt]) ? wr[nt * 32] : 0.f;
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: C/D layout col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const int64_t row_base = (int64_t)blockIdx.x * 128 + wave * 32;
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = n0 + nt * 32 + r;
        if (col >= cout) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            int64_t orow = row_base + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (orow < n_out) Y[orow * cout + col] = acc[nt][e] + bv;
        }
    }
}

// ---------------------------------------------------------------- weight gradient
// dW[k][ci][co] = sum_o X[nbr[k][o]][ci] * dY[o][co]
// grid: x = row chunk, y = offset k, z = (ci tile, co tile) of 64x64.  Each wave walks its share of
// the chunk in 64-row groups; an MFMA step contracts TWO rows (k-slice h = row t + 32*h).
__global__ __launch_bounds__(256) void k_spconv_wgrad(const float *__restrict__ X, const float *__restrict__ dY,
                                                      const int32_t *__restrict__ nbr, float *__restrict__ dW,
                                                      int64_t n_out, int32_t cin, int32_t cout, int64_t chunk_rows,
                                                      int32_t co_tiles, int32_t use_atomics) {
    __shared__ float tile[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int32_t k = blockIdx.y;
    const int ci0 = (blockIdx.z / co_tiles) * 64, co0 = (blockIdx.z % co_tiles) * 64;
    const int64_t row_begin = (int64_t)blockIdx.x * chunk_rows;
    const int64_t row_end = (row_begin + chunk_rows < n_out) ? row_begin + chunk_rows : n_out;

    for (int i = threadIdx.x; i < 64 * 64; i += 256) tile[i] = 0.f;
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const bool ci_ok[2] = {ci0 + r < cin, ci0 + 32 + r < cin};
    const bool co_ok[2] = {co0 + r < cout, co0 + 32 + r < cout};
    const int32_t *nk = nbr + (int64_t)k * n_out;

    for (int64_t g = row_begin + wave * 64; g < row_end; g += 256) {
        const int64_t myrow = g + lane;
        const int32_t myidx = (myrow < row_end) ? nk[myrow] : -1;
        const unsigned long long valid = __ballot(myidx >= 0);
        if (valid == 0ULL) continue;
        for (int t = 0; t < 32; t++) {
            if (((valid >> t) & 0x100000001ULL) == 0ULL) continue;  // both rows of this step absent
            const int src = t + 32 * h;
            const int32_t idx = __shfl(myidx, src);
            const int64_t orow = g + src;
            float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
            if (idx >= 0) {
                const float *xr = X + (int64_t)idx * cin + ci0 + r;
                const float *dr = dY + orow * cout + co0 + r;
                if (ci_ok[0]) a0 = xr[0];
                if (ci_ok[1]) a1 = xr[32];
                if (co_ok[0]) b0 = dr[0];
                if (co_ok[1]) b1 = dr[32];
            }
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
    }
    // cross-wave reduction in LDS (ds_add_f32), then one store / global atomic per element
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                int tr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;  // ci within tile
                int tc = j * 32 + r;                              // co within tile
                atomicAdd(&tile[tr * 64 + tc], acc[i][j][e]);
            }
    __syncthreads();
    float *dst = dW + (int64_t)k * cin * cout;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int ci = ci0 + (i >> 6), co = co0 + (i & 63);
        if (ci < cin && co < cout) {
            if (use_atomics) unsafeAtomicAdd(&dst[(int64_t)ci * cout + co], tile[i]);
            else dst[(int64_t)ci * cout + co] = tile[i];
        }
    }
}

extern "C" int cg3d_spconv_wgrad(const float *X, const float *dY, const int32_t *nbr, float *dW, int64_t n_in,
                                 int64_t n_out, int32_t K, int32_t cin, int32_t cout, int32_t precision,
                                 cg3d_stream_t stream) {
    (void)n_in;
    if (n_out < 0 || K < 1 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    if (precision != 0) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    const int64_t nw = (int64_t)K * cin * cout;
    if (n_out == 0) {
        if (hipMemsetAsync(dW, 0, nw * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
        return CG3D_OK;
    }
    const int32_t ci_tiles = (cin + 63) / 64, co_tiles = (cout + 63) / 64;
    const int64_t base_wgs = (int64_t)K * ci_tiles * co_tiles;
    // enough workgroups to fill 256 CUs x 4; each chunk at least 512 rows
    int64_t chunks = cg3d_divup(2048, base_wgs);
    const int64_t max_chunks = cg3d_divup(n_out, 512);
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    if (chunks > 65535) chunks = 65535;
    int64_t chunk_rows = cg3d_divup(cg3d_divup(n_out, chunks), 256) * 256;
    chunks = cg3d_divup(n_out, chunk_rows);
    const int use_atomics = chunks > 1;
    if (use_atomics && hipMemsetAsync(dW, 0, nw * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (K > 65535) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_spconv_wgrad, dim3((unsigned)chunks, (unsigned)K, (unsigned)(ci_tiles * co_tiles)), dim3(256), 0,
                       s, X, dY, nbr, dW, n_out, cin, cout, chunk_rows, co_tiles, use_atomics);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// =====================================================================================
// Pair-compacted path: gather -> MFMA -> atomic scatter.  No work on absent neighbours.
// Workgroup = one segment of <= 128 pairs of ONE offset k (4 waves x 32 pairs) x one group of
// NT*32 output channels.  W[k][chunk of 64 input channels][column group] is staged in LDS once per
// workgroup and shared by the 4 waves (ds_read_b32 of 32 consecutive columns: conflict-free);
// the gathered rows go straight to registers (32 contiguous channels per lane, 16-byte loads).
// Results are added to Y with fp32 global atomics: 32 lanes x 4 B = one 128-byte row segment per
// instruction half.
// =====================================================================================
// Fast path (cin % 4 == 0, 16-byte aligned): both operands go through LDS in FULL cache lines.
//   * gathered rows: one wave instruction reads 4 rows x 256 B (16 lanes x 16 B per row), so a
//     128-byte line is consumed by a single instruction -- the direct per-lane fragment gather
//     re-fetched every line up to 8x once the 4 waves' 32 KB of in-flight lines overflowed the L1;
//   * the next chunk's global loads (W tile + rows) are issued into registers BEFORE the current
//     chunk's MFMAs and written to LDS after them, so HBM/L2 latency hides under the matrix pipe;
template <int NT>
__global__ __launch_bounds__(256, 2) void k_spconv_pairs_lds(const float *__restrict__ X, const float *__restrict__ W,
                                                          const int32_t *__restrict__ pin,
                                                          const int32_t *__restrict__ pout,
                                                          const int32_t *__restrict__ seg, float *__restrict__ Y,
                                                          int32_t cin, int32_t cout) {
    constexpr int CT = NT * 32;
    constexpr int KC = 64;               // input channels per chunk
    constexpr int AP = KC + 4;           // padded row of the gathered tile (bank-conflict-free b128 reads)
    constexpr int WV = KC * CT / 4 / 256;  // float4 of the W tile per thread
    __shared__ float Ws[KC * CT];
    __shared__ float As[4 * 32 * AP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;               // placeholder entries keep the XCD alignment of a segment table (uniform branch)
    const int n0 = blockIdx.y * CT;
    const int local = wave * 32 + r;
    const bool valid = local < count;
    const int32_t irow = valid ? pin[start + local] : -1;
    const int32_t orow = valid ? pout[start + local] : -1;
    const bool wave_active = wave * 32 < count;
    const float *wk = W + (int64_t)k * cin * cout;
    const bool wvec = (cout % 4 == 0);
    float *Aw = &As[wave * 32 * AP];

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    // row handled by this lane in gather instruction i: 4*i + (lane >> 4); its 16-byte column: lane & 15
    int32_t grow[8];
#pragma unroll
    for (int i = 0; i < 8; i++) grow[i] = __shfl(irow, 4 * i + (lane >> 4));
    const int gcol = (lane & 15) * 4;

    float4 wreg[WV], areg[8];
    auto issue_loads = [&](int32_t c0) {
#pragma unroll
        for (int i = 0; i < WV; i++) {
            const int idx = tid + i * 256;
            const int row = idx / (CT / 4), c4 = (idx % (CT / 4)) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c0 + row < cin) {
                const float *src = wk + (int64_t)(c0 + row) * cout + n0 + c4;
                if (wvec) { if (n0 + c4 < cout) v = *reinterpret_cast<const float4 *>(src); }
                else {
                    if (n0 + c4 < cout) v.x = src[0];
                    if (n0 + c4 + 1 < cout) v.y = src[1];
                    if (n0 + c4 + 2 < cout) v.z = src[2];
                    if (n0 + c4 + 3 < cout) v.w = src[3];
                }
            }
            wreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (grow[i] >= 0 && c0 + gcol < cin)
                v = *reinterpret_cast<const float4 *>(X + (int64_t)grow[i] * cin + c0 + gcol);
            areg[i] = v;
        }
    };
    issue_loads(0);
    for (int32_t c0 = 0; c0 < cin; c0 += KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < WV; i++) {
            const int idx = tid + i * 256;
            *reinterpret_cast<float4 *>(&Ws[(idx / (CT / 4)) * CT + (idx % (CT / 4)) * 4]) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            *reinterpret_cast<float4 *>(&Aw[(4 * i + (lane >> 4)) * AP + gcol]) = areg[i];
        __syncthreads();
        if (c0 + KC < cin) issue_loads(c0 + KC);      // in flight during the MFMAs below
        if (wave_active) {
            float a[32];
#pragma unroll
            for (int t = 0; t < 32; t += 4) {
                float4 v = *reinterpret_cast<const float4 *>(&Aw[r * AP + h * 32 + t]);
                a[t] = v.x; a[t + 1] = v.y; a[t + 2] = v.z; a[t + 3] = v.w;
            }
            const float *wbase = &Ws[(h * 32) * CT + r];
#pragma unroll
            for (int t = 0; t < 32; t++) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], wbase[t * CT + nt * 32], acc[nt], 0, 0, 0);
            }
        }
    }
    if (!wave_active) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int rowl = (e & 3) + 8 * (e >> 2) + 4 * h;
        const int32_t orow_e = __shfl(orow, rowl);
        if (orow_e < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int col = n0 + nt * 32 + r;
            if (col < cout) unsafeAtomicAdd(&Y[(int64_t)orow_e * cout + col], acc[nt][e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16-operand path (precision == 1): v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 in HBM.
// Activations are rounded to bf16 (RNE) while they are staged into LDS; the weights arrive already
// rounded and transposed to [slot][cout][cin] (cg3d_spconv_prep_weights_bf16) so that both MFMA
// operands are 16-byte ds_read_b128 fragments (8 consecutive k per lane).  At 16x the fp32 MFMA rate
// the kernel is bound by the row gather and the atomic scatter, not by the matrix pipe.
// ---------------------------------------------------------------------------------------------
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ static inline uint32_t f2bf(float f) {        // round-to-nearest-even, as the oracle's os_bf16
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// four fp32 -> four bf16 (RNE) with the gfx950 packed convert (v_cvt_pk_bf16_f32)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ static inline uint32_t pack2bf(float a, float b) {
    const f32x2_t v = {a, b};
    const bf16x2_t o = __builtin_convertvector(v, bf16x2_t);
    return __builtin_bit_cast(uint32_t, o);
}
__device__ static inline uint2 pack4bf(float4 v) { return make_uint2(pack2bf(v.x, v.y), pack2bf(v.z, v.w)); }

// Four consecutive channels of a gathered row, as they travel global -> registers -> LDS (bf16):
// fp32 rows (precision 1: 16-byte load, rounded on the way into LDS) or rows already stored as bf16
// (precision 2: 8-byte load, no conversion -- half the gather traffic).
template <typename XT> struct Row4;
template <> struct Row4<float> {
    typedef float4 T;
    __device__ static inline uint2 bf(const T &v) { return pack4bf(v); }
    __device__ static inline T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct Row4<uint16_t> {
    typedef uint2 T;
    __device__ static inline uint2 bf(const T &v) { return v; }
    __device__ static inline T zero() { return make_uint2(0u, 0u); }
};

// How a gather wave of the output-stationary kernel fetches its 32 rows x 64 channels per step: fp32 rows as
// 8 loads of 16 B (16 lanes per 256-byte row chunk), bf16 rows as 4 loads of 16 B (8 lanes per 128-byte chunk) --
// the gather is bound by the number of outstanding requests, so the bf16 rows halve its cost.
template <typename XT> struct Gather;
template <> struct Gather<float> {
    typedef float4 T;
    static constexpr int NL = 8, ROWS = 4, LPR = 16, CH = 4;       // loads, rows per load, lanes per row, channels per lane
    __device__ static inline void stage(uint16_t *dst, const T &v) { *reinterpret_cast<uint2 *>(dst) = pack4bf(v); }
    __device__ static inline T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};
template <> struct Gather<uint16_t> {
    typedef uint4 T;
    static constexpr int NL = 4, ROWS = 8, LPR = 8, CH = 8;
    __device__ static inline void stage(uint16_t *dst, const T &v) { *reinterpret_cast<uint4 *>(dst) = v; }
    __device__ static inline T zero() { return make_uint4(0u, 0u, 0u, 0u); }
};

__global__ void k_to_bf16(const float4 *__restrict__ X, uint2 *__restrict__ Xb, int64_t n4) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < n4) Xb[t] = pack4bf(X[t]);
}
extern "C" int cg3d_to_bf16(const float *X, uint16_t *Xb, int64_t n, cg3d_stream_t stream) {
    if (n < 0 || (n & 3) || ((uintptr_t)X & 15) || ((uintptr_t)Xb & 7)) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)cg3d_divup(n / 4, 256)), dim3(256), 0, cg3d_hs(stream),
                       reinterpret_cast<const float4 *>(X), reinterpret_cast<uint2 *>(Xb), n / 4);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

__global__ void k_from_bf16(const uint4 *__restrict__ Xb, float4 *__restrict__ X, int64_t n8) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n8) return;
    const uint4 u = Xb[t];
    X[2 * t] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                           __uint_as_float(u.y & 0xffff0000u));
    X[2 * t + 1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                               __uint_as_float(u.w & 0xffff0000u));
}
extern "C" int cg3d_from_bf16(const uint16_t *Xb, float *X, int64_t n, cg3d_stream_t stream) {
    if (n < 0 || (n & 7) || ((uintptr_t)X & 15) || ((uintptr_t)Xb & 15)) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_from_bf16, dim3((unsigned)cg3d_divup(n / 8, 256)), dim3(256), 0, cg3d_hs(stream),
                       reinterpret_cast<const uint4 *>(Xb), reinterpret_cast<float4 *>(X), n / 8);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ---------------------------------------------------------------- split operands ("bf16x3": fp32-accurate products on the bf16 pipe)
// x = hi + lo + O(2^-18 |x|) with hi = bf16(x), lo = bf16(x - hi).  A product of two fp32 values is then
//   x w = xhi whi + xlo whi + xhi wlo + O(2^-16 |x w|)
// i.e. three bf16 MFMA products accumulated in fp32 -- 3/16 of the cost of v_mfma_f32_32x32x2_f32.  The split is done on the
// OPERANDS, not in the kernels: rows become [hi | lo | hi] (3 c channels), the contraction side of the weights becomes
// [Whi ; Whi ; Wlo], and every bf16 kernel of this library (tile, dense-map, pair, linear) runs unchanged on a three times
// longer contraction.  BASELINE.json configs[1] words its precision as "bf16 backbone": this is how the two heads keep the
// reference's fp32 arithmetic (cagroup_head.py:227-282, cagroup_roi_head.py:69-91) without the fp32 MFMA rate.
__device__ static inline void split_bf(float v, uint32_t &hi, uint32_t &lo) {
    hi = f2bf(v);
    const float r = v - __uint_as_float(hi << 16);               // exact in fp32
    lo = ((hi & 0x7f80u) == 0x7f80u) ? 0u : f2bf(r);             // inf / NaN stay in the hi part alone
}
__global__ void k_to_bf16_split(const float4 *__restrict__ X, uint2 *__restrict__ Xs, int64_t n_rows, int32_t c4) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n_rows * c4) return;
    const int64_t row = t / c4;
    const int32_t q = (int32_t)(t - row * c4);
    const float4 v = X[t];
    uint32_t h[4], l[4];
    split_bf(v.x, h[0], l[0]); split_bf(v.y, h[1], l[1]); split_bf(v.z, h[2], l[2]); split_bf(v.w, h[3], l[3]);
    const uint2 hi = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16)), lo = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    uint2 *dst = Xs + row * 3 * c4 + q;
    dst[0] = hi;
    dst[c4] = lo;
    dst[2 * c4] = hi;
}
extern "C" int cg3d_to_bf16_split(const float *X, uint16_t *Xs, int64_t n_rows, int32_t c, cg3d_stream_t stream) {
    if (n_rows < 0 || c < 4 || (c & 3) || ((uintptr_t)X & 15) || ((uintptr_t)Xs & 7)) return CG3D_ERR_ARG;
    if (n_rows == 0) return CG3D_OK;
    const int64_t n4 = n_rows * (c / 4);
    hipLaunchKernelGGL(k_to_bf16_split, dim3((unsigned)cg3d_divup(n4, 256)), dim3(256), 0, cg3d_hs(stream),
                       reinterpret_cast<const float4 *>(X), reinterpret_cast<uint2 *>(Xs), n_rows, c / 4);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
// Split weights, element form: fp32 [slot][ci][co] -> the forward operand W_t over the contraction index ci' in [0, 3 cin)
// (ci' = part * cin + ci; parts 0, 1 hold bf16(W), part 2 holds bf16(W - bf16(W))) and / or the data gradient's operand
// W_p over co' in [0, 3 cout), each row-major ([slot][co][3 cin] / [slot][ci][3 cout]) or in MFMA fragment order
// (cg3d_frag_index with kdim = 3 cin / 3 cout).
__global__ __launch_bounds__(256) void k_prep_weights_split(const float *__restrict__ W0, const float *const *__restrict__ Ws,
                                                            uint16_t *__restrict__ W_t, uint16_t *__restrict__ W_p,
                                                            int64_t slots_per, int32_t cin, int32_t cout, int32_t frag) {
    const int64_t slot = blockIdx.x;
    const int64_t per = (int64_t)cin * cout;
    const float *src = Ws ? Ws[slot / slots_per] + (slot % slots_per) * per : W0 + slot * per;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.y * 256) {
        const int ci = (int)(i / cout), co = (int)(i % cout);
        uint32_t hi, lo;
        split_bf(src[i], hi, lo);
#pragma unroll
        for (int part = 0; part < 3; part++) {
            const uint16_t b = (uint16_t)(part < 2 ? hi : lo);
            if (W_t) W_t[slot * 3 * per + (frag ? cg3d_frag_index(co, part * cin + ci, 3 * cin) : (int64_t)co * 3 * cin + part * cin + ci)] = b;
            if (W_p) W_p[slot * 3 * per + (frag ? cg3d_frag_index(ci, part * cout + co, 3 * cout) : (int64_t)ci * 3 * cout + part * cout + co)] = b;
        }
    }
}
extern "C" int cg3d_spconv_prep_weights_split(const float *W0, const float *const *Ws, uint16_t *W_t, uint16_t *W_p, int32_t G,
                                              int64_t slots_per, int32_t cin, int32_t cout, int32_t frag, cg3d_stream_t stream) {
    if (G < 1 || slots_per < 0 || cin < 1 || cout < 1 || (!W0 && !Ws) || (!W_t && !W_p)) return CG3D_ERR_ARG;
    if (frag && W_t && ((cin & 15) || (cout & 31))) return CG3D_ERR_ARG;
    if (frag && W_p && ((cout & 15) || (cin & 31))) return CG3D_ERR_ARG;
    const int64_t slots = (int64_t)G * slots_per;
    if (slots == 0) return CG3D_OK;
    if (slots > 0x7fffffffll) return CG3D_ERR_RANGE;
    const unsigned gy = (unsigned)(cg3d_divup((int64_t)cin * cout, 2048) < 64 ? cg3d_divup((int64_t)cin * cout, 2048) : 64);
    hipLaunchKernelGGL(k_prep_weights_split, dim3((unsigned)slots, gy), dim3(256), 0, cg3d_hs(stream), W0, Ws, W_t, W_p, slots_per,
                       cin, cout, frag ? 1 : 0);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// fp32 weights [slot][ci][co] -> bf16 copies, transposed [slot][co][ci] (the MFMA B operand of forward) and/or plain
// [slot][ci][co] (the B operand of the data gradient), 64 x 64 tiles through LDS so both sides stay coalesced.
// The slots may come from G separate tensors (one per class branch, `Ws` = device array of G pointers): the grouped
// convolutions never materialise an fp32 stack of the per-class weights.
__global__ __launch_bounds__(256) void k_prep_weights(const float *__restrict__ W0, const float *const *__restrict__ Ws,
                                                      uint16_t *__restrict__ Wb_t, uint16_t *__restrict__ Wb,
                                                      int64_t slots_per, int32_t cin, int32_t cout, int32_t co_tiles) {
    __shared__ uint16_t T[64][66];
    const int64_t slot = blockIdx.x;
    const int64_t per = (int64_t)cin * cout;
    const float *src = Ws ? Ws[slot / slots_per] + (slot % slots_per) * per : W0 + slot * per;
    const int ci0 = (blockIdx.y / co_tiles) * 64, co0 = (blockIdx.y % co_tiles) * 64;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int r = i >> 6, c = i & 63;                       // r: input channel, c: output channel
        const bool ok = ci0 + r < cin && co0 + c < cout;
        const uint16_t b = ok ? (uint16_t)f2bf(src[(int64_t)(ci0 + r) * cout + co0 + c]) : (uint16_t)0;
        T[r][c] = b;
        if (Wb && ok) Wb[slot * per + (int64_t)(ci0 + r) * cout + co0 + c] = b;
    }
    __syncthreads();
    if (!Wb_t) return;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int r = i >> 6, c = i & 63;                       // r: output channel, c: input channel
        if (co0 + r < cout && ci0 + c < cin) Wb_t[slot * per + (int64_t)(co0 + r) * cin + ci0 + c] = T[c][r];
    }
}
extern "C" int cg3d_spconv_prep_weights_bf16_multi(const float *W0, const float *const *Ws, uint16_t *Wb_t, uint16_t *Wb,
                                                   int32_t G, int64_t slots_per, int32_t cin, int32_t cout,
                                                   cg3d_stream_t stream) {
    if (G < 1 || slots_per < 0 || cin < 1 || cout < 1 || (!W0 && !Ws) || (!Wb_t && !Wb)) return CG3D_ERR_ARG;
    const int64_t slots = (int64_t)G * slots_per;
    if (slots == 0) return CG3D_OK;
    const int32_t ct = cg3d_divup(cin, 64), ot = cg3d_divup(cout, 64);
    if (slots > 0x7fffffffll || (int64_t)ct * ot > 65535) return CG3D_ERR_RANGE;
    hipLaunchKernelGGL(k_prep_weights, dim3((unsigned)slots, (unsigned)(ct * ot)), dim3(256), 0, cg3d_hs(stream), W0, Ws,
                       Wb_t, Wb, slots_per, cin, cout, ot);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
// Table-driven form: ONE launch for the bf16 copies of every convolution weight of the model (a training step used to
// spend 55 launches of 15 us on them).  Row b of the table describes the 64 x 64 tile block b converts:
//   { address of the fp32 slot [cin][cout], address of its transposed bf16 copy or 0, address of its plain bf16 copy or
//     0, cin, cout, tile index = ci_tile * ceil(cout / 64) + co_tile }
// PREP_ROWS table rows per workgroup: one 64 x 64 tile is 16 KB in and 8-48 KB out -- at one tile per workgroup the launch was
// 48 495 workgroups of a few hundred nanoseconds each for the model's 126.5 M parameters (492 us, 3.6 TB/s against the 5.8 TB/s
// of the optimiser's pass over the same parameters).  Four tiles per workgroup: 463 us; sixteen: 488 -- what is left is the
// 16-byte pieces of the transposed / fragment-order copies, not the workgroup count.
#define PREP_ROWS 4
__device__ static void prep_weights_tile(const int64_t *__restrict__ row, uint16_t (*T)[64][72]);
__global__ __launch_bounds__(256) void k_prep_weights_table(const int64_t *__restrict__ table, int64_t nrows) {
    __shared__ __attribute__((aligned(16))) uint16_t T[2][64][72];       // 144-byte rows: 16-byte aligned pieces; [1]: the lo parts (split rows)
    for (int q = 0; q < PREP_ROWS; q++) {
        const int64_t b = (int64_t)blockIdx.x * PREP_ROWS + q;
        if (b >= nrows) return;                                           // (uniform)
        if (q) __syncthreads();                                           // the previous tile's readers are done with T
        prep_weights_tile(table + b * 6, T);
    }
}
__device__ static void prep_weights_tile(const int64_t *__restrict__ row, uint16_t (*T)[64][72]) {
    const float *src = reinterpret_cast<const float *>(row[0]);
    uint16_t *Wb_t = reinterpret_cast<uint16_t *>(row[1]);
    uint16_t *Wb = reinterpret_cast<uint16_t *>(row[2]);
    const int cin = (int)row[3], cout = (int)row[4], tile = (int)(row[5] & 0xfffffff);
    const bool frag_t = (row[5] >> 30) & 1, frag_p = (row[5] >> 29) & 1;       // copies in MFMA fragment order (cg3d_spconv_tile_fwd)
    // split row (bit 28): the copies are the three-part operands of cg3d_spconv_prep_weights_split -- the transposed copy over
    // the contraction index part * cin + ci, the plain copy over part * cout + co (parts 0, 1: bf16(W); part 2: the remainder)
    const bool split = (row[5] >> 28) & 1;
    const int NP = split ? 3 : 1;
    const int kt = NP * cin, kp = NP * cout;                                   // contraction lengths of the two copies
    const int co_tiles = (cout + 63) / 64;
    const int ci0 = (tile / co_tiles) * 64, co0 = (tile % co_tiles) * 64;
    // Full tiles of 16-byte aligned tensors (every layer of the model but the 3- / 6- / 18-channel ends): a thread moves 8
    // consecutive elements -- two float4 in, one 16-byte piece out (a piece of 8 consecutive contraction indices is contiguous
    // in the plain, the transposed and both fragment layouts).  The element-wise form below wrote 2 bytes per lane: 0.45 ms per
    // step for the model's 1 GB of weights and copies.
    const bool fast = ci0 + 64 <= cin && co0 + 64 <= cout && !(cout & 7) && !(cin & 7) &&
                      !(((uintptr_t)src | (uintptr_t)Wb | (uintptr_t)Wb_t) & 15);
    if (fast) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int piece = threadIdx.x + 256 * j, r = piece >> 3, c8 = (piece & 7) * 8;     // r: input channel, c8: output channel
            const float4 *sp = reinterpret_cast<const float4 *>(src + (int64_t)(ci0 + r) * cout + co0 + c8);
            const float4 a = sp[0], b = sp[1];
            const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            uint32_t h[8], l[8];
#pragma unroll
            for (int e = 0; e < 8; e++) {
                if (split) split_bf(f[e], h[e], l[e]);
                else { h[e] = f2bf(f[e]); l[e] = 0u; }
            }
            const uint4 o = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
            const uint4 ol = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
            *reinterpret_cast<uint4 *>(&T[0][r][c8]) = o;
            if (split) *reinterpret_cast<uint4 *>(&T[1][r][c8]) = ol;
            if (Wb) {
                for (int part = 0; part < NP; part++) {
                    const int kc = part * cout + co0 + c8;
                    *reinterpret_cast<uint4 *>(Wb + (frag_p ? cg3d_frag_index(ci0 + r, kc, kp) : (int64_t)(ci0 + r) * kp + kc)) = part < 2 ? o : ol;
                }
            }
        }
        __syncthreads();
        if (!Wb_t) return;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            // lanes of a wave: r (output channel) fastest, so the eight 2-byte LDS reads of a step hit 32 consecutive columns
            const int piece = threadIdx.x + 256 * j, r = piece & 63, c8 = (piece >> 6) * 8;    // c8: input channel
            for (int part = 0; part < NP; part++) {
                const int sel = part < 2 ? 0 : 1;
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; e++) w[e] = (uint32_t)T[sel][c8 + 2 * e][r] | ((uint32_t)T[sel][c8 + 2 * e + 1][r] << 16);
                const int kc = part * cin + ci0 + c8;
                *reinterpret_cast<uint4 *>(Wb_t + (frag_t ? cg3d_frag_index(co0 + r, kc, kt) : (int64_t)(co0 + r) * kt + kc)) =
                    make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        return;
    }
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int r = i >> 6, c = i & 63;                       // r: input channel, c: output channel
        const bool ok = ci0 + r < cin && co0 + c < cout;
        uint32_t hi = 0u, lo = 0u;
        if (ok) {
            const float v = src[(int64_t)(ci0 + r) * cout + co0 + c];
            if (split) split_bf(v, hi, lo); else hi = f2bf(v);
        }
        T[0][r][c] = (uint16_t)hi;
        T[1][r][c] = (uint16_t)lo;
        if (Wb && ok)
            for (int part = 0; part < NP; part++) {
                const int kc = part * cout + co0 + c;
                Wb[frag_p ? cg3d_frag_index(ci0 + r, kc, kp) : (int64_t)(ci0 + r) * kp + kc] = (uint16_t)(part < 2 ? hi : lo);
            }
    }
    __syncthreads();
    if (!Wb_t) return;
    for (int i = threadIdx.x; i < 4096; i += 256) {
        const int r = i >> 6, c = i & 63;                       // r: output channel, c: input channel
        if (co0 + r < cout && ci0 + c < cin)
            for (int part = 0; part < NP; part++) {
                const int kc = part * cin + ci0 + c;
                Wb_t[frag_t ? cg3d_frag_index(co0 + r, kc, kt) : (int64_t)(co0 + r) * kt + kc] = T[part < 2 ? 0 : 1][c][r];
            }
    }
}
extern "C" int cg3d_spconv_prep_weights_bf16_table(const int64_t *table, int64_t nrows, cg3d_stream_t stream) {
    if (nrows < 0 || nrows > 0x7fffffffll || (nrows > 0 && !table)) return CG3D_ERR_ARG;
    if (nrows == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_prep_weights_table, dim3((unsigned)((nrows + PREP_ROWS - 1) / PREP_ROWS)), dim3(256), 0, cg3d_hs(stream), table, nrows);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_spconv_prep_weights_bf16(const float *W, uint16_t *Wb, int64_t slots, int32_t cin, int32_t cout,
                                             cg3d_stream_t stream) {
    if (slots < 0) return CG3D_ERR_ARG;
    return cg3d_spconv_prep_weights_bf16_multi(W, nullptr, Wb, nullptr, 1, slots, cin, cout, stream);
}

template <int NT, typename XT>
__global__ __launch_bounds__(256, 2) void k_spconv_pairs_bf16(const XT *__restrict__ X, const uint16_t *__restrict__ Wb,
                                                              const int32_t *__restrict__ pin,
                                                              const int32_t *__restrict__ pout,
                                                              const int32_t *__restrict__ seg, float *__restrict__ Y,
                                                              int32_t cin, int32_t cout) {
    constexpr int CT = NT * 32;
    constexpr int KC = 64;                 // input channels per chunk
    constexpr int LP = KC + 8;             // padded LDS row (bf16 elements): 144 B, conflict-free b128 reads
    __shared__ uint16_t As[128 * LP];
    __shared__ uint16_t Ws[CT * LP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kg = lane >> 5;
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;               // placeholder entries keep the XCD alignment of a segment table (uniform branch)
    const int n0 = blockIdx.y * CT;
    const int local = wave * 32 + r;
    const bool valid = local < count;
    const int32_t irow = valid ? pin[start + local] : -1;
    const int32_t orow = valid ? pout[start + local] : -1;
    const bool wave_active = wave * 32 < count;
    const uint16_t *wk = Wb + (int64_t)k * cin * cout;     // [cout][cin]
    uint16_t *Aw = &As[wave * 32 * LP];

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    int32_t grow[8];
#pragma unroll
    for (int i = 0; i < 8; i++) grow[i] = __shfl(irow, 4 * i + (lane >> 4));
    const int gcol = (lane & 15) * 4;

    typedef typename Row4<XT>::T RowT;
    RowT areg[8];
    uint4 wreg[NT];
    auto issue_loads = [&](int32_t c0) {
#pragma unroll
        for (int i = 0; i < NT; i++) {                  // W tile: CT cols x 64 ch bf16 = CT*8 pieces of 16 B
            const int idx = tid + i * 256;
            const int col = idx >> 3, piece = idx & 7;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (n0 + col < cout && c0 + piece * 8 < cin)
                v = *reinterpret_cast<const uint4 *>(wk + (int64_t)(n0 + col) * cin + c0 + piece * 8);
            wreg[i] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            RowT v = Row4<XT>::zero();
            if (grow[i] >= 0 && c0 + gcol < cin)
                v = *reinterpret_cast<const RowT *>(X + (int64_t)grow[i] * cin + c0 + gcol);
            areg[i] = v;
        }
    };
    issue_loads(0);
    for (int32_t c0 = 0; c0 < cin; c0 += KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int idx = tid + i * 256;
            *reinterpret_cast<uint4 *>(&Ws[(idx >> 3) * LP + (idx & 7) * 8]) = wreg[i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            *reinterpret_cast<uint2 *>(&Aw[(4 * i + (lane >> 4)) * LP + gcol]) = Row4<XT>::bf(areg[i]);
        __syncthreads();
        if (c0 + KC < cin) issue_loads(c0 + KC);
        if (wave_active) {
#pragma unroll
            for (int ks = 0; ks < KC / 16; ks++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&Aw[r * LP + ks * 16 + kg * 8]);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Ws[(nt * 32 + r) * LP + ks * 16 + kg * 8]);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);
                }
            }
        }
    }
    if (!wave_active) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int rowl = (e & 3) + 8 * (e >> 2) + 4 * kg;
        const int32_t orow_e = __shfl(orow, rowl);
        if (orow_e < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int col = n0 + nt * 32 + r;
            if (col < cout) unsafeAtomicAdd(&Y[(int64_t)orow_e * cout + col], acc[nt][e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Output-stationary bf16 form: one workgroup owns 128 output rows x CT channels for ALL offsets and
// keeps the accumulators in registers; every output row is written once with a plain coalesced store
// (no atomics, no zero-fill, deterministic).  With bf16 MFMA the padding work on absent neighbours is
// free (the matrix pipe idles anyway) while the atomic scatter of the pair form was its biggest cost
// at the neighbourhood occupancies of the tensor-stride >= 4 maps.
// ---------------------------------------------------------------------------------------------
// Wave-specialised (512 threads): waves 4-7 only gather rows (global -> registers ->
// bf16 -> LDS), waves 0-3 run the matrix pipe and move the (L2-resident, already bf16) weight tile.  In the
// first, single-role version of this kernel every wave walked through load-issue, LDS staging and MFMA phases one after the other and
// with <= 2 waves per SIMD nothing overlapped them (measured: the phases added up); here the gather/convert/stage
// work of step s+1 runs on other waves while step s is on the matrix pipe; one barrier per step.
//   gather waves, step s:  stage the rows of step s+1 (gathered two steps ago; two register sets for the
//                          long-latency gather), issue the gather of step s+3.
//   matrix waves, step s:  issue the weight loads of step s+1, MFMAs of step s, stage those weights.
template <int NT, typename XT>
__global__ __launch_bounds__(512, 2) void k_spconv_implicit_bf16_ws(const XT *__restrict__ X,
                                                                    const uint16_t *__restrict__ Wb,
                                                                    const int32_t *__restrict__ nbr,
                                                                    const float *__restrict__ bias, float *__restrict__ Y,
                                                                    int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                                                                    const int32_t *__restrict__ tiles) {
    constexpr int CT = NT * 32;
    constexpr int KC = 64;
    constexpr int LP = KC + 8;
    __shared__ __attribute__((aligned(16))) uint16_t As[2][128 * LP];
    __shared__ __attribute__((aligned(16))) uint16_t Ws[2][CT * LP];
    __shared__ int32_t live_s[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kg = lane >> 5;
    const int n0 = blockIdx.y * CT;
    const int nchunk = (cin + KC - 1) / KC;
    // gridDim.z > 1: the offsets are split over z (small layers: more, shorter workgroups); partial sums meet in Y
    // through atomics, Y zero-filled by the launcher
    const int32_t kz0 = (int32_t)((int64_t)K * blockIdx.z / gridDim.z), kz1 = (int32_t)((int64_t)K * (blockIdx.z + 1) / gridDim.z);
    Wb += (int64_t)kz0 * cin * cout;
    nbr += (int64_t)kz0 * n_out;
    K = kz1 - kz0;
    const int nstep = K * nchunk;
    // the rows this workgroup owns: blockIdx.x * 128 .. +128, or -- grouped convolutions (one weight set per class
    // branch) -- a tile (group, first row, row count <= 128) of the table, which never straddles two groups
    int64_t tile_row0 = (int64_t)blockIdx.x * 128;
    int tile_rows = (int)(n_out - tile_row0 < 128 ? n_out - tile_row0 : 128);
    if (tiles) {
        Wb += (int64_t)tiles[blockIdx.x * 3] * K * cin * cout;
        tile_row0 = tiles[blockIdx.x * 3 + 1];
        tile_rows = tiles[blockIdx.x * 3 + 2];
    }

    if (wave >= 4) {
        // ------------------------------------------------------------------ gather waves
        const int pw = wave - 4;
        const int64_t row = tile_row0 + pw * 32 + r;
        const bool row_ok = pw * 32 + r < tile_rows;
        const int64_t row_c = row_ok ? row : tile_row0;
        typedef Gather<XT> G;
        typedef typename G::T RowT;
        const int gcol = (lane % G::LPR) * G::CH, grow = lane / G::LPR;
        struct ASet { RowT a[G::NL]; uint32_t amask; };
        ASet s0, s1;
        uint32_t gmask = 0;
        uint32_t goff[G::NL];
        auto set_rows = [&](int32_t idx) {
            gmask = 0;
#pragma unroll
            for (int i = 0; i < G::NL; i++) {
                const int32_t g = __shfl(idx, G::ROWS * i + grow);
                goff[i] = g >= 0 ? (uint32_t)g * (uint32_t)cin + (uint32_t)gcol : 0u;
                gmask |= g >= 0 ? (1u << i) : 0u;
            }
        };
        auto issue_a = [&](ASet &S, int32_t c0) {
            const bool cok = c0 + gcol < cin;
            S.amask = cok ? gmask : 0u;
            const uint32_t cadd = cok ? (uint32_t)c0 : 0u;
#pragma unroll
            for (int i = 0; i < G::NL; i++) S.a[i] = *reinterpret_cast<const RowT *>(X + (goff[i] + cadd));
        };
        auto commit = [&](const ASet &S, int buf, bool live) {
            if (lane == 0) live_s[buf][pw] = live ? 1 : 0;
            if (live) {
                uint16_t *Aw = &As[buf][pw * 32 * LP];
#pragma unroll
                for (int i = 0; i < G::NL; i++) {
                    const RowT v = (S.amask >> i) & 1u ? S.a[i] : G::zero();
                    G::stage(&Aw[(G::ROWS * i + grow) * LP + gcol], v);
                }
            }
        };
        auto load_idx = [&](int32_t k) -> int32_t { return nbr[(int64_t)(k < K ? k : K - 1) * n_out + row_c]; };
        auto masked = [&](int32_t v) -> int32_t { return row_ok ? v : -1; };
        int32_t kq = 0, cq = 0;                       // (kq, cq): the step gathered next
        auto advance = [&]() { cq++; if (cq == nchunk) { cq = 0; kq++; } };

        // prologue: step 0 staged; rows of steps 1 and 2 in flight
        int32_t idx = masked(load_idx(0));
        bool lg = __any(idx >= 0);
        set_rows(idx);
        issue_a(s0, 0);
        const bool L0 = lg;
        advance();                                    // step 1
        if (cq == 0) { idx = masked(load_idx(kq)); lg = __any(idx >= 0); set_rows(idx); }
        issue_a(s1, cq * KC);
        bool La = lg;                                 // liveness of the step staged next
        commit(s0, 0, L0);
        advance();                                    // step 2
        if (cq == 0) { idx = masked(load_idx(kq)); lg = __any(idx >= 0); set_rows(idx); }
        issue_a(s0, cq * KC);
        bool Lb = lg, Lc = lg;
        advance();                                    // step 3
        int32_t idx_pre = load_idx(kq);
        __syncthreads();

        auto iteration = [&](int st, ASet &S) {
            // the matrix waves are on step st; S holds the rows of step st+1
            commit(S, (st + 1) & 1, La);
            if (cq == 0) { idx = masked(idx_pre); lg = __any(idx >= 0); set_rows(idx); }
            issue_a(S, cq * KC);                      // rows of step st+3 (past the end: clamped, unused)
            Lc = lg;
            advance();
            idx_pre = load_idx(kq);                   // index column of the offset of step st+4
            La = Lb; Lb = Lc;
            __syncthreads();
        };
        for (int st = 0; st < nstep; st += 2) {
            iteration(st, s1);
            if (st + 1 < nstep) iteration(st + 1, s0);
        }
        return;
    }
    // ---------------------------------------------------------------------- matrix waves
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    uint4 w[NT];
    uint32_t wmask = 0;
    auto issue_w = [&](int32_t k, int32_t c0) {
        const uint16_t *wk = Wb + (int64_t)(k < K ? k : K - 1) * cin * cout;   // past the end: unused re-read
        wmask = 0;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int j = tid + i * 256;
            const int col = n0 + (j >> 3), cc = c0 + (j & 7) * 8;
            const bool ok = col < cout && cc < cin;
            const uint32_t off = ok ? (uint32_t)col * (uint32_t)cin + (uint32_t)cc : 0u;
            w[i] = *reinterpret_cast<const uint4 *>(wk + off);
            wmask |= ok ? (1u << i) : 0u;
        }
    };
    auto commit_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int j = tid + i * 256;
            const uint4 v = (wmask >> i) & 1u ? w[i] : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(&Ws[buf][(j >> 3) * LP + (j & 7) * 8]) = v;
        }
    };
    int32_t kw = 0, cw = 0;
    issue_w(0, 0);
    commit_w(0);
    __syncthreads();                                               // step 0 staged
    for (int st = 0; st < nstep; st++) {
        const int buf = st & 1;
        cw++;
        if (cw == nchunk) { cw = 0; kw++; }
        issue_w(kw, cw * KC);                                      // weights of step st+1, in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        if (live_s[buf][wave]) {
            const uint16_t *Aw = &As[buf][wave * 32 * LP];
#pragma unroll
            for (int ks = 0; ks < KC / 16; ks++) {
                const bf16x8 a = *reinterpret_cast<const bf16x8 *>(&Aw[r * LP + ks * 16 + kg * 8]);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Ws[buf][(nt * 32 + r) * LP + ks * 16 + kg * 8]);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);
                }
            }
        }
        commit_w(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = n0 + nt * 32 + r;
        if (col >= cout) continue;
        const float bv = bias && blockIdx.z == 0 ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int lrow = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (lrow >= tile_rows) continue;
            if (gridDim.z == 1) Y[(tile_row0 + lrow) * cout + col] = acc[nt][e] + bv;
            else unsafeAtomicAdd(&Y[(tile_row0 + lrow) * cout + col], acc[nt][e] + bv);
        }
    }
}

// Direct-operand form for bf16 rows (precision 2): a wave loads the MFMA A fragments of its 32 rows straight from the
// row matrix into registers -- lane (r, kg) owns row r and the 32 channels [kg*32, kg*32+32) of the chunk as four 16-byte
// pieces (the contraction order over channels is free, the weight fragments are read from LDS in the same order) --
// three register sets deep, so there are no gather waves, no row staging through LDS and no liveness flags; only the
// weight tile goes through LDS (double buffer, one barrier per step among 4 waves).  256 threads and 37 KB of LDS per
// workgroup: three independent workgroups per CU instead of two barrier-coupled 8-wave ones.
template <int NT>
__global__ __launch_bounds__(256, 3) void k_spconv_implicit_bf16_ad(const uint16_t *__restrict__ X,
                                                                    const uint16_t *__restrict__ Wb,
                                                                    const int32_t *__restrict__ nbr,
                                                                    const float *__restrict__ bias, float *__restrict__ Y,
                                                                    int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                                                                    const int32_t *__restrict__ tiles) {
    constexpr int CT = NT * 32;
    constexpr int KC = 64;
    constexpr int LP = KC + 8;
    __shared__ __attribute__((aligned(16))) uint16_t Ws[2][CT * LP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kg = lane >> 5;
    const int n0 = blockIdx.y * CT;
    const int nchunk = (cin + KC - 1) / KC;
    // gridDim.z > 1: the offsets are split over z (small layers: more, shorter workgroups); partial sums meet in Y
    // through atomics, Y zero-filled by the launcher
    const int32_t kz0 = (int32_t)((int64_t)K * blockIdx.z / gridDim.z), kz1 = (int32_t)((int64_t)K * (blockIdx.z + 1) / gridDim.z);
    Wb += (int64_t)kz0 * cin * cout;
    nbr += (int64_t)kz0 * n_out;
    K = kz1 - kz0;
    const int nstep = K * nchunk;
    // XCD-aware tile order: workgroup x runs on XCD x % 8 (own L2), so XCD c takes the c-th contiguous eighth of the
    // tiles and walks it in order -- the halo rows of neighbouring tiles are then shared through ONE L2
    const unsigned t_lo = gridDim.x >> 3, t_rem = gridDim.x & 7, xcd = blockIdx.x & 7;
    const unsigned bx = xcd * t_lo + (xcd < t_rem ? xcd : t_rem) + (blockIdx.x >> 3);
    int64_t tile_row0 = (int64_t)bx * 128;
    int tile_rows = (int)(n_out - tile_row0 < 128 ? n_out - tile_row0 : 128);
    if (tiles) {
        Wb += (int64_t)tiles[bx * 3] * K * cin * cout;
        tile_row0 = tiles[bx * 3 + 1];
        tile_rows = tiles[bx * 3 + 2];
    }
    const bool row_ok = wave * 32 + r < tile_rows;
    const int64_t row_c = row_ok ? tile_row0 + wave * 32 + r : tile_row0;
    const bool partial = (cin & (KC - 1)) != 0;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    // ---- rows: three register sets (step s on the matrix pipe, s+1 and s+2 in flight)
    struct ASet { uint4 a[4]; uint32_t ok; bool live; };
    ASet s0, s1, s2;
    uint32_t goff = 0;                 // element offset of this lane's 32 channels inside the current neighbour row
    bool gvalid = false, glive = false;
    auto load_idx = [&](int32_t k) -> int32_t { return nbr[(int64_t)(k < K ? k : K - 1) * n_out + row_c]; };
    auto set_row = [&](int32_t raw) {
        const int32_t g = row_ok ? raw : -1;
        gvalid = g >= 0;
        goff = gvalid ? (uint32_t)g * (uint32_t)cin + (uint32_t)(kg * 32) : (uint32_t)(kg * 32);
        glive = __any(gvalid);
    };
    auto issue_a = [&](ASet &S, int32_t c0) {
        S.live = glive;
        S.ok = gvalid ? 0xFu : 0u;
        if (partial) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool cok = c0 + kg * 32 + i * 8 < cin;
                if (!cok) S.ok &= ~(1u << i);
                S.a[i] = *reinterpret_cast<const uint4 *>(X + (goff + (cok ? (uint32_t)(c0 + i * 8) : 0u)));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) S.a[i] = *reinterpret_cast<const uint4 *>(X + (goff + (uint32_t)(c0 + i * 8)));
        }
    };
    int32_t kq = 0, cq = 0;                       // (kq, cq): the step whose rows are requested next
    auto advance = [&]() { cq++; if (cq == nchunk) { cq = 0; kq++; } };

    // ---- weight tile: L2 -> registers -> LDS, one step ahead
    uint4 w[NT];
    uint32_t wmask = 0;
    auto issue_w = [&](int32_t k, int32_t c0) {
        const uint16_t *wk = Wb + (int64_t)(k < K ? k : K - 1) * cin * cout;   // past the end: unused re-read
        wmask = 0;
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int j = tid + i * 256;
            const int col = n0 + (j >> 3), cc = c0 + (j & 7) * 8;
            const bool ok = col < cout && cc < cin;
            const uint32_t off = ok ? (uint32_t)col * (uint32_t)cin + (uint32_t)cc : 0u;
            w[i] = *reinterpret_cast<const uint4 *>(wk + off);
            wmask |= ok ? (1u << i) : 0u;
        }
    };
    auto commit_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NT; i++) {
            const int j = tid + i * 256;
            const uint4 v = (wmask >> i) & 1u ? w[i] : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4 *>(&Ws[buf][(j >> 3) * LP + (j & 7) * 8]) = v;
        }
    };

    // prologue: rows of steps 0 and 1 requested, weights of step 0 staged
    int32_t idx_pre;
    set_row(load_idx(0));
    issue_a(s0, 0);
    advance();                                    // step 1
    if (cq == 0) set_row(load_idx(kq));
    issue_a(s1, cq * KC);
    advance();                                    // step 2
    idx_pre = load_idx(kq);
    int32_t kw = 0, cw = 0;
    issue_w(0, 0);
    commit_w(0);
    __syncthreads();

    auto step = [&](int st, const ASet &Cur, ASet &Far) {
        const int buf = st & 1;
        cw++;
        if (cw == nchunk) { cw = 0; kw++; }
        issue_w(kw, cw * KC);                     // weights of step st+1
        if (cq == 0) set_row(idx_pre);
        issue_a(Far, cq * KC);                    // rows of step st+2 (past the end: clamped, unused)
        advance();
        idx_pre = load_idx(kq);                   // index column of the offset of step st+3
        if (Cur.live) {
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const uint4 av = (Cur.ok >> ks) & 1u ? Cur.a[ks] : make_uint4(0u, 0u, 0u, 0u);
                const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    const bf16x8 b = *reinterpret_cast<const bf16x8 *>(&Ws[buf][(nt * 32 + r) * LP + kg * 32 + ks * 8]);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nt], 0, 0, 0);
                }
            }
        }
        commit_w(buf ^ 1);
        __syncthreads();
    };
    for (int st = 0; st < nstep; st += 3) {
        step(st, s0, s2);
        if (st + 1 < nstep) step(st + 1, s1, s0);
        if (st + 2 < nstep) step(st + 2, s2, s1);
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const int col = n0 + nt * 32 + r;
        if (col >= cout) continue;
        const float bv = bias && blockIdx.z == 0 ? bias[col] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
            const int lrow = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
            if (lrow >= tile_rows) continue;
            if (gridDim.z == 1) Y[(tile_row0 + lrow) * cout + col] = acc[nt][e] + bv;
            else unsafeAtomicAdd(&Y[(tile_row0 + lrow) * cout + col], acc[nt][e] + bv);
        }
    }
}

// Generic path (any cin / alignment; also the 3-channel input layer with KH = 2): per-lane fragment
// gather straight to registers, W tile through LDS.
template <int NT, int KH, bool VEC4>
__global__ __launch_bounds__(256) void k_spconv_pairs(const float *__restrict__ X, const float *__restrict__ W,
                                                      const int32_t *__restrict__ pin,
                                                      const int32_t *__restrict__ pout,
                                                      const int32_t *__restrict__ seg, float *__restrict__ Y,
                                                      int32_t cin, int32_t cout) {
    constexpr int CT = NT * 32;
    __shared__ float Ws[2 * KH * CT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;               // placeholder entries keep the XCD alignment of a segment table (uniform branch)
    const int n0 = blockIdx.y * CT;
    const int local = wave * 32 + r;
    const bool valid = local < count;
    const int32_t irow = valid ? pin[start + local] : -1;
    const int32_t orow = valid ? pout[start + local] : -1;
    const bool wave_active = wave * 32 < count;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    const float *xrow = X + (int64_t)(irow < 0 ? 0 : irow) * cin;
    const float *wk = W + (int64_t)k * cin * cout;

    for (int32_t c0 = 0; c0 < cin; c0 += 2 * KH) {
        __syncthreads();
        for (int i = tid; i < 2 * KH * CT; i += 256) {
            const int row = i / CT, c = i % CT;
            Ws[i] = (c0 + row < cin && n0 + c < cout) ? wk[(int64_t)(c0 + row) * cout + n0 + c] : 0.f;
        }
        const int cb = c0 + h * KH;
        float a[KH];
        if (VEC4) {
#pragma unroll
            for (int t = 0; t < KH; t += 4) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (irow >= 0 && cb + t < cin) v = *reinterpret_cast<const float4 *>(xrow + cb + t);
                a[t] = v.x; a[t + 1] = v.y; a[t + 2] = v.z; a[t + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int t = 0; t < KH; t++) a[t] = (irow >= 0 && cb + t < cin) ? xrow[cb + t] : 0.f;
        }
        __syncthreads();
        if (wave_active) {
#pragma unroll
            for (int t = 0; t < KH; t++) {
                const float *wr = &Ws[(h * KH + t) * CT + r];
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], wr[nt * 32], acc[nt], 0, 0, 0);
            }
        }
    }
    if (!wave_active) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const int rowl = (e & 3) + 8 * (e >> 2) + 4 * h;
        const int32_t orow_e = __shfl(orow, rowl);
        if (orow_e < 0) continue;
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            const int col = n0 + nt * 32 + r;
            if (col < cout) unsafeAtomicAdd(&Y[(int64_t)orow_e * cout + col], acc[nt][e]);
        }
    }
}

__global__ void k_init_rows(float *__restrict__ Y, const float *__restrict__ bias, int64_t total, int32_t cout) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < total) Y[t] = bias[t % cout];
}

extern "C" int cg3d_spconv_pairs_fwd(const float *X, const float *W, const int32_t *pair_in, const int32_t *pair_out,
                                     const int32_t *seg, int64_t nseg, const float *bias, float *Y, int64_t n_out,
                                     int32_t cin, int32_t cout, int32_t precision, int32_t accumulate,
                                     cg3d_stream_t stream) {
    if (n_out < 0 || nseg < 0 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    if (precision < 0 || precision > 2) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (n_out == 0) return CG3D_OK;
    const int64_t total = n_out * cout;
    if (precision >= 1 && (cin % 8 != 0 || ((uintptr_t)X & 15) || ((uintptr_t)W & 15))) return CG3D_ERR_ARG;
    if (!accumulate) {
        if (bias) hipLaunchKernelGGL(k_init_rows, dim3((unsigned)cg3d_divup(total, 256)), dim3(256), 0, s, Y, bias, total, cout);
        else if (hipMemsetAsync(Y, 0, total * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    }
    if (nseg == 0) return CG3D_OK;
    if (precision >= 1) {     // W is the prepared bf16 [slot][cout][cin] buffer (cg3d_spconv_prep_weights_bf16)
        const uint16_t *Wb = reinterpret_cast<const uint16_t *>(W);
#define LAUNCH_BF(NT, XT)                                                                                          \
    hipLaunchKernelGGL((k_spconv_pairs_bf16<NT, XT>), dim3((unsigned)nseg, (unsigned)cg3d_divup(cout, NT * 32)), dim3(256), 0, \
                       s, reinterpret_cast<const XT *>(X), Wb, pair_in, pair_out, seg, Y, cin, cout)
        if (precision == 1) {
            if (cout > 64) LAUNCH_BF(4, float); else if (cout > 32) LAUNCH_BF(2, float); else LAUNCH_BF(1, float);
        } else {              // X is bf16 [n_in][cin] (cg3d_to_bf16)
            if (cout > 64) LAUNCH_BF(4, uint16_t); else if (cout > 32) LAUNCH_BF(2, uint16_t); else LAUNCH_BF(1, uint16_t);
        }
#undef LAUNCH_BF
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    const bool aligned = (((uintptr_t)X & 15) == 0) && (((uintptr_t)W & 15) == 0);
#define LAUNCH_LDS(NT)                                                                                             \
    hipLaunchKernelGGL((k_spconv_pairs_lds<NT>), dim3((unsigned)nseg, (unsigned)cg3d_divup(cout, NT * 32)), dim3(256),    \
                       0, s, X, W, pair_in, pair_out, seg, Y, cin, cout)
#define LAUNCH(NT, KH, V)                                                                                     \
    hipLaunchKernelGGL((k_spconv_pairs<NT, KH, V>), dim3((unsigned)nseg, (unsigned)cg3d_divup(cout, NT * 32)),   \
                       dim3(256), 0, s, X, W, pair_in, pair_out, seg, Y, cin, cout)
    if (cin % 4 == 0 && cin >= 16 && aligned) {
        if (cout > 64) LAUNCH_LDS(4);
        else if (cout > 32) LAUNCH_LDS(2);
        else LAUNCH_LDS(1);
    } else if (cin <= 4) {
        if (cout > 64) LAUNCH(4, 2, false); else LAUNCH(2, 2, false);
    } else if (cin % 4 == 0 && aligned) {
        if (cout > 64) LAUNCH(4, 32, true); else if (cout > 32) LAUNCH(2, 32, true); else LAUNCH(1, 32, true);
    } else {
        if (cout > 64) LAUNCH(4, 32, false); else if (cout > 32) LAUNCH(2, 32, false); else LAUNCH(1, 32, false);
    }
#undef LAUNCH
#undef LAUNCH_LDS
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// weight gradient over compacted pairs: every MFMA step contracts two real pairs.
// The four operand loads of a step (X[ci0+r], X[ci0+32+r], dY[co0+r], dY[co0+32+r]; each a full
// 128-byte line per half-wave) are issued a whole batch of 8 steps AHEAD of the MFMAs that consume
// them (register double buffer, 64 loads in flight per wave) -- the plain loop waited an L2/HBM
// round trip per step.
#define WG_B 8
__global__ __launch_bounds__(256) void k_spconv_pairs_wgrad(const float *__restrict__ X, const float *__restrict__ dY,
                                                            const int32_t *__restrict__ pin,
                                                            const int32_t *__restrict__ pout,
                                                            const int32_t *__restrict__ seg, float *__restrict__ dW,
                                                            int32_t cin, int32_t cout, int32_t co_tiles) {
    __shared__ float tile[64 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;               // placeholder entries keep the XCD alignment of a segment table (uniform branch)
    const int ci0 = (blockIdx.y / co_tiles) * 64, co0 = (blockIdx.y % co_tiles) * 64;

    for (int i = threadIdx.x; i < 64 * 64; i += 256) tile[i] = 0.f;
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const bool ci_ok[2] = {ci0 + r < cin, ci0 + 32 + r < cin};
    const bool co_ok[2] = {co0 + r < cout, co0 + 32 + r < cout};
    // channel-tail lanes read a clamped in-row address and are zeroed afterwards
    const int ca0 = ci_ok[0] ? ci0 + r : 0, ca1 = ci_ok[1] ? ci0 + 32 + r : 0;
    const int cb0 = co_ok[0] ? co0 + r : 0, cb1 = co_ok[1] ? co0 + 32 + r : 0;
    const float *Xb = X, *Db = dY;
    const int off_a0 = ca0, off_a1 = ca1, off_b0 = cb0, off_b1 = cb1;

    for (int32_t g = wave * 64; g < count; g += 256) {
        const int32_t p = g + lane;
        const int32_t my_in = (p < count) ? pin[start + p] : -1;
        const int32_t my_out = (p < count) ? pout[start + p] : -1;
        const int nbatch = ((count - g >= 64 ? 64 : count - g) + 2 * WG_B - 1) / (2 * WG_B);
        float va[2][WG_B][2], vb[2][WG_B][2];
        auto issue = [&](int buf, int t0) {
#pragma unroll
            for (int tt = 0; tt < WG_B; tt++) {
                const int src = 2 * (t0 + tt) + h;
                const int32_t ii = __shfl(my_in, src);
                const int32_t oo = __shfl(my_out, src);
                // UNCONDITIONAL loads from clamped (always valid) addresses, masked afterwards: a load
                // inside an exec-masked branch makes hipcc fall back to vmcnt(0) waits (no pipelining)
                const float *xr = Xb + (int64_t)(ii < 0 ? 0 : ii) * cin;
                const float *dr = Db + (int64_t)(oo < 0 ? 0 : oo) * cout;
                va[buf][tt][0] = xr[off_a0]; va[buf][tt][1] = xr[off_a1];     // raw; masked at use
                vb[buf][tt][0] = dr[off_b0]; vb[buf][tt][1] = dr[off_b1];
            }
        };
        issue(0, 0);
#pragma unroll
        for (int bt = 0; bt < 32 / WG_B; bt++) {
            if (bt < nbatch) {
                if (bt + 1 < nbatch) issue((bt + 1) & 1, (bt + 1) * WG_B);
                __builtin_amdgcn_sched_barrier(0);   // keep the next batch's loads ahead of this batch's MFMAs
#pragma unroll
                for (int tt = 0; tt < WG_B; tt++) {
                    const bool ok = __shfl(my_in, 2 * (bt * WG_B + tt) + h) >= 0;
                    const float a0 = (ok && ci_ok[0]) ? va[bt & 1][tt][0] : 0.f;
                    const float a1 = (ok && ci_ok[1]) ? va[bt & 1][tt][1] : 0.f;
                    const float b0 = co_ok[0] ? vb[bt & 1][tt][0] : 0.f;
                    const float b1 = co_ok[1] ? vb[bt & 1][tt][1] : 0.f;
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                int tr = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                int tc = j * 32 + r;
                atomicAdd(&tile[tr * 64 + tc], acc[i][j][e]);
            }
    __syncthreads();
    float *dst = dW + (int64_t)k * cin * cout;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        int ci = ci0 + (i >> 6), co = co0 + (i & 63);
        if (ci < cin && co < cout) unsafeAtomicAdd(&dst[(int64_t)ci * cout + co], tile[i]);
    }
}

// Wide-channel weight gradient (cin, cout >= 128): one workgroup owns a 128 x 128 (ci x co) tile; its four
// waves are the four 64 x 64 quadrants and all consume the SAME pairs, which are staged once per workgroup
// in LDS (32 pairs x 128 channels of X and of dY per stage, coalesced 512-byte row reads, register
// prefetch of the next stage).  The 64 x 64 kernel above re-reads every gathered row once per tile of the
// other operand: that traffic (2 KB per pair on a 128 x 128 layer), not the MFMA pipe, bounded it.
#define WT_S 32            // pairs per stage
#define WT_LD 132          // padded LDS row (floats)
__global__ __launch_bounds__(256, 2) void k_spconv_pairs_wgrad_t128(const float *__restrict__ X,
                                                                    const float *__restrict__ dY,
                                                                    const int32_t *__restrict__ pin,
                                                                    const int32_t *__restrict__ pout,
                                                                    const int32_t *__restrict__ seg,
                                                                    float *__restrict__ dW, int32_t cin, int32_t cout,
                                                                    int32_t co_tiles) {
    __shared__ float Xs[2][WT_S * WT_LD];
    __shared__ float Ds[2][WT_S * WT_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;               // placeholder entries keep the XCD alignment of a segment table (uniform branch)
    const int ci0 = (blockIdx.y / co_tiles) * 128, co0 = (blockIdx.y % co_tiles) * 128;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    // staging: thread owns float4 #(tid + 256*i) of the 32 x 128 tile: pair = idx >> 5, column = (idx & 31) * 4
    const int c4 = (tid & 31) * 4;
    const bool xin = ci0 + c4 < cin, din = co0 + c4 < cout;       // cin, cout are multiples of 4 here
    float4 xr[4], dr[4];
    auto issue = [&](int32_t p0) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pr = (tid >> 5) + 8 * i;
            const int32_t p = p0 + pr;
            // clamped, unconditional addresses (masked below) keep the loads countable for vmcnt
            const int32_t pp = p < count ? p : count - 1;
            const int32_t ii = pin[start + pp], oo = pout[start + pp];
            float4 a = *reinterpret_cast<const float4 *>(X + (int64_t)ii * cin + (xin ? ci0 + c4 : 0));
            float4 b = *reinterpret_cast<const float4 *>(dY + (int64_t)oo * cout + (din ? co0 + c4 : 0));
            const bool ok = p < count;
            xr[i] = (ok && xin) ? a : make_float4(0.f, 0.f, 0.f, 0.f);
            dr[i] = (ok && din) ? b : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int pr = (tid >> 5) + 8 * i;
            *reinterpret_cast<float4 *>(&Xs[buf][pr * WT_LD + c4]) = xr[i];
            *reinterpret_cast<float4 *>(&Ds[buf][pr * WT_LD + c4]) = dr[i];
        }
    };
    const int nstage = (count + WT_S - 1) / WT_S;
    issue(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nstage; st++) {
        const int buf = st & 1;
        if (st + 1 < nstage) issue((st + 1) * WT_S);            // in flight during the MFMAs
        const float *xa = &Xs[buf][h * WT_LD + wi * 64 + r];
        const float *db = &Ds[buf][h * WT_LD + wj * 64 + r];
#pragma unroll
        for (int t = 0; t < WT_S / 2; t++) {
            const float a0 = xa[2 * t * WT_LD], a1 = xa[2 * t * WT_LD + 32];
            const float b0 = db[2 * t * WT_LD], b1 = db[2 * t * WT_LD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (st + 1 < nstage) commit(buf ^ 1);                   // the other buffer was last read two stages ago
        __syncthreads();
    }
    float *dst = dW + (int64_t)k * cin * cout;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int ci = ci0 + wi * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int co = co0 + wj * 64 + j * 32 + r;
                if (ci < cin && co < cout) unsafeAtomicAdd(&dst[(int64_t)ci * cout + co], acc[i][j][e]);
            }
}

// bf16-operand weight gradient (precision 1): the same (segment x channel-tile) decomposition as the fp32 kernels,
// 64 pairs per stage.  The contraction runs over pairs, which is the strided dimension of the row-major
// operands, so each thread transposes a 4 pair x 4 channel block in registers on the way into LDS
// ([channel][pair] bf16, 16-byte granules XOR-swizzled by channel>>4 so that the transposing 8-byte writes
// spread over the banks); the MFMA fragments are then single 16-byte LDS reads.
#define WB_S 64            // pairs per stage
#define WB_LD 72           // padded LDS row (bf16): 144 B
// 4 pairs x 4 channels -> four [channel][4 pairs] bf16 rows (8 bytes each)
__device__ static inline void transpose4x4bf(const float4 *v, uint2 *o) {
    o[0] = pack4bf(make_float4(v[0].x, v[1].x, v[2].x, v[3].x));
    o[1] = pack4bf(make_float4(v[0].y, v[1].y, v[2].y, v[3].y));
    o[2] = pack4bf(make_float4(v[0].z, v[1].z, v[2].z, v[3].z));
    o[3] = pack4bf(make_float4(v[0].w, v[1].w, v[2].w, v[3].w));
}

template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void k_spconv_pairs_wgrad_bf16(const float *__restrict__ X,
                                                                    const float *__restrict__ dY,
                                                                    const int32_t *__restrict__ pin,
                                                                    const int32_t *__restrict__ pout,
                                                                    const int32_t *__restrict__ seg,
                                                                    float *__restrict__ dW, int32_t cin, int32_t cout,
                                                                    int32_t co_tiles) {
    __shared__ __attribute__((aligned(16))) uint16_t Xs[2][TM * WB_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Ds[2][TN * WB_LD];
    typedef float ST;
    typedef Row4<float>::T RowT;
    constexpr int XT = TM / 4, DT = TN / 4;              // threads per gathered row
    constexpr int XP = 16 / (256 / XT), DP = 16 / (256 / DT);   // passes over the 16 pair quads of a stage
    constexpr int MI = TM / 64, NJ = TN / 64;            // 32 x 32 tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;               // placeholder entries keep the XCD alignment of a segment table (uniform branch)
    const int ci0 = (blockIdx.y / co_tiles) * TM, co0 = (blockIdx.y % co_tiles) * TN;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    const int xc4 = (tid % XT) * 4, dc4 = (tid % DT) * 4;
    const bool xin = ci0 + xc4 < cin, din = co0 + dc4 < cout;       // cin, cout are multiples of 4 here
    const ST *xbase = X + (xin ? ci0 + xc4 : 0);
    const ST *dbase = dY + (din ? co0 + dc4 : 0);
    RowT xr[XP][4], dr[DP][4];
    auto issue = [&](int32_t p0) {
#pragma unroll
        for (int ps = 0; ps < XP; ps++) {
            const int q = ps * (256 / XT) + tid / XT;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int32_t p = p0 + 4 * q + i;
                const int32_t pp = p < count ? p : count - 1;      // clamped, unconditional (masked below)
                const RowT a = *reinterpret_cast<const RowT *>(xbase + (int64_t)pin[start + pp] * cin);
                xr[ps][i] = (p < count && xin) ? a : Row4<ST>::zero();
            }
        }
#pragma unroll
        for (int ps = 0; ps < DP; ps++) {
            const int q = ps * (256 / DT) + tid / DT;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int32_t p = p0 + 4 * q + i;
                const int32_t pp = p < count ? p : count - 1;
                const RowT b = *reinterpret_cast<const RowT *>(dbase + (int64_t)pout[start + pp] * cout);
                dr[ps][i] = (p < count && din) ? b : Row4<ST>::zero();
            }
        }
    };
    auto put = [&](uint16_t *T, int c4, int q, const RowT *v) {
        const int col = (((q >> 1) ^ ((c4 >> 4) & 7)) << 3) + ((q & 1) << 2);
        uint2 o[4];
        transpose4x4bf(v, o);
#pragma unroll
        for (int j = 0; j < 4; j++) *reinterpret_cast<uint2 *>(&T[(c4 + j) * WB_LD + col]) = o[j];
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < XP; ps++) put(Xs[buf], xc4, ps * (256 / XT) + tid / XT, xr[ps]);
#pragma unroll
        for (int ps = 0; ps < DP; ps++) put(Ds[buf], dc4, ps * (256 / DT) + tid / DT, dr[ps]);
    };
    const int nstage = (count + WB_S - 1) / WB_S;
    issue(0);
    commit(0);
    __syncthreads();
    for (int st = 0; st < nstage; st++) {
        const int buf = st & 1;
        if (st + 1 < nstage) issue((st + 1) * WB_S);            // in flight during the MFMAs
#pragma unroll
        for (int kk = 0; kk < WB_S / 16; kk++) {
            bf16x8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                const int row = wi * (TM / 2) + i * 32 + r;
                a[i] = *reinterpret_cast<const bf16x8 *>(&Xs[buf][row * WB_LD + (((kk * 2 + h) ^ ((row >> 4) & 7)) << 3)]);
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int row = wj * (TN / 2) + j * 32 + r;
                b[j] = *reinterpret_cast<const bf16x8 *>(&Ds[buf][row * WB_LD + (((kk * 2 + h) ^ ((row >> 4) & 7)) << 3)]);
            }
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (st + 1 < nstage) commit(buf ^ 1);                   // the other buffer was last read two stages ago
        __syncthreads();
    }
    float *dst = dW + (int64_t)k * cin * cout;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int ci = ci0 + wi * (TM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int co = co0 + wj * (TN / 2) + j * 32 + r;
                if (ci < cin && co < cout) unsafeAtomicAdd(&dst[(int64_t)ci * cout + co], acc[i][j][e]);
            }
}

// Weight gradient over rows stored as bf16 (precision 2).  The gather of this kernel is bound by the NUMBER of
// outstanding row requests, not by bytes (measured: halving the bytes per request changed nothing, removing the
// address scatter made it 3x faster), so every lane fetches 16 bytes = 8 channels here and a stage needs half
// the requests of the fp32-row kernel; the next TWO stages are kept in flight.  A thread transposes
// PQ pairs x 8 channels (PQ = 4 for a 128-channel operand tile, 2 for a 64-channel one) into the [channel][pair]
// LDS layout of k_spconv_pairs_wgrad_bf16; the MFMA phase is the same.
// NW = 8 (512 threads, 128 x 128 tile only): the same tile and LDS budget worked by twice the waves -- each owns a 64 x 32
// block (32 accumulator registers instead of 64) and transposes 2 pairs per stage instead of 4, so a wave needs <= 128
// registers and FOUR fit per SIMD (two workgroups per CU as before) where the 4-wave form holds two: the kernel is bound
// by dependent latencies (SQ_WAIT_ANY 46 % of its wave cycles), which more resident waves hide.
template <int T, int NW> struct WRows {
    static constexpr int TPR = T / 8;              // threads per gathered row
    static constexpr int PQ = 64 * TPR / (NW * 64);   // pairs per thread per stage (64 pairs / (threads / TPR))
    uint4 v[PQ];
};
template <int TM, int TN, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 2) void k_spconv_pairs_wgrad_rows16(const uint16_t *__restrict__ X,
                                                                      const uint16_t *__restrict__ dY,
                                                                      const int32_t *__restrict__ pin,
                                                                      const int32_t *__restrict__ pout,
                                                                      const int32_t *__restrict__ seg,
                                                                      float *__restrict__ dW, int32_t cin, int32_t cout,
                                                                      int32_t co_tiles, int32_t ldx, int32_t ldy) {
    // ldx / ldy: row pitch of X / dY in elements (cin / cout for plain rows; 3 cin / 3 cout when the operands are the hi or
    // lo part of split rows, cg3d_to_bf16_split)
    __shared__ __attribute__((aligned(16))) uint16_t Xs[2][TM * WB_LD];
    __shared__ __attribute__((aligned(16))) uint16_t Ds[2][TN * WB_LD];
    constexpr int WJ = NW / 2;                           // waves along the output-channel direction (2 along the input channels)
    constexpr int MI = TM / 64, NJ = TN / (32 * WJ);     // 32 x 32 tiles per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wi = wave / WJ, wj = wave % WJ;
    // Which channel of its 32-channel block lane r feeds to the MFMA.  A ds_read_b128 is served in the lane groups
    // {0-3,12-15,20-27} / {4-11,16-19,28-31} (and the same + 32); the XOR swizzle of the transposing stores depends on
    // bit 4 of the channel row, so with channel = r a group mixes rows of both halves and 7 of its 8 row pairs share a
    // bank (SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles).  pr sends the first group to channels 0-15 and the second to
    // 16-31: equal swizzle inside a group, rows 9 bank-quads apart -> conflict free.  The epilogue undoes it.
    auto perm32 = [](int l) -> int {
        return l < 4 ? l : l < 12 ? l + 12 : l < 16 ? l - 8 : l < 20 ? l + 8 : l < 28 ? l - 12 : l;
    };
    const int pr = perm32(r);
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;               // placeholder entries keep the XCD alignment of a segment table (uniform branch)
    const int ci0 = (blockIdx.y / co_tiles) * TM, co0 = (blockIdx.y % co_tiles) * TN;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

    constexpr int XTPR = WRows<TM, NW>::TPR, XPQ = WRows<TM, NW>::PQ, DTPR = WRows<TN, NW>::TPR, DPQ = WRows<TN, NW>::PQ;
    const int xc8 = (tid % XTPR) * 8, xp0 = (tid / XTPR) * XPQ;
    const int dc8 = (tid % DTPR) * 8, dp0 = (tid / DTPR) * DPQ;
    const bool xin = ci0 + xc8 < cin, din = co0 + dc8 < cout;       // cin, cout are multiples of 8 here
    const uint16_t *xbase = X + (xin ? ci0 + xc8 : 0);
    const uint16_t *dbase = dY + (din ? co0 + dc8 : 0);
    // The row gather of a pair needs the pair's row INDEX first: two dependent memory round trips per stage, and the
    // kernel ran at (index latency + row latency) / 2 per stage.  The indices are therefore requested two issues
    // ahead of the rows they address (8 registers per set), so a stage costs one round trip.
    struct WIdx { int32_t xi[XPQ], di[DPQ]; };
    auto load_idx = [&](WIdx &ix, int32_t p0) {
#pragma unroll
        for (int i = 0; i < XPQ; i++) {
            const int32_t p = p0 + xp0 + i;
            ix.xi[i] = pin[start + (p < count ? p : count - 1)];     // clamped, unconditional (rows masked below)
        }
#pragma unroll
        for (int i = 0; i < DPQ; i++) {
            const int32_t p = p0 + dp0 + i;
            ix.di[i] = pout[start + (p < count ? p : count - 1)];
        }
    };
    auto issue = [&](WRows<TM, NW> &xr, WRows<TN, NW> &dr, const WIdx &ix, int32_t p0) {
#pragma unroll
        for (int i = 0; i < XPQ; i++) {
            const int32_t p = p0 + xp0 + i;
            const uint4 a = *reinterpret_cast<const uint4 *>(xbase + (int64_t)ix.xi[i] * ldx);
            xr.v[i] = (p < count && xin) ? a : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int i = 0; i < DPQ; i++) {
            const int32_t p = p0 + dp0 + i;
            const uint4 b = *reinterpret_cast<const uint4 *>(dbase + (int64_t)ix.di[i] * ldy);
            dr.v[i] = (p < count && din) ? b : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    // PQ pairs x 8 channels -> eight [channel][PQ pairs] rows
    constexpr uint32_t LO = 0x05040100u, HI = 0x07060302u;
    auto put4 = [&](uint16_t *Tl, int c8, int p0, const uint4 *v) {
        const int col = (((p0 >> 3) ^ ((c8 >> 4) & 7)) << 3) + (p0 & 7);
        const uint32_t w0[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, w1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
        const uint32_t w2[4] = {v[2].x, v[2].y, v[2].z, v[2].w}, w3[4] = {v[3].x, v[3].y, v[3].z, v[3].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            *reinterpret_cast<uint2 *>(&Tl[(c8 + 2 * q) * WB_LD + col]) =
                make_uint2(__builtin_amdgcn_perm(w1[q], w0[q], LO), __builtin_amdgcn_perm(w3[q], w2[q], LO));
            *reinterpret_cast<uint2 *>(&Tl[(c8 + 2 * q + 1) * WB_LD + col]) =
                make_uint2(__builtin_amdgcn_perm(w1[q], w0[q], HI), __builtin_amdgcn_perm(w3[q], w2[q], HI));
        }
    };
    auto put2 = [&](uint16_t *Tl, int c8, int p0, const uint4 *v) {
        const int col = (((p0 >> 3) ^ ((c8 >> 4) & 7)) << 3) + (p0 & 7);
        const uint32_t w0[4] = {v[0].x, v[0].y, v[0].z, v[0].w}, w1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            *reinterpret_cast<uint32_t *>(&Tl[(c8 + 2 * q) * WB_LD + col]) = __builtin_amdgcn_perm(w1[q], w0[q], LO);
            *reinterpret_cast<uint32_t *>(&Tl[(c8 + 2 * q + 1) * WB_LD + col]) = __builtin_amdgcn_perm(w1[q], w0[q], HI);
        }
    };
    auto commit = [&](const WRows<TM, NW> &xr, const WRows<TN, NW> &dr, int buf) {
        if (XPQ == 4) put4(Xs[buf], xc8, xp0, xr.v); else put2(Xs[buf], xc8, xp0, xr.v);
        if (DPQ == 4) put4(Ds[buf], dc8, dp0, dr.v); else put2(Ds[buf], dc8, dp0, dr.v);
    };
    auto mfma_stage = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < WB_S / 16; kk++) {
            bf16x8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                const int row = wi * (TM / 2) + i * 32 + pr;
                a[i] = *reinterpret_cast<const bf16x8 *>(&Xs[buf][row * WB_LD + (((kk * 2 + h) ^ ((row >> 4) & 7)) << 3)]);
            }
#pragma unroll
            for (int j = 0; j < NJ; j++) {
                const int row = wj * (TN / WJ) + j * 32 + pr;
                b[j] = *reinterpret_cast<const bf16x8 *>(&Ds[buf][row * WB_LD + (((kk * 2 + h) ^ ((row >> 4) & 7)) << 3)]);
            }
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };
    // stage st lives in LDS buffer st&1; register set A holds stage st+1, set B stage st+2 (roles swap every stage)
    const int nstage = (count + WB_S - 1) / WB_S;
    WRows<TM, NW> xa, xb;
    WRows<TN, NW> da, db;
    WIdx i0, i1, iA, iB;
    load_idx(i0, 0);
    load_idx(i1, WB_S);                      // past the end: clamped rows, masked to zero, never used
    load_idx(iA, 2 * WB_S);
    load_idx(iB, 3 * WB_S);
    issue(xa, da, i0, 0);
    issue(xb, db, i1, WB_S);
    load_idx(i0, 4 * WB_S);                  // i0 / iB from here on: the index sets of the odd / even half below
    commit(xa, da, 0);
    issue(xa, da, iA, 2 * WB_S);
    __syncthreads();
    // iB holds the indices of stage 3 (next xb issue), i0 those of stage 4 (next xa issue)
    for (int st = 0; st < nstage; st += 2) {
        // even stage: xb/db hold stage st+1, xa/da stage st+2
        mfma_stage(0);
        commit(xb, db, 1);
        issue(xb, db, iB, (st + 3) * WB_S);
        load_idx(iB, (st + 5) * WB_S);
        __syncthreads();
        if (st + 1 < nstage) {
            mfma_stage(1);
            commit(xa, da, 0);
            issue(xa, da, i0, (st + 4) * WB_S);
            load_idx(i0, (st + 6) * WB_S);
            __syncthreads();
        }
    }
    float *dst = dW + (int64_t)k * cin * cout;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int ci = ci0 + wi * (TM / 2) + i * 32 + perm32((e & 3) + 8 * (e >> 2) + 4 * h);
                const int co = co0 + wj * (TN / WJ) + j * 32 + pr;
#if defined(W_DBG) && (W_DBG & 1)        // dev knock-outs (tools/build_tile_dbg.sh, SRC=spconv.hip DEF=W_DBG): 1 no atomics, 2 plain stores
                if (ci < cin && co < cout && ldx < 0) dst[(int64_t)ci * cout + co] = acc[i][j][e];
#elif defined(W_DBG) && (W_DBG & 2)
                if (ci < cin && co < cout) dst[(int64_t)ci * cout + co] = acc[i][j][e];
#else
                if (ci < cin && co < cout) unsafeAtomicAdd(&dst[(int64_t)ci * cout + co], acc[i][j][e]);
#endif
            }
}

// Weight gradient with ONE narrow side (fp32): the stem (3 -> 64), the semantic / offset / per-class score, box and centerness
// layers (64 -> 18 / 3 / 6 / 1).  A 64 x 64 MFMA tile is 5-28 % full there and the generic kernel ran 80-200 us per launch for
// 11-110 MB of rows.  Here the WIDE operand's channels sit on the lanes (coalesced row reads), the narrow operand's <= 32
// values ride on the first lanes of one coalesced load and reach every lane by v_readlane (as wave-uniform scalar loads
// 4 x 18 of them spilled the scalar file: 182 us at cout = 18), a wave streams the pairs of its quarter of the segment PF at a time and
// keeps dW[wide channel = lane][narrow channel] in registers; the waves meet in LDS, one atomic per element at the end.  Every
// workgroup ends with wide x narrow atomics onto the SAME addresses: with 18 narrow channels 609 four-wave workgroups spent
// most of 160 us in that queue -- 16 waves per workgroup (NW) and a quarter of the workgroups keep the parallelism.
//   NARROW_OUT: narrow = dY (cout <= 32), wide = X;  else narrow = X (cin <= 32), wide = dY.
template <int S, bool NARROW_OUT, int NW>
__global__ __launch_bounds__(NW * 64) void k_spconv_pairs_wgrad_skinny(const float *__restrict__ X, const float *__restrict__ dY,
                                                                   const int32_t *__restrict__ pin,
                                                                   const int32_t *__restrict__ pout,
                                                                   const int32_t *__restrict__ seg, float *__restrict__ dW,
                                                                   int32_t cin, int32_t cout) {
    constexpr int PF = 8;
    __shared__ float red[NW - 1][64][S + 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int32_t k = seg[blockIdx.x * 3], start = seg[blockIdx.x * 3 + 1], count = seg[blockIdx.x * 3 + 2];
    if (count <= 0) return;
    const int wide = NARROW_OUT ? cin : cout, narrow = NARROW_OUT ? cout : cin;
    const int w0 = blockIdx.y * 64;                              // this workgroup's 64 wide channels
    const bool wok = w0 + lane < wide;
    const float *Wd = NARROW_OUT ? X : dY, *Nr = NARROW_OUT ? dY : X;
    const int32_t *wi = NARROW_OUT ? pin : pout, *ni = NARROW_OUT ? pout : pin;
    float acc[S];
#pragma unroll
    for (int j = 0; j < S; j++) acc[j] = 0.f;
    const int nl = lane < narrow ? lane : 0;                     // the narrow row rides on the first `narrow` lanes
    for (int32_t p0 = wave * PF; p0 < count; p0 += NW * PF) {
        float wv[PF], nv[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int32_t p = p0 + u < count ? p0 + u : count - 1;          // clamped, unconditional loads; masked below
            const int64_t wr = wi[start + p], nr = ni[start + p];
            wv[u] = wok ? Wd[wr * wide + w0 + lane] : 0.f;
            nv[u] = Nr[nr * narrow + nl];
        }
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const float x = p0 + u < count ? wv[u] : 0.f;
#pragma unroll
            for (int j = 0; j < S; j++)          // lane j's value to every lane (v_readlane: no memory, no long-lived scalars)
                acc[j] += x * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(nv[u]), j));
        }
    }
    if (wave) {
#pragma unroll
        for (int j = 0; j < S; j++) red[wave - 1][lane][j] = acc[j];
    }
    __syncthreads();
    if (wave || !wok) return;
    float *dst = dW + (int64_t)k * cin * cout;
#pragma unroll
    for (int j = 0; j < S; j++) {
        if (j >= narrow) break;
        float v = acc[j];
#pragma unroll
        for (int w = 0; w < NW - 1; w++) v += red[w][lane][j];
        // dW is [cin][cout]: wide = cin -> element (w0 + lane, j); wide = cout -> element (j, w0 + lane)
        unsafeAtomicAdd(NARROW_OUT ? &dst[(int64_t)(w0 + lane) * cout + j] : &dst[(int64_t)j * cout + w0 + lane], v);
    }
}

extern "C" int cg3d_spconv_pairs_wgrad(const float *X, const float *dY, const int32_t *pair_in,
                                       const int32_t *pair_out, const int32_t *seg, int64_t nseg, float *dW,
                                       int32_t K, int32_t cin, int32_t cout, int32_t precision, cg3d_stream_t stream) {
    if (nseg < 0 || K < 1 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    // CG3D_WGRAD_ACCUMULATE: dW += ... (the caller initialised dW -- e.g. one zero-fill for every weight gradient of a pass)
    const bool accumulate = (precision & CG3D_WGRAD_ACCUMULATE) != 0;
    precision &= ~CG3D_WGRAD_ACCUMULATE;
    if (precision < 0 || precision > 3) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (!accumulate && hipMemsetAsync(dW, 0, (int64_t)K * cin * cout * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (nseg == 0) return CG3D_OK;
    if (precision >= 1) {   // bf16 operands: fp32 rows rounded on the fly (1) or rows stored as bf16 (2); fp32 accumulate
        if (cin % 4 != 0 || cout % 4 != 0 || (((uintptr_t)X | (uintptr_t)dY) & 15)) return CG3D_ERR_ARG;
        const bool m128 = cin > 64, n128 = cout > 64;
        const int32_t ct = cg3d_divup(cin, m128 ? 128 : 64), ot = cg3d_divup(cout, n128 ? 128 : 64);
        if ((int64_t)ct * ot > 65535) return CG3D_ERR_ARG;
#define LAUNCH_WB(TM, TN, ST)                                                                                      \
    hipLaunchKernelGGL((k_spconv_pairs_wgrad_bf16<TM, TN>), dim3((unsigned)nseg, (unsigned)(ct * ot)), dim3(256), 0, s, X, dY, \
                       pair_in, pair_out, seg, dW, cin, cout, ot)
#define LAUNCH_WB_ALL(ST)                                                                                          \
    do {                                                                                                           \
        if (m128 && n128) LAUNCH_WB(128, 128, ST);                                                                 \
        else if (m128) LAUNCH_WB(128, 64, ST);                                                                     \
        else if (n128) LAUNCH_WB(64, 128, ST);                                                                     \
        else LAUNCH_WB(64, 64, ST);                                                                                \
    } while (0)
        if (precision >= 2) {   // rows stored as bf16: 16-byte gathers need 8-channel multiples
            if (cin % 8 != 0 || cout % 8 != 0) return CG3D_ERR_ARG;
            // precision 3: split rows [hi | lo | hi] of 3 cin / 3 cout channels (cg3d_to_bf16_split):
            //   dW = Xhi^T dYhi + Xlo^T dYhi + Xhi^T dYlo   -- three accumulating launches on the parts of the same rows
            const int npass = precision == 3 ? 3 : 1;
            const int32_t ldx = precision == 3 ? 3 * cin : cin, ldy = precision == 3 ? 3 * cout : cout;
            static const bool w8 = !(getenv("CG3D_WGRAD_W8") && atoi(getenv("CG3D_WGRAD_W8")) == 0);
            for (int pass = 0; pass < npass; pass++) {
                const uint16_t *Xp = reinterpret_cast<const uint16_t *>(X) + (pass == 1 ? cin : 0);
                const uint16_t *Dp = reinterpret_cast<const uint16_t *>(dY) + (pass == 2 ? cout : 0);
#define LAUNCH_WR(TM, TN, NW)                                                                                      \
    hipLaunchKernelGGL((k_spconv_pairs_wgrad_rows16<TM, TN, NW>), dim3((unsigned)nseg, (unsigned)(ct * ot)), dim3(NW * 64), 0, s, \
                       Xp, Dp, pair_in, pair_out, seg, dW, cin, cout, ot, ldx, ldy)
                if (m128 && n128 && w8) LAUNCH_WR(128, 128, 8);
                else if (m128 && n128) LAUNCH_WR(128, 128, 4);
                else if (m128) LAUNCH_WR(128, 64, 4);
                else if (n128) LAUNCH_WR(64, 128, 4);
                else LAUNCH_WR(64, 64, 4);
#undef LAUNCH_WR
                CG3D_CHECK_LAUNCH();
            }
            return CG3D_OK;
        }
        LAUNCH_WB_ALL(float);
#undef LAUNCH_WB_ALL
#undef LAUNCH_WB
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    static const bool skinny = !(getenv("CG3D_WGRAD_SKINNY") && atoi(getenv("CG3D_WGRAD_SKINNY")) == 0);
    if (skinny && ((cout <= 32 && cin >= 32) || (cin <= 32 && cout >= 32))) {
        const bool narrow_out = cout <= 32 && cin >= 32;
        const int narrow = narrow_out ? cout : cin, wide = narrow_out ? cin : cout;
        const dim3 grid((unsigned)nseg, (unsigned)cg3d_divup(wide, 64));
#define LAUNCH_SK(S, NW)                                                                                                        \
    do {                                                                                                                        \
        if (narrow_out) hipLaunchKernelGGL((k_spconv_pairs_wgrad_skinny<S, true, NW>), grid, dim3(NW * 64), 0, s, X, dY, pair_in, pair_out, seg, dW, cin, cout); \
        else hipLaunchKernelGGL((k_spconv_pairs_wgrad_skinny<S, false, NW>), grid, dim3(NW * 64), 0, s, X, dY, pair_in, pair_out, seg, dW, cin, cout);          \
    } while (0)
        if (narrow <= 4) LAUNCH_SK(4, 4);
        else if (narrow <= 8) LAUNCH_SK(8, 4);
        else if (narrow <= 20) LAUNCH_SK(20, 16);
        else LAUNCH_SK(32, 16);
#undef LAUNCH_SK
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    if (cin >= 128 && cout >= 128 && cin % 4 == 0 && cout % 4 == 0 && !(((uintptr_t)X | (uintptr_t)dY) & 15)) {
        const int32_t ct = (cin + 127) / 128, ot = (cout + 127) / 128;
        hipLaunchKernelGGL(k_spconv_pairs_wgrad_t128, dim3((unsigned)nseg, (unsigned)(ct * ot)), dim3(256), 0, s, X, dY,
                           pair_in, pair_out, seg, dW, cin, cout, ot);
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    const int32_t ci_tiles = (cin + 63) / 64, co_tiles = (cout + 63) / 64;
    if ((int64_t)ci_tiles * co_tiles > 65535) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_spconv_pairs_wgrad, dim3((unsigned)nseg, (unsigned)(ci_tiles * co_tiles)), dim3(256), 0, s, X,
                       dY, pair_in, pair_out, seg, dW, cin, cout, co_tiles);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ---------------------------------------------------------------- dense-map (output-stationary) entry point
extern "C" int cg3d_spconv_fwd_tiled(const float *X, const float *W, const int32_t *nbr, const int32_t *tiles,
                                     int64_t ntile, const float *bias, float *Y, int64_t n_in, int64_t n_out, int32_t K,
                                     int32_t cin, int32_t cout, int32_t precision, cg3d_stream_t stream) {
    if (n_out < 0 || n_in < 0 || K < 1 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    if (precision < 0 || precision > 2) return CG3D_ERR_ARG;
    if (n_out == 0 || (tiles && ntile == 0)) return CG3D_OK;
    if (tiles && (precision == 0 || ntile < 0)) return CG3D_ERR_ARG;     // row groups: bf16 forms only
    hipStream_t s = cg3d_hs(stream);
    const unsigned gx = tiles ? (unsigned)ntile : (unsigned)cg3d_divup(n_out, 128);
    if (precision >= 1) {   // W is the prepared bf16 [K][cout][cin] buffer (cg3d_spconv_prep_weights_bf16)
        // 32-bit element offsets inside the kernel: rows * cin and the weight tensor must stay below 2^31 elements
        if (cin % 8 != 0 || ((uintptr_t)X & 15) || ((uintptr_t)W & 15)) return CG3D_ERR_ARG;
        if (n_in * (int64_t)cin >= (1ll << 31) || (int64_t)cin * cout >= (1ll << 31)) return CG3D_ERR_RANGE;
        const uint16_t *Wb = reinterpret_cast<const uint16_t *>(W);
        // small layers: fewer workgroups than the chip has slots -> split the offsets over gridDim.z
        const int64_t nwg = (int64_t)gx * cg3d_divup(cout, cout > 64 ? 128 : (cout > 32 ? 64 : 32));
        const int zmax = 4;
        unsigned gz = 1;
        while (!tiles && (int)gz < zmax && nwg * (gz + 1) <= 768 && (int)(gz + 1) * 4 <= K) gz++;
        if (gz > 1 && hipMemsetAsync(Y, 0, (size_t)n_out * cout * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
#define LAUNCH_WS(NT, XT)                                                                                       \
    hipLaunchKernelGGL((k_spconv_implicit_bf16_ws<NT, XT>), dim3(gx, (unsigned)cg3d_divup(cout, NT * 32), gz), dim3(512), 0, s, \
                       reinterpret_cast<const XT *>(X), Wb, nbr, bias, Y, n_out, K, cin, cout, tiles)
#define LAUNCH_AD(NT)                                                                                            \
    hipLaunchKernelGGL((k_spconv_implicit_bf16_ad<NT>), dim3(gx, (unsigned)cg3d_divup(cout, NT * 32), gz), dim3(256), 0, s, \
                       reinterpret_cast<const uint16_t *>(X), Wb, nbr, bias, Y, n_out, K, cin, cout, tiles)
        if (precision == 2) {   // X is bf16 [n_in][cin] (cg3d_to_bf16): fragments straight from the row matrix
            if (cout > 64) LAUNCH_AD(4); else if (cout > 32) LAUNCH_AD(2); else LAUNCH_AD(1);
        } else {                // fp32 rows, rounded to bf16 on the way into LDS by the gather waves
            if (cout > 64) LAUNCH_WS(4, float); else if (cout > 32) LAUNCH_WS(2, float); else LAUNCH_WS(1, float);
        }
#undef LAUNCH_AD
#undef LAUNCH_WS
        CG3D_CHECK_LAUNCH();
        return CG3D_OK;
    }
    const bool vec4 = (cin % 4 == 0) && (((uintptr_t)X & 15) == 0);
#define LAUNCH(NT, KH, V)                                                                                        \
    hipLaunchKernelGGL((k_spconv_fwd<NT, KH, V>), dim3(gx, (unsigned)cg3d_divup(cout, NT * 32)), dim3(256), 0, s, X, \
                       W, nbr, bias, Y, n_out, K, cin, cout)
    if (cin <= 4) {
        if (cout > 64) LAUNCH(4, 2, false); else LAUNCH(2, 2, false);
    } else if (vec4) {
        if (cout > 64) LAUNCH(4, 32, true); else if (cout > 32) LAUNCH(2, 32, true); else LAUNCH(1, 32, true);
    } else {
        if (cout > 64) LAUNCH(4, 32, false); else if (cout > 32) LAUNCH(2, 32, false); else LAUNCH(1, 32, false);
    }
#undef LAUNCH
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

extern "C" int cg3d_spconv_fwd(const float *X, const float *W, const int32_t *nbr, const float *bias, float *Y,
                               int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                               int32_t precision, cg3d_stream_t stream) {
    return cg3d_spconv_fwd_tiled(X, W, nbr, nullptr, 0, bias, Y, n_in, n_out, K, cin, cout, precision, stream);
}
