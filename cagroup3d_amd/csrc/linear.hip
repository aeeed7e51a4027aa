// linear.hip -- Y = X @ W (+ bias) on bf16 rows with fp32 accumulation: the 1x1x1 convolutions of the path (gfx950).
//
// Replaces the library GEMMs (hipBLASLt through torch) behind every kernel-size-1 MinkowskiConvolution --
// pcdet/models/backbones_3d/biresnet.py:270-280,308-315 (Bottleneck conv1 / conv3, compression3/4, DAPPM scale and
// compression convolutions, `out`), dense_heads/cagroup_head.py:163-188 (offset block, the shared head) -- forward and data
// gradient (dX = dY @ W^T: the same call on the other fragment-ordered copy of the weights), and, with `ksplit`, the per-RoI
// 7^3 -> centre contraction of roi_heads/cagroup_roi_head.py:74-91 ([R, 343 * 128] x [343 * 128, 128]).
//
// These products are HBM-bound (a 155 773 x 64 x 64 layer: 1.3 GFLOP against 60 MB of rows), so the kernel is a streaming
// one: a workgroup (4 waves) owns 128 rows x 128 (64) output channels; per 64-channel chunk of the contraction the rows
// (16 KB) AND the weight chunk (16 / 8 KB, MFMA fragment order: cg3d_spconv_prep_weights_frag with one slot) reach LDS by
// LDS-DMA, wave w multiplies rows [32 w, 32 w + 32) by all output blocks (v_mfma_f32_32x32x16_bf16, 4-16 per chunk), the
// tile leaves row-major through LDS with 16-byte stores, and the same epilogue as the tile convolution adds the per-channel
// sum / sum of squares into the layer's BatchNorm statistics table.  32 KB of LDS and <= 100 VGPRs: four workgroups per CU
// hide each other's load latency (the chunk loop is single-buffered); the hardware dispatches the ~1 200 workgroups of a
// large layer as slots free up.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include "cg3d_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define LN_TM 128
#define LN_LDS (32 * 1024)          // A chunk 16 KB | B chunk <= 16 KB ... reused as the 4 x 8 KB output regions

template <int NCO>                  // output channels of a workgroup = NCO * 64
__global__ __launch_bounds__(256, 2) void k_linear_tile(const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wf,
                                                        const float *__restrict__ bias, float *__restrict__ Y, int64_t n,
                                                        int32_t cin, int32_t cout, int32_t ny, int32_t gz,
                                                        float *__restrict__ stats, float *__restrict__ part, int32_t out16) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int NB = NCO * 2;                          // 32-channel output blocks of the workgroup
    constexpr int NC = NCO * 64;
    uint8_t *const As = smem;                            // [128 rows][128 B], granule g of row r at g ^ ((r >> 1) & 7)
    uint8_t *const Bs = smem + 16384;                    // [NB][4 ks][64 lanes][16 B]
    __shared__ float sacc[2 * 128];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = blockIdx.x;
    const int64_t tile = u / (ny * gz);
    const int yb = (u / gz) % ny, zi = u % gz;
    const int64_t row0 = tile * LN_TM;
    const int rows = (int)(n - row0 < LN_TM ? n - row0 : LN_TM);
    const int nchunk = cin >> 6, ks_total = cin >> 4;
    const int c_lo = (int)((int64_t)nchunk * zi / gz), c_hi = (int)((int64_t)nchunk * (zi + 1) / gz);
    const int nt0 = yb * NB;
    if (stats && tid < 2 * NC) sacc[tid] = 0.f;
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[b][e] = 0.f;
    const int r = lane & 31, kg = lane >> 5;
    const int arow = wave * 32 + r;                      // this lane's row of the tile
    const uint32_t a0 = (uint32_t)arow * 128u + ((((uint32_t)kg) ^ (((uint32_t)arow >> 1) & 7u)) << 4);   // ks 0; ks: ^ (ks << 5)
    for (int c = c_lo; c < c_hi; c++) {
        if (c > c_lo) __syncthreads();                   // every wave has read the previous chunk
        // rows: 128 x 8 granules = 16 requests of 64 lanes; request q of wave w covers rows 8 (4 w + q) .. + 8
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = (wave * 4 + q) * 64 + lane;    // PHYSICAL granule index of the tile
            const int rr = i >> 3, gr = (i & 7) ^ ((rr >> 1) & 7);
            const int64_t grow = row0 + (rr < rows ? rr : rows - 1);            // rows past the end: a valid row, never stored
            const uint16_t *src = X + (grow * cin + c * 64 + gr * 8);
            uint8_t *dst = As + (size_t)((wave * 4 + q) * 64) * 16;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        // weights: fragments (nb, ks) of this chunk, 1 KB each, NB * 4 of them
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const int f = wave * NB + q, nb = f >> 2, ks = f & 3;               // NB * 4 fragments over 4 waves
            const uint16_t *src = Wf + (((int64_t)(nt0 + nb) * ks_total + c * 4 + ks) * 512 + lane * 8);
            uint8_t *dst = Bs + (size_t)f * 1024;
            __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            const bf16x8 a = *reinterpret_cast<const bf16x8 *>(As + (a0 ^ ((uint32_t)ks << 5)));
#pragma unroll
            for (int nb = 0; nb < NB; nb++) {
                const bf16x8 b = *reinterpret_cast<const bf16x8 *>(Bs + (size_t)(nb * 4 + ks) * 1024 + lane * 16);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
            }
        }
    }
    // ---- the wave's 32 rows, 64 channels at a time, row-major in its own 8 KB region, then 16-byte stores
    __syncthreads();
    // BatchNorm statistics EARLY (as in spconv_tile2.hip: the workgroup's global atomics then travel while it lays out and stores
    // its rows, instead of being the tail every workgroup ends with): from the accumulators, rows past the end masked (they
    // hold a duplicate of the last row).  With a bias the store loop's path below stays.
    const bool early_stats = stats != nullptr && !(bias && zi == 0 && !part);
    if (early_stats) {
#pragma unroll
        for (int nb = 0; nb < NB; nb++) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const float v = (wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg < rows) ? acc[nb][e] : 0.f;
                s0 += v; s1 += v * v;
            }
            s0 += __shfl_xor(s0, 32);
            s1 += __shfl_xor(s1, 32);
            if (kg == 0) { unsafeAtomicAdd(sacc + nb * 32 + r, s0); unsafeAtomicAdd(sacc + NC + nb * 32 + r, s1); }
        }
        __syncthreads();
        if (tid < 2 * NC)
            unsafeAtomicAdd(&stats[((blockIdx.x % CG3D_BN_SLOTS) * 2 + tid / NC) * (int64_t)cout + yb * NC + tid % NC], sacc[tid]);
    }
    float *tb = reinterpret_cast<float *>(smem) + (size_t)wave * 2048;          // [32][64]
    const int c4 = (lane & 15) * 4, rq = lane >> 4;      // 16 threads per row, 4 rows per iteration of the wave
#pragma unroll
    for (int h = 0; h < NCO; h++) {
#pragma unroll
        for (int nb = 0; nb < 2; nb++)
#pragma unroll
            for (int e = 0; e < 16; e++) tb[((e & 3) + 8 * (e >> 2) + 4 * kg) * 64 + nb * 32 + r] = acc[2 * h + nb][e];
        // (the wave reads back what it wrote itself: LDS operations of one wave execute in order)
        const int col0 = yb * NC + h * 64 + c4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && zi == 0 && !part) bv = *reinterpret_cast<const float4 *>(bias + col0);
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int rr = i * 4 + rq;
            float4 v = *reinterpret_cast<const float4 *>(tb + rr * 64 + c4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (wave * 32 + rr < rows) {
                float *dst = (part ? part + (int64_t)zi * n * cout : Y) + (row0 + wave * 32 + rr) * (int64_t)cout + col0;
                if (out16) {                              // Y holds bf16 rows (CG3D_LINEAR_OUT_BF16; gz == 1, no partials)
                    typedef float ln_f32x2 __attribute__((ext_vector_type(2)));
                    typedef __bf16 ln_bf16x2 __attribute__((ext_vector_type(2)));
                    const ln_f32x2 p0 = {v.x, v.y}, p1 = {v.z, v.w};
                    *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(Y) + (row0 + wave * 32 + rr) * (int64_t)cout + col0) =
                        make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(p0, ln_bf16x2)),
                                   __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, ln_bf16x2)));
                } else if (gz == 1 || part) *reinterpret_cast<float4 *>(dst) = v;
                else { unsafeAtomicAdd(dst, v.x); unsafeAtomicAdd(dst + 1, v.y); unsafeAtomicAdd(dst + 2, v.z); unsafeAtomicAdd(dst + 3, v.w); }
                t0.x += v.x; t0.y += v.y; t0.z += v.z; t0.w += v.w;
                t1.x += v.x * v.x; t1.y += v.y * v.y; t1.z += v.z * v.z; t1.w += v.w * v.w;
            }
        }
        if (stats && !early_stats) {
            float *s0 = sacc + h * 64 + c4, *s1 = sacc + NC + h * 64 + c4;
            unsafeAtomicAdd(s0, t0.x); unsafeAtomicAdd(s0 + 1, t0.y); unsafeAtomicAdd(s0 + 2, t0.z); unsafeAtomicAdd(s0 + 3, t0.w);
            unsafeAtomicAdd(s1, t1.x); unsafeAtomicAdd(s1 + 1, t1.y); unsafeAtomicAdd(s1 + 2, t1.z); unsafeAtomicAdd(s1 + 3, t1.w);
        }
    }
    if (stats && !early_stats) {
        __syncthreads();
        if (tid < 2 * NC)
            unsafeAtomicAdd(&stats[((blockIdx.x % CG3D_BN_SLOTS) * 2 + tid / NC) * (int64_t)cout + yb * NC + tid % NC], sacc[tid]);
    }
}

// Y = bias + sum_z part[z]: the second half of a split contraction whose partial products were stored, not added atomically.
// A workgroup owns 64 float4 columns; its four 64-thread groups each sum a quarter of the ranges, LDS joins them.
__global__ __launch_bounds__(256) void k_linear_reduce(const float4 *__restrict__ part, const float *__restrict__ bias,
                                                       float4 *__restrict__ Y, int64_t total4, int32_t cout4, int32_t gz) {
    __shared__ float4 sh[3][64];
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + col;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4)
        for (int z = q; z < gz; z += 4) {
            const float4 v = part[(int64_t)z * total4 + i];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
    if (q) sh[q - 1][col] = a;
    __syncthreads();
    if (q || i >= total4) return;
    for (int k = 0; k < 3; k++) { const float4 v = sh[k][col]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    if (bias) { const float4 v = reinterpret_cast<const float4 *>(bias)[i % cout4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    Y[i] = a;
}

extern "C" int cg3d_linear_fwd(const uint16_t *X, const uint16_t *Wf, const float *bias, float *Y, int64_t n, int32_t cin,
                               int32_t cout, int32_t ksplit, float *stats, float *partials, cg3d_stream_t stream) {
    const int32_t out16 = (ksplit & CG3D_LINEAR_OUT_BF16) ? 1 : 0;          // Y: uint16 [n, cout] bf16 rows (ksplit == 1 only)
    ksplit &= ~CG3D_LINEAR_OUT_BF16;
    if (out16 && ksplit != 1) return CG3D_ERR_ARG;
    if (n < 0 || cin < 64 || (cin & 63) || cout < 64 || (cout & 63) || ksplit < 1 || ksplit > 256 || ksplit > (cin >> 6)) return CG3D_ERR_ARG;
    if (((uintptr_t)X & 15) || ((uintptr_t)Wf & 15) || ((uintptr_t)Y & 15) || ((uintptr_t)partials & 15) || (stats && ksplit != 1)) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipStream_t s = cg3d_hs(stream);
    if (ksplit == 1) partials = nullptr;
    if (ksplit > 1 && !partials && hipMemsetAsync(Y, 0, (size_t)n * cout * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    const int nco = (cout % 128 == 0) ? 2 : 1;
    const int32_t ny = cout / (nco * 64);
    const int64_t nunit = cg3d_divup(n, LN_TM) * ny * ksplit;
    if (nunit > 0x7fffffffll) return CG3D_ERR_ARG;
#define LN_LAUNCH(NCO)                                                                                                            \
    do {                                                                                                                          \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_linear_tile<NCO>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    LN_LDS) != hipSuccess)                                                                        \
                return CG3D_ERR_LAUNCH;                                                                                           \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((k_linear_tile<NCO>), dim3((unsigned)nunit), dim3(256), LN_LDS, s, X, Wf, bias, Y, n, cin, cout, ny,   \
                           ksplit, stats, partials, out16);                                                                       \
    } while (0)
    if (nco == 2) LN_LAUNCH(2); else LN_LAUNCH(1);
#undef LN_LAUNCH
    CG3D_CHECK_LAUNCH();
    if (partials) {
        const int64_t total4 = n * cout / 4;
        hipLaunchKernelGGL(k_linear_reduce, dim3((unsigned)cg3d_divup(total4, 64)), dim3(256), 0, s,
                           reinterpret_cast<const float4 *>(partials), bias, reinterpret_cast<float4 *>(Y), total4, cout / 4, ksplit);
        CG3D_CHECK_LAUNCH();
    }
    return CG3D_OK;
}
