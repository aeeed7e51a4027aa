// cg3d_common.h -- shared device/host helpers of the gfx950 hot-path library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/cagroup3d_hip.h"

#define CG3D_EMPTY_KEY (~0ULL)

#define CG3D_CHECK_LAUNCH()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return CG3D_ERR_LAUNCH;        \
    } while (0)

static inline hipStream_t cg3d_hs(cg3d_stream_t s) { return (hipStream_t)s; }
static inline int64_t cg3d_divup(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Packed coordinate key: batch 19 bits | x 15 | y 15 | z 15 (each biased by 2^14).
__host__ __device__ static inline bool cg3d_pack(int32_t b, int32_t x, int32_t y, int32_t z, uint64_t *key) {
    if ((uint32_t)b >= (uint32_t)CG3D_BATCH_LIMIT) return false;
    uint32_t ux = (uint32_t)(x + CG3D_COORD_LIMIT), uy = (uint32_t)(y + CG3D_COORD_LIMIT),
             uz = (uint32_t)(z + CG3D_COORD_LIMIT);
    if ((ux | uy | uz) >= (uint32_t)(2 * CG3D_COORD_LIMIT)) return false;
    *key = ((uint64_t)b << 45) | ((uint64_t)ux << 30) | ((uint64_t)uy << 15) | (uint64_t)uz;
    return true;
}
__host__ __device__ static inline uint64_t cg3d_hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
__host__ __device__ static inline int32_t cg3d_floordiv(int32_t a, int32_t s) {
    int32_t q = a / s;
    if ((a % s != 0) && ((a < 0) != (s < 0))) q--;
    return q;
}
__device__ static inline int32_t cg3d_lookup(const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals,
                                             uint64_t capm1, uint64_t key) {
    uint64_t slot = cg3d_hash(key) & capm1;
    for (;;) {
        uint64_t k = keys[slot];
        if (k == key) return vals[slot];
        if (k == CG3D_EMPTY_KEY) return -1;
        slot = (slot + 1) & capm1;
    }
}

// Element (n_idx = MFMA column, k_idx = contraction index) of a [N][kdim] bf16 operand stored in MFMA B-fragment order
// (cg3d_spconv_prep_weights_frag): [n/32][k/16][lane = (k/8 & 1)*32 + n%32][k%8] -- a wave's fragment is one contiguous KB.
__host__ __device__ static inline int64_t cg3d_frag_index(int n_idx, int k_idx, int kdim) {
    return ((((int64_t)(n_idx >> 5) * (kdim >> 4) + (k_idx >> 4)) * 64) + ((k_idx >> 3) & 1) * 32 + (n_idx & 31)) * 8 + (k_idx & 7);
}
