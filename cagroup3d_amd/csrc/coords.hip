// coords.hip -- hash-built voxel grid for gfx950: coordinate-map build (insert, first-occurrence
// representative, ordered compaction), kernel maps, interpolation maps, pooling maps.
//
// Replaces what the reference gets from MinkowskiEngine's CoordinateManager (un-vendored; call
// sites: pcdet/models/detectors/cagroup3d.py:18-25, backbones_3d/biresnet.py (every strided
// conv), dense_heads/cagroup_head.py:254-276, roi_heads/cagroup_roi_head.py:62-69).
// All integer work: HBM/L2-latency bound random probes; kernels are one thread per probe with
// coalesced coordinate reads and coalesced map writes (k-major maps: nbr[k][row]).
#include <rocprim/device/device_radix_sort.hpp>
#include "cg3d_common.h"

extern "C" int cg3d_is_device_library(void) { return 1; }
extern "C" int cg3d_abi_version(void) { return 2; }
extern "C" int cg3d_h2d_async(void *dst, const void *src, int64_t nbytes, cg3d_stream_t stream) {
    if (nbytes < 0 || (nbytes > 0 && (!dst || !src))) return CG3D_ERR_ARG;
    if (nbytes == 0) return CG3D_OK;
    return hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyHostToDevice, cg3d_hs(stream)) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}

extern "C" int64_t cg3d_hash_capacity(int64_t n) {
    int64_t cap = 64;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}
extern "C" int64_t cg3d_coord_map_ws_bytes(int64_t n) { return (2 * n + n / 1024 + 64) * (int64_t)sizeof(int32_t); }

// ------------------------------------------------------------------ build
__device__ static inline bool quantised_key(const int32_t *__restrict__ coords, int64_t i, int32_t qs, int4 *c,
                                            uint64_t *key) {
    int4 v = reinterpret_cast<const int4 *>(coords)[i];
    if (qs > 1) {
        v.y = cg3d_floordiv(v.y, qs) * qs;
        v.z = cg3d_floordiv(v.z, qs) * qs;
        v.w = cg3d_floordiv(v.w, qs) * qs;
    }
    *c = v;
    return cg3d_pack(v.x, v.y, v.z, v.w, key);
}

// A run of consecutive rows with the same key (the 343 grid points of a degenerate RoI all name one voxel; neighbouring grid
// points of a small RoI share theirs: cagroup_roi_head.py:199-224) is inserted by its FIRST lane only -- the lowest row of the
// run, which is what atomicMin would leave anyway -- and the slot is handed to the rest of the run by a shuffle: 22 k
// same-address atomics per step serialised at the L2 atomic unit (0.47 ms in the RoI stage's map build).
__global__ __launch_bounds__(256) void k_insert(const int32_t *__restrict__ coords, int64_t n, int32_t qs, unsigned long long *keys,
                                                int32_t *vals, uint64_t capm1, int32_t *slot_of, int32_t *status) {
    // wave64 only (gfx950): the lane index, the 64-bit ballot masks and the default-width shuffles below assume it -- on a
    // 32-lane target lane 32 would read a slot out of another wave's ballot bits and the map would be silently corrupt
    // (cg3d_coord_map_build refuses a device whose wavefronts are not 64 lanes wide)
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int4 c;
    uint64_t key = ~0ull;                               // (CG3D_EMPTY_KEY is no valid key: bit 63 of a packed key is 0)
    bool valid = false;
    if (i < n) {
        valid = quantised_key(coords, i, qs, &c, &key);
        if (!valid) { *status = CG3D_ERR_RANGE; key = ~0ull; }
    }
    const uint64_t prev = __shfl_up((unsigned long long)key, 1);
    const bool head = lane == 0 || key != prev || !valid;
    int32_t slot_i = -1;
    if (valid && head) {
        uint64_t slot = cg3d_hash(key) & capm1;
        for (;;) {
            unsigned long long was = atomicCAS(&keys[slot], CG3D_EMPTY_KEY, (unsigned long long)key);
            if (was == CG3D_EMPTY_KEY || was == key) {
                atomicMin(&vals[slot], (int32_t)i);  // representative = first occurrence
                slot_i = (int32_t)slot;
                break;
            }
            slot = (slot + 1) & capm1;
        }
    }
    const unsigned long long heads = __ballot(head);
    const int src = 63 - __clzll(heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull)));
    slot_i = __shfl(slot_i, src);
    if (i < n) slot_of[i] = slot_i;
}

// Two-kernel exclusive scan of 0/1 flags (2048 elements per block), the flags computed on the fly:
//   k_scan_count   block b: number of set flags in its 2048 elements -> bsum[b]
//   k_scan_write   block b: its base = sum of bsum[0..b) (every block adds the few thousand block counts up itself: they
//                  sit in L2, and it saves the single-block middle kernel and its two launch gaps), then the exclusive
//                  positions of its elements; the last block stores the total.
// FLAG 1: nbr[i] >= 0 (pair lists).  FLAG 2: input row i is the representative of its voxel (coordinate maps).
#define SCAN_ELEMS 2048
struct ScanSrc { const int32_t *a; const int32_t *b; };
template <int FLAG> __device__ static inline int32_t scan_flag(const ScanSrc &S, int64_t i) {
    if (FLAG == 1) return S.a[i] >= 0 ? 1 : 0;
    const int32_t sl = S.b[i];                                   // FLAG 2: a = table values, b = slot of row i
    return (sl >= 0 && S.a[sl] == (int32_t)i) ? 1 : 0;
}
__device__ static inline int32_t block_sum_256(int32_t v, int32_t *red) {       // sum over the 256 threads, to all
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const int32_t t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}
template <int FLAG>
__global__ __launch_bounds__(256) void k_scan_count(ScanSrc S, int64_t n, int32_t *__restrict__ bsum) {
    __shared__ int32_t red[4];
    const int64_t base = blockIdx.x * (int64_t)SCAN_ELEMS + threadIdx.x * 8;
    int32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (base + j < n) s += scan_flag<FLAG>(S, base + j);
    const int32_t t = block_sum_256(s, red);
    if (threadIdx.x == 0) bsum[blockIdx.x] = t;
}
template <int FLAG>
__global__ __launch_bounds__(256) void k_scan_write(ScanSrc S, int64_t n, const int32_t *__restrict__ bsum,
                                                    int32_t *__restrict__ pos, int32_t *__restrict__ total) {
    __shared__ int32_t red[4];
    __shared__ int32_t wsum[4];
    // base of this block
    int32_t pre = 0;
    for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x; i += 256) pre += bsum[i];
    const int32_t blk_base = block_sum_256(pre, red);
    const int64_t base = blockIdx.x * (int64_t)SCAN_ELEMS + threadIdx.x * 8;
    int32_t v[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        v[j] = (base + j < n) ? scan_flag<FLAG>(S, base + j) : 0;
        s += v[j];
    }
    // exclusive prefix of s over the block: inclusive wave scan, then the waves before
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int32_t run = blk_base + inc - s;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (base + j < n) pos[base + j] = run;
        run += v[j];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *total = run;       // the last thread has walked past every element
}

__global__ void k_table_init(unsigned long long *__restrict__ keys, int32_t *__restrict__ vals, int64_t cap, int32_t *n_out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < cap) { keys[i] = ~0ull; vals[i] = 0x7f7f7f7f; }
    if (i < 2) n_out[i] = 0;
}

__global__ void k_compact(const int32_t *__restrict__ coords, int64_t n, int32_t qs,
                          const int32_t *__restrict__ vals, const int32_t *__restrict__ slot_of,
                          const int32_t *__restrict__ pos, int32_t *out_coords, int32_t *unique_index,
                          int32_t *inverse) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t s = slot_of[i];
    if (s < 0) { inverse[i] = -1; return; }
    int32_t rep = vals[s];
    int32_t m = pos[rep];
    inverse[i] = m;
    if (rep == (int32_t)i) {
        int4 c;
        uint64_t key;
        quantised_key(coords, i, qs, &c, &key);
        reinterpret_cast<int4 *>(out_coords)[m] = c;
        unique_index[m] = (int32_t)i;
    }
}
__global__ void k_retarget(int64_t n, int32_t *vals, const int32_t *__restrict__ slot_of,
                           const int32_t *__restrict__ unique_index, const int32_t *__restrict__ n_out) {
    // table value: representative input row -> compact output row
    int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (m >= *n_out) return;
    vals[slot_of[unique_index[m]]] = (int32_t)m;
}

extern "C" int cg3d_coord_map_build(const int32_t *coords, int64_t n, int32_t qstride, uint64_t *keys, int32_t *vals,
                                    int64_t cap, void *ws, int32_t *out_coords, int32_t *unique_index,
                                    int32_t *inverse, int32_t *n_out, cg3d_stream_t stream) {
    if (n < 0 || qstride < 1 || cap < 2 * n || (cap & (cap - 1)) || cap > (1LL << 30)) return CG3D_ERR_ARG;
    if (((uintptr_t)coords & 15) || ((uintptr_t)out_coords & 15)) return CG3D_ERR_ARG;
    {   // k_insert's ballots and shuffles are written for 64-lane wavefronts: checked once per process, not assumed
        static int wave = 0;
        if (!wave) {
            int dev = 0, w = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&w, hipDeviceAttributeWarpSize, dev) != hipSuccess) return CG3D_ERR_LAUNCH;
            wave = w;
        }
        if (wave != 64) return CG3D_ERR_LAUNCH;
    }
    hipStream_t s = cg3d_hs(stream);
    hipLaunchKernelGGL(k_table_init, dim3((unsigned)cg3d_divup(cap, 256)), dim3(256), 0, s, (unsigned long long *)keys, vals, cap,
                       n_out);                        // empty keys (all ones), "no row" values, row count and status = 0: one launch
    if (n == 0) { CG3D_CHECK_LAUNCH(); return CG3D_OK; }
    int32_t *slot_of = (int32_t *)ws;
    int32_t *pos = slot_of + n;
    int64_t nb = cg3d_divup(n, SCAN_ELEMS);
    int32_t *bsum = pos + n;
    int32_t *status = n_out + 1;           // read back by the caller together with the row count
    unsigned g = (unsigned)cg3d_divup(n, 256);
    hipLaunchKernelGGL(k_insert, dim3(g), dim3(256), 0, s, coords, n, qstride, (unsigned long long *)keys, vals,
                       (uint64_t)(cap - 1), slot_of, status);
    const ScanSrc src = {vals, slot_of};
    hipLaunchKernelGGL(k_scan_count<2>, dim3((unsigned)nb), dim3(256), 0, s, src, n, bsum);
    hipLaunchKernelGGL(k_scan_write<2>, dim3((unsigned)nb), dim3(256), 0, s, src, n, bsum, pos, n_out);
    hipLaunchKernelGGL(k_compact, dim3(g), dim3(256), 0, s, coords, n, qstride, vals, slot_of, pos, out_coords,
                       unique_index, inverse);
    hipLaunchKernelGGL(k_retarget, dim3(g), dim3(256), 0, s, n, vals, slot_of, unique_index, n_out);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------ Morton row order
// order[i] = input row of the i-th row in (batch, Morton(x, y, z)) order -- the row order every map inserted by the host
// engine is built in, so that 128 consecutive rows are a spatially compact patch (the tile plans of spconv_tile.hip stage
// the distinct neighbour rows of such a patch in LDS: ~2 x 128 rows when the rows are Morton ordered, ~11 x 128 in the
// random order points arrive in).  The strided maps derived from a Morton-ordered map inherit the order: the first
// occurrence of floor(c / s) * s along a Morton-sorted sequence is Morton sorted again (bit interleaving nests).
__device__ static inline uint64_t spread3(uint32_t v) {           // 15 bits -> every third bit
    uint64_t x = v & 0x7fffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__global__ void k_morton_keys(const int32_t *__restrict__ coords, int64_t n, unsigned long long *__restrict__ keys,
                              int32_t *__restrict__ vals) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int4 c = reinterpret_cast<const int4 *>(coords)[i];
    const uint32_t ux = (uint32_t)(c.y + CG3D_COORD_LIMIT), uy = (uint32_t)(c.z + CG3D_COORD_LIMIT), uz = (uint32_t)(c.w + CG3D_COORD_LIMIT);
    unsigned long long key = ~0ull;                                // out of range: last (cg3d_coord_map_build reports it)
    if ((uint32_t)c.x < (uint32_t)CG3D_BATCH_LIMIT && (ux | uy | uz) < (uint32_t)(2 * CG3D_COORD_LIMIT))
        key = ((unsigned long long)c.x << 45) | (spread3(ux) << 2) | (spread3(uy) << 1) | spread3(uz);
    keys[i] = key;
    vals[i] = (int32_t)i;
}
static size_t morton_sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, (unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                  (int32_t *)nullptr, (int32_t *)nullptr, (size_t)n, 0, 64, (hipStream_t)0) != hipSuccess)
        bytes = (size_t)n * 32 + (1 << 20);
    return (bytes + 255) & ~(size_t)255;
}
extern "C" int64_t cg3d_morton_order_ws_bytes(int64_t n) {
    const int64_t n8 = (n * 8 + 255) & ~255ll, n4 = (n * 4 + 255) & ~255ll;
    return 2 * n8 + n4 + (int64_t)morton_sort_temp_bytes(n > 0 ? n : 1);
}
extern "C" int cg3d_morton_order(const int32_t *coords, int64_t n, int32_t *order, void *ws, cg3d_stream_t stream) {
    if (n < 0 || n >= (1ll << 31) || ((uintptr_t)coords & 15)) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipStream_t s = cg3d_hs(stream);
    char *w = (char *)ws;
    const int64_t n8 = (n * 8 + 255) & ~255ll, n4 = (n * 4 + 255) & ~255ll;
    unsigned long long *keys = (unsigned long long *)w, *keys_s = (unsigned long long *)(w + n8);
    int32_t *vals = (int32_t *)(w + 2 * n8);
    void *temp = w + 2 * n8 + n4;
    size_t temp_bytes = morton_sort_temp_bytes(n);
    hipLaunchKernelGGL(k_morton_keys, dim3((unsigned)cg3d_divup(n, 256)), dim3(256), 0, s, coords, n, keys, vals);
    // radix sort is stable: rows of one voxel keep their input order, the representative stays the first occurrence
    if (rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_s, vals, order, (size_t)n, 0, 64, s) != hipSuccess) return CG3D_ERR_LAUNCH;
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------ kernel map
__global__ void k_kernel_map(const int32_t *__restrict__ q, int64_t nq, const int32_t *__restrict__ off, int32_t K,
                             const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals, uint64_t capm1,
                             int32_t *__restrict__ nbr) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= nq) return;
    int4 c = reinterpret_cast<const int4 *>(q)[i];
    for (int32_t k = blockIdx.y; k < K; k += gridDim.y) {
        uint64_t key;
        int32_t r = -1;
        if (cg3d_pack(c.x, c.y + off[k * 3], c.z + off[k * 3 + 1], c.w + off[k * 3 + 2], &key))
            r = cg3d_lookup(keys, vals, capm1, key);
        nbr[(int64_t)k * nq + i] = r;
    }
}
extern "C" int cg3d_kernel_map(const int32_t *q, int64_t nq, const int32_t *off, int32_t K, const uint64_t *keys,
                               const int32_t *vals, int64_t cap, int32_t *nbr, cg3d_stream_t stream) {
    if (nq < 0 || K < 1 || ((uintptr_t)q & 15)) return CG3D_ERR_ARG;
    if (nq == 0) return CG3D_OK;
    unsigned gy = (unsigned)(K < 1024 ? K : 1024);
    hipLaunchKernelGGL(k_kernel_map, dim3((unsigned)cg3d_divup(nq, 256), gy), dim3(256), 0, cg3d_hs(stream), q, nq, off,
                       K, keys, vals, (uint64_t)(cap - 1), nbr);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// Map of a coordinate map ONTO ITSELF with a centred odd kernel (offsets symmetric: off[K-1-k] == -off[k], centre 0):
// nbr[K-1-k][j] == i  <=>  nbr[k][i] == j, so only the first K/2 offsets are looked up in the hash table; every hit writes
// its mirror entry too, the centre is the identity, and the mirrored half starts out as -1.  Bit-identical to
// cg3d_kernel_map on the same inputs with half the probes.
__global__ void k_kernel_map_self(const int32_t *__restrict__ q, int64_t n, const int32_t *__restrict__ off, int32_t K,
                                  const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals, uint64_t capm1,
                                  int32_t *__restrict__ nbr) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 c = reinterpret_cast<const int4 *>(q)[i];
    const int32_t half = K >> 1;
    if (blockIdx.y == 0) nbr[(int64_t)half * n + i] = (int32_t)i;
    for (int32_t k = blockIdx.y; k < half; k += gridDim.y) {
        uint64_t key;
        int32_t r = -1;
        if (cg3d_pack(c.x, c.y + off[k * 3], c.z + off[k * 3 + 1], c.w + off[k * 3 + 2], &key))
            r = cg3d_lookup(keys, vals, capm1, key);
        nbr[(int64_t)k * n + i] = r;
        if (r >= 0) nbr[(int64_t)(K - 1 - k) * n + r] = (int32_t)i;
    }
}
extern "C" int cg3d_kernel_map_self(const int32_t *q, int64_t n, const int32_t *off, int32_t K, const uint64_t *keys,
                                    const int32_t *vals, int64_t cap, int32_t *nbr, cg3d_stream_t stream) {
    if (n < 0 || K < 1 || !(K & 1) || ((uintptr_t)q & 15)) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipStream_t s = cg3d_hs(stream);
    const int32_t half = K >> 1;
    if (half > 0 && hipMemsetAsync(nbr + (int64_t)(half + 1) * n, 0xFF, (size_t)half * n * sizeof(int32_t), s) != hipSuccess)
        return CG3D_ERR_LAUNCH;
    unsigned gy = (unsigned)(half < 1 ? 1 : (half < 1024 ? half : 1024));
    hipLaunchKernelGGL(k_kernel_map_self, dim3((unsigned)cg3d_divup(n, 256), gy), dim3(256), 0, s, q, n, off, K, keys, vals,
                       (uint64_t)(cap - 1), nbr);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// Transposed kernel map by scattering the map itself: nbrT[k][i] = o  <=>  nbr[k][o] = i (a kernel map is injective per
// offset), so the transposed map of a strided / transposed convolution needs no hash lookups -- K x n_out coalesced reads
// and one 4-byte write per pair instead of K x n_in probes on the (larger) fine side.  nbrT starts out as -1.
__global__ void k_kernel_map_transpose(const int32_t *__restrict__ nbr, int64_t total, int64_t n_out, int64_t n_in,
                                       int32_t *__restrict__ nbrT) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int32_t i = nbr[t];
    if (i >= 0) nbrT[(t / n_out) * n_in + i] = (int32_t)(t % n_out);
}
extern "C" int cg3d_kernel_map_transpose(const int32_t *nbr, int32_t K, int64_t n_out, int64_t n_in, int32_t *nbrT,
                                         cg3d_stream_t stream) {
    if (K < 1 || n_out < 0 || n_in < 0 || (int64_t)K * n_out >= (1ll << 40)) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (n_in > 0 && hipMemsetAsync(nbrT, 0xFF, (size_t)K * n_in * sizeof(int32_t), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    const int64_t total = (int64_t)K * n_out;
    if (total == 0 || n_in == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_kernel_map_transpose, dim3((unsigned)cg3d_divup(total, 256)), dim3(256), 0, s, nbr, total, n_out, n_in, nbrT);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------ interpolation map
__global__ void k_interp_map(const float *__restrict__ q, int64_t nq, int32_t ts, const uint64_t *__restrict__ keys,
                             const int32_t *__restrict__ vals, uint64_t capm1, int32_t *__restrict__ idx,
                             float *__restrict__ w) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int64_t i = t >> 3;
    int j = (int)(t & 7);
    if (i >= nq) return;
    float4 v = reinterpret_cast<const float4 *>(q)[i];
    const float fts = (float)ts;
    float qv[3] = {v.y, v.z, v.w};
    int dsel[3] = {(j >> 2) & 1, (j >> 1) & 1, j & 1};
    int32_t c[3];
    float wd[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        float fl = floorf(qv[d] / fts);
        float lo = fl * fts;
        float r = (qv[d] - lo) / fts;
        c[d] = (int32_t)lo + dsel[d] * ts;
        wd[d] = dsel[d] ? r : (1.0f - r);
    }
    float wt = (wd[0] * wd[1]) * wd[2];
    uint64_t key;
    int32_t r = -1;
    if (cg3d_pack((int32_t)v.x, c[0], c[1], c[2], &key)) r = cg3d_lookup(keys, vals, capm1, key);
    idx[t] = r;
    w[t] = wt;
}
extern "C" int cg3d_interp_map(const float *q, int64_t nq, int32_t ts, const uint64_t *keys, const int32_t *vals,
                               int64_t cap, int32_t *idx, float *w, cg3d_stream_t stream) {
    if (nq < 0 || ts < 1 || ((uintptr_t)q & 15)) return CG3D_ERR_ARG;
    if (nq == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_interp_map, dim3((unsigned)cg3d_divup(nq * 8, 256)), dim3(256), 0, cg3d_hs(stream), q, nq, ts,
                       keys, vals, (uint64_t)(cap - 1), idx, w);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------ pooling map
__global__ void k_pool_map(const int32_t *__restrict__ in, int64_t n_in, int32_t os, int32_t half,
                           const uint64_t *__restrict__ keys, const int32_t *__restrict__ vals, uint64_t capm1,
                           int32_t *__restrict__ pmap) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    int j = blockIdx.y;
    if (i >= n_in) return;
    int4 c = reinterpret_cast<const int4 *>(in)[i];
    int32_t cc[3] = {c.y, c.z, c.w};
    int32_t dd[3] = {j / 9 - 1, (j / 3) % 3 - 1, j % 3 - 1};
    int32_t o[3];
    bool ok = true;
#pragma unroll
    for (int d = 0; d < 3; d++) {
        o[d] = (cg3d_floordiv(cc[d], os) + dd[d]) * os;
        int32_t diff = o[d] - cc[d];
        diff = diff < 0 ? -diff : diff;
        ok = ok && (diff <= half);
    }
    int32_t r = -1;
    uint64_t key;
    if (ok && cg3d_pack(c.x, o[0], o[1], o[2], &key)) r = cg3d_lookup(keys, vals, capm1, key);
    pmap[(int64_t)j * n_in + i] = r;
}
extern "C" int cg3d_pool_map(const int32_t *in, int64_t n_in, int32_t out_stride, int32_t half_extent,
                             const uint64_t *keys, const int32_t *vals, int64_t cap, int32_t *pmap,
                             cg3d_stream_t stream) {
    if (n_in < 0 || out_stride < 1 || half_extent < 0 || ((uintptr_t)in & 15)) return CG3D_ERR_ARG;
    if (n_in == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_pool_map, dim3((unsigned)cg3d_divup(n_in, 256), 27), dim3(256), 0, cg3d_hs(stream), in, n_in,
                       out_stride, half_extent, keys, vals, (uint64_t)(cap - 1), pmap);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------ pair-compacted kernel maps
extern "C" int64_t cg3d_pairs_ws_bytes(int64_t total) { return (total + total / 1024 + 64) * (int64_t)sizeof(int32_t); }

// Round 4: the scanned positions are no longer stored (a 4-byte store and re-load per table entry, and a launch): the counting
// pass leaves only the per-block sums in `ws`; the offsets of the (offset, group) lists are summed from them by one wave per
// entry, and the fill pass redoes the in-block scan of its 2 048 entries while it writes the pairs.
__global__ __launch_bounds__(64) void k_pair_offsets(const int32_t *__restrict__ nbr, const int32_t *__restrict__ bsum, int32_t K,
                                                     int64_t n_out, const int32_t *__restrict__ row_bounds, int32_t G,
                                                     int32_t *__restrict__ pair_off) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const int64_t total = (int64_t)K * n_out;
    int64_t at = total;
    if (t < K * G) {
        const int k = t / G, g = t % G;
        at = (int64_t)k * n_out + (row_bounds ? row_bounds[g] : 0);
        if (at > total) at = total;
    }
    const int64_t blk = at / SCAN_ELEMS;
    int32_t s = 0;
    for (int64_t i = lane; i < blk; i += 64) s += bsum[i];
    for (int64_t j = blk * SCAN_ELEMS + lane; j < at; j += 64) s += nbr[j] >= 0 ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) pair_off[t] = s;
}
extern "C" int cg3d_pairs_count(const int32_t *nbr, int32_t K, int64_t n_out, const int32_t *row_bounds, int32_t G,
                                void *ws, int32_t *pair_off, cg3d_stream_t stream) {
    if (K < 1 || n_out < 0 || G < 1) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    const int64_t total = (int64_t)K * n_out;
    if (total >= (1LL << 31)) return CG3D_ERR_ARG;
    if (total == 0) {
        if (hipMemsetAsync(pair_off, 0, ((int64_t)K * G + 1) * sizeof(int32_t), s) != hipSuccess) return CG3D_ERR_LAUNCH;
        return CG3D_OK;
    }
    int32_t *bsum = (int32_t *)ws;
    const int64_t nb = cg3d_divup(total, SCAN_ELEMS);
    const ScanSrc src = {nbr, nullptr};
    hipLaunchKernelGGL(k_scan_count<1>, dim3((unsigned)nb), dim3(256), 0, s, src, total, bsum);
    hipLaunchKernelGGL(k_pair_offsets, dim3((unsigned)((int64_t)K * G + 1)), dim3(64), 0, s, nbr, bsum, K, n_out, row_bounds, G, pair_off);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
__global__ __launch_bounds__(256) void k_pair_fill(const int32_t *__restrict__ nbr, int64_t total, int64_t n_out,
                                                   const int32_t *__restrict__ bsum, int32_t *__restrict__ pin,
                                                   int32_t *__restrict__ pout) {
    __shared__ int32_t red[4];
    __shared__ int32_t wsum[4];
    int32_t pre = 0;
    for (int64_t i = threadIdx.x; i < (int64_t)blockIdx.x; i += 256) pre += bsum[i];
    const int32_t blk_base = block_sum_256(pre, red);
    const int64_t base = blockIdx.x * (int64_t)SCAN_ELEMS + threadIdx.x * 8;
    int32_t v[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        v[j] = (base + j < total) ? nbr[base + j] : -1;
        s += v[j] >= 0 ? 1 : 0;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int32_t inc = s;
    for (int o = 1; o < 64; o <<= 1) {
        const int32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int32_t run = blk_base + inc - s;
    for (int w = 0; w < wave; w++) run += wsum[w];
#pragma unroll
    for (int j = 0; j < 8; j++)
        if (v[j] >= 0) {
            pin[run] = v[j];
            pout[run] = (int32_t)((base + j) % n_out);
            run++;
        }
}
extern "C" int cg3d_pairs_fill(const int32_t *nbr, int32_t K, int64_t n_out, const void *ws, int32_t *pair_in,
                               int32_t *pair_out, cg3d_stream_t stream) {
    if (K < 1 || n_out < 0) return CG3D_ERR_ARG;
    const int64_t total = (int64_t)K * n_out;
    if (total == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_pair_fill, dim3((unsigned)cg3d_divup(total, SCAN_ELEMS)), dim3(256), 0, cg3d_hs(stream), nbr, total, n_out,
                       (const int32_t *)ws, pair_in, pair_out);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
