// spconv_tile.hip -- sparse convolution forward / data gradient on LDS-staged neighbour tiles (gfx950).
//
// Replaces MinkowskiEngine's ConvolutionForwardGPU / ConvolutionBackwardGPU(dX) (un-vendored) for the
// call sites pcdet/models/backbones_3d/biresnet.py:358-406 (every 3^3 convolution of BiResNet) and
// dense_heads/cagroup_head.py:259-275 (the grouped class-branch convolutions).
//
// The output-stationary kernel of spconv.hip gathers the neighbour row of every (offset, output row) from global
// memory: K x 128 rows per workgroup although a tile of 128 spatially coherent output rows only touches ~2 x 128
// DISTINCT input rows (rows are Morton ordered, cg3d_coord_map_build_sorted).  Here the gather is split in two:
//
//   cg3d_tile_plan_build   once per kernel map: for every tile of 128 output rows the list of distinct input rows its
//                          K offsets touch (`ulist`), and for every (offset, row) the position of the neighbour in
//                          that list (`slots`, 0 = absent) -- a wave-ballot / prefix-sum compaction over an LDS hash.
//   cg3d_tile_row_order    (optional, sparse maps) a permutation of the output rows that groups rows with the same set of
//                          live offsets inside windows of 1024 rows: the tiles are cut from the permuted order.
//   cg3d_spconv_tile_fwd   (spconv_tile2.hip) per launch: a workgroup stages the distinct rows of its tile ONCE into LDS by
//                          LDS-DMA, then runs all K offsets' MFMAs with the A fragments read from LDS through the slot table
//                          and the weight fragments streamed from L2 in MFMA fragment order.
// This file: the plan builder, the row order and the fragment-order weight copies.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "cg3d_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define TP_TM 128            // output rows per tile
#define TP_KB 32             // offsets whose slot table is resident in LDS at a time
#define TP_HASH 2048         // LDS hash entries of the plan builder (ucap <= 1023)

// ------------------------------------------------------------------------------------------------ tile plan
// One WAVE per tile, lane l = output rows l and l + 64 of the tile: every step of the loop over the offsets is
// wave-synchronous (no workgroup barrier; the first version ran two waves with five barriers per offset and took ~2 us
// per offset -- 1.5 ms for a 9^3 map), and the neighbour rows of the next TP_PF offsets are requested ahead.  Offsets are
// processed in order; the neighbour rows of one offset are distinct (a kernel map is injective per offset), so "new"
// rows get their slots from a ballot + popcount prefix sum in row order.  When an offset would push the number of
// staged rows past `ucap` the current pass is closed and a new one starts with that offset (the conv kernel restages
// per pass).
#define TP_PF 8
#define TP_CHUNK_K 96       // offsets per wave when a large kernel (5^3, 9^3) is split over the waves of the workgroup
// Large K: the offsets are cut into up to 8 runs, one WAVE each, with their own hash table and pass list (any partition
// of the offsets into passes is a valid plan; a run boundary just forces a pass boundary), so a 9^3 map takes the time
// of ~92 offsets instead of 729 (0.55 ms -> ~0.1 ms per plan).  Wave w writes its passes at pass_tab[tile][k_lo(w) + i];
// wave 0 closes the gaps at the end.
__global__ __launch_bounds__(512) void k_tile_plan(const int32_t *__restrict__ nbr, int32_t K, int64_t n_out,
                                                   const int32_t *__restrict__ tiles, int32_t ucap, int32_t maxpass,
                                                   uint16_t *__restrict__ slots, uint8_t *__restrict__ live,
                                                   int32_t *__restrict__ pass_tab, int32_t *__restrict__ npass,
                                                   int32_t *__restrict__ ulist, int64_t ulist_cap,
                                                   int32_t *__restrict__ cursor, const int32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) uint8_t plan_smem[];
    __shared__ int32_t np_of[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    int32_t *hkey = reinterpret_cast<int32_t *>(plan_smem + (size_t)wave * (TP_HASH * 6 + 4096));
    uint16_t *hval = reinterpret_cast<uint16_t *>(hkey + TP_HASH);
    int32_t *ul = reinterpret_cast<int32_t *>(hval + TP_HASH);
    const int64_t tile = blockIdx.x;
    int64_t row0 = tile * TP_TM;
    int rows = (int)(n_out - row0 < TP_TM ? n_out - row0 : TP_TM);
    if (tiles) { row0 = tiles[tile * 3 + 1]; rows = tiles[tile * 3 + 2]; }
    const bool ok0 = lane < rows, ok1 = lane + 64 < rows;
    uint16_t *slots_t = slots + tile * (int64_t)K * TP_TM;
    uint8_t *live_t = live + tile * (int64_t)K;
    int32_t *ptab = pass_tab + tile * (int64_t)maxpass * 4;
    const uint64_t below = (1ull << lane) - 1ull;
    const int k_lo = (int)((int64_t)K * wave / nwave), k_hi = (int)((int64_t)K * (wave + 1) / nwave);   // this wave's offsets

    // one wave: its LDS operations execute in program order, so this only has to stop the COMPILER from moving LDS
    // accesses across (a workgroup fence over all address spaces would also wait for the prefetched global loads)
    auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup", "local"); __builtin_amdgcn_wave_barrier(); };
    auto clear = [&]() {
        for (int i = lane; i < TP_HASH; i += 64) hkey[i] = -1;
        wave_sync();
    };
    auto lookup = [&](int32_t g) -> int {          // slot of row g, 0 = not in the table
        uint32_t h = ((uint32_t)g * 2654435761u) >> 21;            // 11 bits
        for (;;) {
            const int32_t k = hkey[h];
            if (k == g) return hval[h];
            if (k == -1) return 0;
            h = (h + 1) & (TP_HASH - 1);
        }
    };
    int ucount = 0, pass_k0 = k_lo, np = 0;
    auto flush = [&](int k1) {                      // close the pass [pass_k0, k1) with ucount staged rows
        wave_sync();
        int32_t base = 0;
        const int at = k_lo + np;                   // np <= offsets done so far: stays inside this wave's range of pass_tab
        if (lane == 0) {
            base = atomicAdd(&cursor[0], ucount);
            if ((int64_t)base + ucount > ulist_cap || at >= maxpass) { cursor[1] = 1; base = 0; }
            if (at < maxpass) { ptab[at * 4] = pass_k0; ptab[at * 4 + 1] = k1; ptab[at * 4 + 2] = base; ptab[at * 4 + 3] = ucount; }
        }
        base = __shfl(base, 0);
        if ((int64_t)base + ucount <= ulist_cap)
            for (int i = lane; i < ucount; i += 64) ulist[base + i] = ul[i];
        np++;
    };
    auto insert = [&](int32_t g, int s) -> int {    // claim slot s for row g; -1 = already there (a duplicate within the offset)
        uint32_t h = ((uint32_t)g * 2654435761u) >> 21;
        for (;;) {
            const int32_t prev = atomicCAS(&hkey[h], -1, g);
            if (prev == -1) { hval[h] = (uint16_t)s; ul[s - 1] = g; return s; }
            if (prev == g) return -1;
            h = (h + 1) & (TP_HASH - 1);
        }
    };
    clear();
    int32_t pf0[TP_PF], pf1[TP_PF];
    // (unconditional loads from clamped addresses: a load under a divergent branch makes the compiler wait for ALL
    // outstanding memory operations right after it -- no prefetch)
    // position p of the tile is output row order[row0 + p] (cg3d_tile_row_order) or row0 + p
    int64_t r0c = row0 + (ok0 ? lane : 0), r1c = row0 + (ok1 ? lane + 64 : 0);
    if (order) { r0c = order[r0c]; r1c = order[r1c]; }
    auto fetch = [&](int k, int32_t &g0, int32_t &g1) {
        const int32_t *src = nbr + (int64_t)(k < K ? k : K - 1) * n_out;
        g0 = src[r0c];
        g1 = src[r1c];
    };
#pragma unroll
    for (int j = 0; j < TP_PF; j++) fetch(k_lo + j, pf0[j], pf1[j]);
    for (int kb = k_lo; kb < k_hi; kb += TP_PF) {
#pragma unroll
        for (int j = 0; j < TP_PF; j++) {
            const int k = kb + j;
            const int32_t g0 = ok0 ? pf0[j] : -1, g1 = ok1 ? pf1[j] : -1;
            fetch(k + TP_PF, pf0[j], pf1[j]);      // this register pair's next use
            if (k < k_hi) {
                int s0 = g0 >= 0 ? lookup(g0) : 0, s1 = g1 >= 0 ? lookup(g1) : 0;
                bool new0 = g0 >= 0 && s0 == 0, new1 = g1 >= 0 && s1 == 0;
                uint64_t b0 = __ballot(new0), b1 = __ballot(new1);
                int newcount = __popcll(b0) + __popcll(b1);
                if (ucount + newcount > ucap) {             // uniform: close the pass before this offset
                    flush(k);
                    clear();
                    ucount = 0;
                    pass_k0 = k;
                    new0 = g0 >= 0; new1 = g1 >= 0;
                    s0 = s1 = 0;
                    b0 = __ballot(new0); b1 = __ballot(new1);
                    newcount = __popcll(b0) + __popcll(b1);
                }
                if (new0) s0 = insert(g0, ucount + __popcll(b0 & below) + 1);
                if (new1) s1 = insert(g1, ucount + __popcll(b0) + __popcll(b1 & below) + 1);
                ucount += newcount;
                wave_sync();
                if (s0 < 0) s0 = lookup(g0);
                if (s1 < 0) s1 = lookup(g1);
                slots_t[(int64_t)k * TP_TM + lane] = (uint16_t)s0;
                slots_t[(int64_t)k * TP_TM + lane + 64] = (uint16_t)s1;
                // liveness of the four 32-row blocks of the tile for this offset
                const uint64_t l0 = __ballot(g0 >= 0), l1 = __ballot(g1 >= 0);
                if (lane == 0)
                    live_t[k] = (uint8_t)(((l0 & 0xffffffffull) ? 1 : 0) | ((l0 >> 32) ? 2 : 0) | ((l1 & 0xffffffffull) ? 4 : 0) | ((l1 >> 32) ? 8 : 0));
            }
        }
    }
    if (k_hi > k_lo) flush(k_hi);
    if (lane == 0) np_of[wave] = np;
    __threadfence_block();
    __syncthreads();
    if (wave == 0) {
        // close the gaps between the waves' pass lists (wave w's entries start at its k_lo); 4 ints per entry, lanes 0-3
        int total = np_of[0];
        for (int w = 1; w < nwave; w++) {
            const int src0 = (int)((int64_t)K * w / nwave), cnt = np_of[w];
            if (src0 != total && lane < 4)
                for (int i = 0; i < cnt; i++)
                    if (src0 + i < maxpass && total + i < maxpass) ptab[(total + i) * 4 + lane] = ptab[(src0 + i) * 4 + lane];
            total += cnt;
        }
        if (lane == 0) npass[tile] = total < maxpass ? total : maxpass;
    }
}

extern "C" int cg3d_tile_plan_build(const int32_t *nbr, int32_t K, int64_t n_out, const int32_t *tiles, int64_t ntile,
                                    int32_t ucap, int32_t maxpass, uint16_t *slots, uint8_t *live, int32_t *pass_tab,
                                    int32_t *npass, int32_t *ulist, int64_t ulist_cap, int32_t *cursor,
                                    const int32_t *order, cg3d_stream_t stream) {
    if (K < 1 || n_out < 0 || ntile < 0 || ucap < TP_TM || ucap > 1023 || maxpass < 1 || (order && tiles)) return CG3D_ERR_ARG;
    if (!tiles && ntile != cg3d_divup(n_out, TP_TM)) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (hipMemsetAsync(cursor, 0, 2 * sizeof(int32_t), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (ntile == 0) return CG3D_OK;
    const int nwave = K <= 32 ? 1 : (int)(cg3d_divup(K, TP_CHUNK_K) < 8 ? cg3d_divup(K, TP_CHUNK_K) : 8);
    const size_t plan_lds = (size_t)nwave * (TP_HASH * 6 + 4096);          // per wave: hash keys + values, the pass's row list
    static bool plan_attr = false;
    if (!plan_attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_plan), hipFuncAttributeMaxDynamicSharedMemorySize,
                                8 * (TP_HASH * 6 + 4096)) != hipSuccess)
            return CG3D_ERR_LAUNCH;
        plan_attr = true;
    }
    hipLaunchKernelGGL(k_tile_plan, dim3((unsigned)ntile), dim3(64 * nwave), plan_lds, s, nbr, K, n_out, tiles, ucap, maxpass,
                       slots, live, pass_tab, npass, ulist, ulist_cap, cursor, order);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ row order
// One workgroup per window of TP_WINDOW output rows, one thread per row: signature = bit k set iff the row has a
// neighbour at offset k (K <= 32), then a bitonic sort of (signature, row) in LDS -- stable by construction (the row index
// is the low half of the key).  On the transposed map of a strided convolution / the map of a transposed convolution a row
// has neighbours only at the offsets of its parity class: in arrival (Morton) order every 128-row tile holds all classes
// and multiplies 7-8 x the rows it needs; cut from the sorted window a tile holds one or two classes (2.1-2.7 x).
#define TP_WINDOW 1024
__global__ __launch_bounds__(TP_WINDOW) void k_tile_row_order(const int32_t *__restrict__ nbr, int32_t K, int64_t n_out,
                                                              int32_t *__restrict__ order) {
    __shared__ unsigned long long key[TP_WINDOW];
    const int t = threadIdx.x, window = blockDim.x;
    const int64_t row = (int64_t)blockIdx.x * window + t;
    unsigned long long kv = ~0ull;                       // rows past the end sort behind every real row
    if (row < n_out) {
        uint32_t sig = 0;
        for (int k = 0; k < K; k++) sig |= (nbr[(int64_t)k * n_out + row] >= 0 ? 1u : 0u) << k;
        kv = ((unsigned long long)sig << 32) | (unsigned)t;
    }
    key[t] = kv;
    __syncthreads();
    for (int size = 2; size <= window; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = t ^ stride;
            if (partner > t) {
                const unsigned long long a = key[t], b = key[partner];
                const bool up = (t & size) == 0;
                if ((a > b) == up) { key[t] = b; key[partner] = a; }
            }
            __syncthreads();
        }
    if (row < n_out) order[row] = (int32_t)((int64_t)blockIdx.x * window + (int64_t)(key[t] & 0xffffffffull));
}
extern "C" int cg3d_tile_row_order(const int32_t *nbr, int32_t K, int64_t n_out, int32_t window, int32_t *order,
                                   cg3d_stream_t stream) {
    if (K < 1 || K > 32 || n_out < 0 || window < 128 || window > TP_WINDOW || (window & (window - 1))) return CG3D_ERR_ARG;
    if (n_out == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_tile_row_order, dim3((unsigned)cg3d_divup(n_out, window)), dim3((unsigned)window), 0, cg3d_hs(stream), nbr, K,
                       n_out, order);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ weights
// fp32 [slot][cin][cout] -> bf16 in MFMA B-fragment order, so that a wave's weight fragment is ONE contiguous 1 KB
// load (64 lanes x 16 B):
//   transposed copy (forward):       Wf_t[slot][nt = co/32][ks = ci/16][lane = (ci/8 & 1)*32 + co%32][j = ci%8]
//   plain copy (data gradient, the swapped problem cin' = cout, cout' = cin):
//                                    Wf  [slot][nt = ci/32][ks = co/16][lane = (co/8 & 1)*32 + ci%32][j = co%8]
// cin % 16 == 0 and cout % 32 == 0 (transposed copy), cout % 16 == 0 and cin % 32 == 0 (plain copy).
__device__ static inline uint32_t tf2bf(float f) {        // round-to-nearest-even, as spconv.hip / the oracle
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__global__ __launch_bounds__(256) void k_prep_weights_frag(const float *__restrict__ W0, const float *const *__restrict__ Ws,
                                                           uint16_t *__restrict__ Wf_t, uint16_t *__restrict__ Wf,
                                                           int64_t slots_per, int32_t cin, int32_t cout) {
    const int64_t slot = blockIdx.x;
    const int64_t per = (int64_t)cin * cout;
    const float *src = Ws ? Ws[slot / slots_per] + (slot % slots_per) * per : W0 + slot * per;
    for (int64_t i = (int64_t)blockIdx.y * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.y * 256) {
        const int ci = (int)(i / cout), co = (int)(i % cout);
        const uint16_t b = (uint16_t)tf2bf(src[i]);
        if (Wf_t) Wf_t[slot * per + cg3d_frag_index(co, ci, cin)] = b;
        if (Wf) Wf[slot * per + cg3d_frag_index(ci, co, cout)] = b;
    }
}
extern "C" int cg3d_spconv_prep_weights_frag(const float *W0, const float *const *Ws, uint16_t *Wf_t, uint16_t *Wf,
                                             int32_t G, int64_t slots_per, int32_t cin, int32_t cout,
                                             cg3d_stream_t stream) {
    if (G < 1 || slots_per < 0 || cin < 1 || cout < 1 || (!W0 && !Ws) || (!Wf_t && !Wf)) return CG3D_ERR_ARG;
    if (Wf_t && ((cin & 15) || (cout & 31))) return CG3D_ERR_ARG;
    if (Wf && ((cout & 15) || (cin & 31))) return CG3D_ERR_ARG;
    const int64_t slots = (int64_t)G * slots_per;
    if (slots == 0) return CG3D_OK;
    if (slots > 0x7fffffffll) return CG3D_ERR_RANGE;
    const unsigned gy = (unsigned)(cg3d_divup((int64_t)cin * cout, 2048) < 64 ? cg3d_divup((int64_t)cin * cout, 2048) : 64);
    hipLaunchKernelGGL(k_prep_weights_frag, dim3((unsigned)slots, gy), dim3(256), 0, cg3d_hs(stream), W0, Ws, Wf_t, Wf,
                       slots_per, cin, cout);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

