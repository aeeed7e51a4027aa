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
//   cg3d_spconv_tile_fwd   per launch: a workgroup stages the distinct rows of its tile ONCE into LDS (coalesced
//                          16-byte loads, 64 input channels at a time), then runs all K offsets' MFMAs with the A
//                          fragments read from LDS through the slot table (ds_read_b128, XOR-swizzled rows) and the
//                          weight fragments streamed straight from L2 into registers in MFMA fragment order.
//
// Wave w of a workgroup owns ALL 128 rows x 32 output channels: its weight fragments are private (no LDS weight
// tile, no per-step barrier -- the barrier per 64-channel step was the bound of the previous kernels),
// the accumulators (4 x 16 registers) stay in registers over all offsets and every output row is stored once.
// Barriers: two per (pass, 64-channel chunk) of a tile, i.e. 2-16 per workgroup instead of one per step.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdio.h>
#include <stdlib.h>
#include "cg3d_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define TP_TM 128            // output rows per tile
#define TP_KB 32             // offsets whose slot table is resident in LDS at a time
#define TP_HASH 2048         // LDS hash entries of the plan builder (ucap <= 1023)

// ------------------------------------------------------------------------------------------------ tile plan
// One workgroup of 128 threads per tile, thread t = output row t of the tile.  Offsets are processed in order; the
// neighbour rows of one offset are distinct (a kernel map is injective per offset), so "new" rows get their slots from
// a ballot + popcount prefix sum.  When an offset would push the number of staged rows past `ucap` the current pass is
// closed and a new one starts with that offset (the conv kernel restages per pass).
__global__ __launch_bounds__(TP_TM) void k_tile_plan(const int32_t *__restrict__ nbr, int32_t K, int64_t n_out,
                                                      const int32_t *__restrict__ tiles, int32_t ucap, int32_t maxpass,
                                                      uint16_t *__restrict__ slots, uint8_t *__restrict__ live,
                                                      int32_t *__restrict__ pass_tab, int32_t *__restrict__ npass,
                                                      int32_t *__restrict__ ulist, int64_t ulist_cap,
                                                      int32_t *__restrict__ cursor) {
    __shared__ int32_t hkey[TP_HASH];
    __shared__ uint16_t hval[TP_HASH];
    __shared__ int32_t ul[1024];
    __shared__ int32_t wcnt[2];
    __shared__ int32_t sh_base;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int64_t tile = blockIdx.x;
    int64_t row0 = tile * TP_TM;
    int rows = (int)(n_out - row0 < TP_TM ? n_out - row0 : TP_TM);
    if (tiles) { row0 = tiles[tile * 3 + 1]; rows = tiles[tile * 3 + 2]; }
    const bool row_ok = t < rows;
    uint16_t *slots_t = slots + tile * (int64_t)K * TP_TM;
    uint8_t *live_t = live + tile * (int64_t)K;
    int32_t *ptab = pass_tab + tile * (int64_t)maxpass * 4;

    for (int i = t; i < TP_HASH; i += TP_TM) hkey[i] = -1;
    __syncthreads();
    int ucount = 0, pass_k0 = 0, np = 0;

    auto lookup = [&](int32_t g) -> int {          // slot of row g, 0 = not in the table
        uint32_t h = ((uint32_t)g * 2654435761u) >> 21;            // 11 bits
        for (;;) {
            const int32_t k = hkey[h];
            if (k == g) return hval[h];
            if (k == -1) return 0;
            h = (h + 1) & (TP_HASH - 1);
        }
    };
    auto flush = [&](int k1) {                      // close the pass [pass_k0, k1) with ucount staged rows
        __syncthreads();
        if (t == 0) {
            int32_t base = atomicAdd(&cursor[0], ucount);
            if ((int64_t)base + ucount > ulist_cap || np >= maxpass) { cursor[1] = 1; base = 0; }
            sh_base = base;
            if (np < maxpass) { ptab[np * 4] = pass_k0; ptab[np * 4 + 1] = k1; ptab[np * 4 + 2] = base; ptab[np * 4 + 3] = ucount; }
        }
        __syncthreads();
        const int32_t base = sh_base;
        if ((int64_t)base + ucount <= ulist_cap)
            for (int i = t; i < ucount; i += TP_TM) ulist[base + i] = ul[i];
        np++;
    };

    for (int k = 0; k < K; k++) {
        const int32_t g = row_ok ? nbr[(int64_t)k * n_out + row0 + t] : -1;
        int s = g >= 0 ? lookup(g) : 0;
        bool isnew = g >= 0 && s == 0;
        uint64_t bal = __ballot(isnew);
        if (lane == 0) wcnt[wave] = __popcll(bal);
        __syncthreads();
        int newcount = wcnt[0] + wcnt[1];
        if (ucount + newcount > ucap) {             // uniform: close the pass before this offset
            flush(k);
            for (int i = t; i < TP_HASH; i += TP_TM) hkey[i] = -1;
            ucount = 0;
            pass_k0 = k;
            isnew = g >= 0;
            s = 0;
            bal = __ballot(isnew);
            __syncthreads();
            if (lane == 0) wcnt[wave] = __popcll(bal);
            __syncthreads();
            newcount = wcnt[0] + wcnt[1];
        }
        if (isnew) {
            const int rank = __popcll(bal & ((1ull << lane) - 1ull)) + (wave ? wcnt[0] : 0);
            s = ucount + rank + 1;
            uint32_t h = ((uint32_t)g * 2654435761u) >> 21;
            for (;;) {
                const int32_t prev = atomicCAS(&hkey[h], -1, g);
                if (prev == -1) { hval[h] = (uint16_t)s; ul[s - 1] = g; break; }
                if (prev == g) { s = -1; break; }       // a duplicate within the offset (not a kernel map): resolved below
                h = (h + 1) & (TP_HASH - 1);
            }
        }
        ucount += newcount;
        __syncthreads();
        if (s < 0) s = lookup(g);
        slots_t[(int64_t)k * TP_TM + t] = (uint16_t)s;
        // liveness of the four 32-row blocks of the tile for this offset
        const uint64_t lb = __ballot(g >= 0);
        if (lane == 0) wcnt[wave] = ((lb & 0xffffffffull) ? 1 : 0) | ((lb >> 32) ? 2 : 0);
        __syncthreads();
        if (t == 0) live_t[k] = (uint8_t)(wcnt[0] | (wcnt[1] << 2));
        __syncthreads();
    }
    flush(K);
    if (t == 0) npass[tile] = np < maxpass ? np : maxpass;
}

extern "C" int cg3d_tile_plan_build(const int32_t *nbr, int32_t K, int64_t n_out, const int32_t *tiles, int64_t ntile,
                                    int32_t ucap, int32_t maxpass, uint16_t *slots, uint8_t *live, int32_t *pass_tab,
                                    int32_t *npass, int32_t *ulist, int64_t ulist_cap, int32_t *cursor,
                                    cg3d_stream_t stream) {
    if (K < 1 || n_out < 0 || ntile < 0 || ucap < TP_TM || ucap > 1023 || maxpass < 1) return CG3D_ERR_ARG;
    if (!tiles && ntile != cg3d_divup(n_out, TP_TM)) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (hipMemsetAsync(cursor, 0, 2 * sizeof(int32_t), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (ntile == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_tile_plan, dim3((unsigned)ntile), dim3(TP_TM), 0, s, nbr, K, n_out, tiles, ucap, maxpass, slots,
                       live, pass_tab, npass, ulist, ulist_cap, cursor);
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
__device__ static inline int64_t frag_index(int n_idx, int k_idx, int kdim) {
    // element (n_idx = MFMA column, k_idx = contraction index) of a [N][kdim] operand
    return ((((int64_t)(n_idx >> 5) * (kdim >> 4) + (k_idx >> 4)) * 64) + ((k_idx >> 3) & 1) * 32 + (n_idx & 31)) * 8 + (k_idx & 7);
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
        if (Wf_t) Wf_t[slot * per + frag_index(co, ci, cin)] = b;
        if (Wf) Wf[slot * per + frag_index(ci, co, cout)] = b;
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

// ------------------------------------------------------------------------------------------------ convolution
// Workgroup = 4 waves on one tile of 128 output rows x (NCO x 64) output channels.  Wave (g, h): h = its 64-channel
// output block, g = its share of the tile's live offsets (KG = 4 / NCO offset groups; the groups' partial sums meet
// through LDS at the end).  A wave's register tile is 128 rows x 64 channels = 4 x 2 MFMA blocks (128 accumulator
// registers): every A fragment read from LDS feeds 2 MFMAs and every weight fragment streamed from L2 feeds 4 -- the
// 128 x 32 tile of the first version read one A fragment per MFMA and ran into the LDS pipe (16 waves' indexed
// ds_read_b128, x1.8 bank conflicts) at a quarter of the matrix peak.
//
// LDS (dynamic): A tile (ucap + 1) x 128 B = the pass's distinct input rows, 64 channels at a time (row 0 = zeros;
// 16-byte granule g of row s sits at g ^ ((s >> 1) & 7): 16 lanes reading the same channel granule of 16 consecutive
// slots hit 16 different bank groups); slot table of the resident offsets [TP_KB][32][4] uint16 (lane r reads the slots
// of rows r, 32+r, 64+r, 96+r as ONE 8-byte word); list of live offsets.  The A tile doubles as the exchange buffer of
// the final reduction over offset groups.
//
// Software pipeline of one wave, unit = (offset, 16 channels) = 4 A fragments x 2 weight fragments -> 8 MFMAs on 8
// different accumulators: while a unit is on the matrix pipe the 4 A fragments of the next unit are on their way from
// LDS (two fragment sets) and each weight fragment is re-requested for the NEXT offset right after its last use, i.e.
// 3-4 units (~1000 cycles) ahead.
template <int NCO, int DBG>
__global__ __launch_bounds__(256, 2) void k_spconv_tile(
    const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wf, const uint16_t *__restrict__ slots,
    const uint8_t *__restrict__ live, const int32_t *__restrict__ pass_tab, const int32_t *__restrict__ npass,
    const int32_t *__restrict__ ulist, int32_t maxpass, int32_t ucap, const int32_t *__restrict__ tiles,
    const float *__restrict__ bias, float *__restrict__ Y, int64_t n_out, int32_t K, int32_t cin, int32_t cout, int32_t maxk_dbg) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int KG = 4 / NCO;
    constexpr int NTHR = 256;
    const int a_bytes = (ucap + 1) * 128 > 65536 ? (ucap + 1) * 128 : 65536;
    uint8_t *As = smem;
    uint16_t *slot_s = reinterpret_cast<uint16_t *>(smem + a_bytes);
    uint16_t *klist = slot_s + TP_KB * TP_TM;          // [TP_KB] (kk | live << 8), then nlive
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, kg = lane >> 5;
    const int g = wave / NCO, h = wave % NCO;
    const int64_t tile = blockIdx.x;
    int64_t row0 = tile * TP_TM;
    int rows = (int)(n_out - row0 < TP_TM ? n_out - row0 : TP_TM);
    int64_t wslot0 = 0;                                 // first weight slot of this tile's group
    if (tiles) { wslot0 = (int64_t)tiles[tile * 3] * K; row0 = tiles[tile * 3 + 1]; rows = tiles[tile * 3 + 2]; }
    const int nt0 = (blockIdx.y * NCO + h) * 2;         // this wave's first 32-channel output block
    const int nt_total = cout >> 5, ks_total = cin >> 4, nchunk = cin >> 6;
    const int gz = gridDim.z, zi = blockIdx.z;
    const int first = zi * KG + g, stride = gz * KG;    // this wave's share of the live offsets

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][n][e] = 0.f;
    if (tid < 8) reinterpret_cast<uint4 *>(As)[tid] = make_uint4(0u, 0u, 0u, 0u);      // the zero row

    const int np = npass[tile];
    const uint16_t *slots_t = slots + tile * (int64_t)K * TP_TM;
    const uint8_t *live_t = live + tile * (int64_t)K;
    const uint32_t cg[4] = {(uint32_t)kg, 2u + kg, 4u + kg, 6u + kg};        // channel granule of (ks, this lane's half)
    for (int p = 0; p < np; p++) {
        const int32_t *pt = pass_tab + (tile * (int64_t)maxpass + p) * 4;
        const int k0 = pt[0], k1 = pt[1], uoff = pt[2], ucnt = pt[3];
        for (int kb = k0; kb < k1; kb += TP_KB) {
            const int nk = k1 - kb < TP_KB ? k1 - kb : TP_KB;
            for (int c = 0; c < nchunk; c++) {
                __syncthreads();                        // every wave is done with the previous A tile / slot table
                // ---- stage the distinct rows of the pass, channels [64c, 64c+64): 8 granules of 16 B per row
                if (!(DBG & 8) && (nchunk > 1 || kb == k0)) {
                    const int ngran = ucnt * 8;
                    for (int i0 = 0; i0 < ngran; i0 += NTHR * 4) {
                        uint4 v[4];
                        int sl[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int i = i0 + j * NTHR + tid;
                            const int ic = i < ngran ? i : ngran - 1;
                            const int32_t grow = ulist[uoff + (ic >> 3)];
                            v[j] = *reinterpret_cast<const uint4 *>(X + ((int64_t)grow * cin + c * 64 + (ic & 7) * 8));
                            sl[j] = i < ngran ? i : -1;
                        }
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            if (sl[j] < 0) continue;
                            const int s = (sl[j] >> 3) + 1, gr = sl[j] & 7;
                            *reinterpret_cast<uint4 *>(As + s * 128 + ((gr ^ ((s >> 1) & 7)) << 4)) = v[j];
                        }
                    }
                }
                if (c == 0) {
                    // ---- slot table of offsets [kb, kb+nk): global [k][row] -> LDS [kk][r][m]
                    for (int i = tid; i < nk * (TP_TM / 8); i += NTHR) {           // 8 slots (16 B) per thread
                        const int kk = i >> 4, row8 = (i & 15) * 8;
                        const uint4 v = *reinterpret_cast<const uint4 *>(slots_t + (int64_t)(kb + kk) * TP_TM + row8);
                        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int row = row8 + j;
                            slot_s[kk * TP_TM + (row & 31) * 4 + (row >> 5)] = (uint16_t)(w4[j >> 1] >> ((j & 1) * 16));
                        }
                    }
                    if (wave == 0) {                    // live offsets of this block, compacted with one ballot
                        const int lv = lane < nk ? live_t[kb + lane] : 0;
                        const uint64_t bal = __ballot(lv != 0);
                        if (lv) klist[__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(lane | (lv << 8));
                        if (lane == 0) klist[TP_KB] = (uint16_t)__popcll(bal);
                    }
                }
                __syncthreads();
                int nlive = klist[TP_KB];
                if (DBG & 16) nlive = nlive < maxk_dbg ? nlive : maxk_dbg;       // dev aid: time per step = slope over the step count
                if (first >= nlive) continue;
                // weight fragments of (offset kk, 16-channel group ks, output block n): Wf[slot][nt][ks][lane][8]
                const uint16_t *wbase = Wf + ((wslot0 + kb) * nt_total + nt0) * (int64_t)ks_total * 512 + (int64_t)c * 4 * 512 + lane * 8;
                const int64_t wstride = (int64_t)nt_total * ks_total * 512;       // per offset
                const int64_t wn = (int64_t)ks_total * 512;                        // per 32-channel output block
                struct Rows { uint32_t base[4], sw[4]; };                          // LDS row address / swizzle of the 4 row blocks
                auto rows_of = [&](int kk) -> Rows {
                    const uint2 sv = *reinterpret_cast<const uint2 *>(slot_s + kk * TP_TM + r * 4);
                    uint32_t s4[4] = {sv.x & 0xffffu, sv.x >> 16, sv.y & 0xffffu, sv.y >> 16};
                    if (DBG & 2) { s4[0] = 1 + r; s4[1] = 33 + r; s4[2] = 65 + r; s4[3] = 97 + r; }      // conflict-free reads
                    Rows R;
#pragma unroll
                    for (int m = 0; m < 4; m++) { R.base[m] = s4[m] * 128u; R.sw[m] = (s4[m] >> 1) & 7u; }
                    return R;
                };
                auto read_a = [&](bf16x8 (&a)[4], const Rows &R, int ks) {
#pragma unroll
                    for (int m = 0; m < 4; m++)
                        a[m] = *reinterpret_cast<const bf16x8 *>(As + R.base[m] + ((cg[ks] ^ R.sw[m]) << 4));
                };
                auto load_b = [&](uint4 (&b)[2], int kk, int ks) {
                    const uint16_t *wk = wbase + kk * wstride + ks * 512;
                    b[0] = *reinterpret_cast<const uint4 *>(wk);
                    b[1] = *reinterpret_cast<const uint4 *>(wk + wn);
                };
                auto mma = [&](const bf16x8 (&a)[4], const uint4 (&b)[2]) {
                    const bf16x8 b0 = __builtin_bit_cast(bf16x8, b[0]), b1 = __builtin_bit_cast(bf16x8, b[1]);
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        if (DBG & 4) { acc[m][0][0] += (float)a[m][0] * (float)b0[0]; acc[m][1][0] += (float)a[m][0] * (float)b1[0]; }
                        else {
                            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b0, acc[m][0], 0, 0, 0);
                            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b1, acc[m][1], 0, 0, 0);
                        }
                    }
                };
                const int nstep = (nlive - first + stride - 1) / stride;          // offsets of this wave in this block
                auto kk_of = [&](int st) -> int { return klist[first + (st < nstep ? st : nstep - 1) * stride] & 0xff; };
                uint4 b[4][2];
                bf16x8 aA[4], aB[4];
                int kcur = kk_of(0);
#pragma unroll
                for (int ks = 0; ks < 4; ks++) load_b(b[ks], kcur, ks);
                Rows R = rows_of(kcur);
                read_a(aA, R, 0);
                for (int st = 0; st < nstep; st++) {
                    const int knext = kk_of(st + 1);                              // past the end: this offset again, unused
                    // unit 0
                    read_a(aB, R, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(aA, b[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG & 1)) load_b(b[0], knext, 0);
                    // unit 1
                    read_a(aA, R, 2);
                    const Rows Rn = rows_of(knext);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(aB, b[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG & 1)) load_b(b[1], knext, 1);
                    // unit 2
                    read_a(aB, R, 3);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(aA, b[2]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG & 1)) load_b(b[2], knext, 2);
                    // unit 3
                    read_a(aA, Rn, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mma(aB, b[3]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (!(DBG & 1)) load_b(b[3], knext, 3);
                    R = Rn;
                }
            }
        }
    }
    // ---- reduction over the offset groups of the workgroup (binary tree through LDS; the A tile is free now)
    float4 *xch = reinterpret_cast<float4 *>(smem);
#pragma unroll
    for (int sd = 1; sd < KG; sd *= 2) {
        __syncthreads();
        if ((g % (2 * sd)) == sd) {
            float4 *dst = xch + ((size_t)((g / (2 * sd)) * NCO + h) * 32) * 64 + lane;
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        dst[((m * 2 + n) * 4 + q) * 64] = make_float4(acc[m][n][q * 4], acc[m][n][q * 4 + 1], acc[m][n][q * 4 + 2], acc[m][n][q * 4 + 3]);
        }
        __syncthreads();
        if ((g % (2 * sd)) == 0 && g + sd < KG) {
            const float4 *src = xch + ((size_t)((g / (2 * sd)) * NCO + h) * 32) * 64 + lane;
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 v = src[((m * 2 + n) * 4 + q) * 64];
                        acc[m][n][q * 4] += v.x; acc[m][n][q * 4 + 1] += v.y; acc[m][n][q * 4 + 2] += v.z; acc[m][n][q * 4 + 3] += v.w;
                    }
        }
    }
    if (g != 0) return;
    // ---- epilogue: every output row of the tile is written once (or added, when the offsets are split over z)
#pragma unroll
    for (int n = 0; n < 2; n++) {
        const int col = (nt0 + n) * 32 + r;
        const float bv = bias && zi == 0 ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int lrow = m * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg;
                if (lrow >= rows) continue;
                float *dst = &Y[(row0 + lrow) * (int64_t)cout + col];
                if (gz == 1) *dst = acc[m][n][e] + bv;
                else unsafeAtomicAdd(dst, acc[m][n][e] + bv);
            }
    }
}

extern "C" int64_t cg3d_spconv_tile_lds_bytes(int32_t ucap) {
    const int64_t a = (int64_t)(ucap + 1) * 128;
    return (a > 65536 ? a : 65536) + TP_KB * TP_TM * 2 + (TP_KB + 2) * 2;       // the A tile doubles as the 64 KB reduction buffer
}

extern "C" int cg3d_spconv_tile_fwd(const uint16_t *X, const uint16_t *Wf, const uint16_t *slots, const uint8_t *live,
                                    const int32_t *pass_tab, const int32_t *npass, const int32_t *ulist,
                                    int32_t maxpass, int32_t ucap, const int32_t *tiles, int64_t ntile, const float *bias,
                                    float *Y, int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                                    int32_t ksplit, cg3d_stream_t stream) {
    if (n_out < 0 || n_in < 0 || K < 1 || cin < 64 || (cin & 63) || cout < 64 || (cout & 63) || (cout > 64 && (cout & 127)))
        return CG3D_ERR_ARG;
    if (ucap < TP_TM || ucap > 1023 || ksplit < 1 || ksplit > 8 || ((uintptr_t)X & 15) || ((uintptr_t)Wf & 15)) return CG3D_ERR_ARG;
    if (ntile == 0) return CG3D_OK;
    hipStream_t s = cg3d_hs(stream);
    if (ksplit > 1 && hipMemsetAsync(Y, 0, (size_t)n_out * cout * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    static const int ldspad = getenv("CG3D_TILE_LDSPAD") ? atoi(getenv("CG3D_TILE_LDSPAD")) : 0;   // dev aid: occupancy experiments
    const size_t lds = (size_t)cg3d_spconv_tile_lds_bytes(ucap) + (size_t)ldspad;
    static const int dbg = getenv("CG3D_TILE_DBG") ? atoi(getenv("CG3D_TILE_DBG")) : 0;      // dev aid: knock-out variants (wrong results)
#define TILE_LAUNCH(NW, DBG)                                                                                                   \
    do {                                                                                                                       \
        static bool attr = false;                                                                                              \
        if (!attr) {                                                                                                           \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spconv_tile<NW, DBG>),                                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)                     \
                return CG3D_ERR_LAUNCH;                                                                                        \
            attr = true;                                                                                                       \
            if (getenv("CG3D_TILE_INFO")) {                                                                                    \
                int nb = -1;                                                                                                   \
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spconv_tile<NW, DBG>, 256, lds);                 \
                fprintf(stderr, "k_spconv_tile<%d,%d>: lds %zu B, occupancy %d workgroups/CU\n", NW, DBG, lds, nb);            \
            }                                                                                                                  \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_spconv_tile<NW, DBG>), dim3((unsigned)ntile, (unsigned)(cout / (NW * 64)), (unsigned)ksplit),     \
                           dim3(256), lds, s, X, Wf, slots, live, pass_tab, npass, ulist, maxpass, ucap, tiles, bias, Y,       \
                           n_out, K, cin, cout, maxk);                                                                         \
    } while (0)
#define TILE_LAUNCH_NW(DBG)                                                                                                    \
    do { if (cout >= 128) TILE_LAUNCH(2, DBG); else TILE_LAUNCH(1, DBG); } while (0)
    const int maxk = getenv("CG3D_TILE_MAXK") ? atoi(getenv("CG3D_TILE_MAXK")) : 1 << 20;
    switch (dbg) {
    case 16: TILE_LAUNCH_NW(16); break;
    case 17: TILE_LAUNCH_NW(17); break;
    case 18: TILE_LAUNCH_NW(18); break;
    case 20: TILE_LAUNCH_NW(20); break;
    case 24: TILE_LAUNCH_NW(24); break;
    case 31: TILE_LAUNCH_NW(31); break;
    case 1: TILE_LAUNCH_NW(1); break;
    case 2: TILE_LAUNCH_NW(2); break;
    case 3: TILE_LAUNCH_NW(3); break;
    case 4: TILE_LAUNCH_NW(4); break;
    case 8: TILE_LAUNCH_NW(8); break;
    case 7: TILE_LAUNCH_NW(7); break;
    case 15: TILE_LAUNCH_NW(15); break;
    default: TILE_LAUNCH_NW(0); break;
    }
#undef TILE_LAUNCH_NW
#undef TILE_LAUNCH
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
