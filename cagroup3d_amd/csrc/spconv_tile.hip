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
                                                   int32_t *__restrict__ cursor) {
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
    const int r0c = ok0 ? lane : 0, r1c = ok1 ? lane + 64 : 0;
    auto fetch = [&](int k, int32_t &g0, int32_t &g1) {
        const int32_t *src = nbr + (int64_t)(k < K ? k : K - 1) * n_out + row0;
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
                                    cg3d_stream_t stream) {
    if (K < 1 || n_out < 0 || ntile < 0 || ucap < TP_TM || ucap > 1023 || maxpass < 1) return CG3D_ERR_ARG;
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
                       slots, live, pass_tab, npass, ulist, ulist_cap, cursor);
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

// ------------------------------------------------------------------------------------------------ convolution
// Persistent workgroup, one per CU: 4 consumer waves + 1 loader wave, two LDS stage buffers.
//
// A STAGE is (unit, pass, slot-table block of <= TP_KB offsets, 64-channel chunk); a unit is (tile of 128 output rows,
// 128-channel output block, offset share z).  The loader wave walks the stages of the workgroup's units (u = blockIdx.x,
// + gridDim.x, ...) one stage AHEAD of the consumers: it gathers the stage's distinct input rows from HBM into the
// other LDS buffer (coalesced 16-byte loads, 32 in flight per lane), re-lays the slot table, compacts the live offsets
// with a ballot and leaves a descriptor; one barrier per stage hands the buffer over.  Row gathers (HBM latency) and
// weight fragments (L2 latency) therefore sit in DIFFERENT waves' memory queues -- in one wave the in-order vmcnt made
// every weight fragment wait behind the outstanding row loads.
//
// Consumer wave (g, h): h = its 64-channel output block, g = its share of the stage's live offsets (KG = 4 / NCO offset
// groups).  Register tile 128 rows x 64 channels = 4 x 2 MFMA blocks (128 accumulators): every A fragment read from
// LDS feeds 2 MFMAs, every weight fragment streamed from L2 feeds 4 (the 128 x 32 tile of the first version read one A
// fragment per MFMA and ran into the LDS pipe at a quarter of the matrix peak).  Unit of the software
// pipeline = (offset, 16 channels) = 4 A fragments x 2 weight fragments -> 8 MFMAs on 8 different accumulators: while
// it is on the matrix pipe the A fragments of the next unit are on their way from LDS and each weight fragment is
// re-requested for the NEXT offset right after its last use (3-4 units ahead).  At the end of a unit the offset groups
// exchange halves of their partial sums through the LDS buffer just consumed (pairwise flags, no workgroup barrier:
// the loader keeps filling the other buffer) and every output row is stored once.
//
// Stage buffer: A tile (ucap + 1) x 128 B (row 0 = zeros; 16-byte granule g of row s sits at g ^ ((s >> 1) & 7): 16 lanes
// reading the same channel granule of 16 consecutive slots hit 16 different bank groups), >= 64 KB (it doubles as the
// exchange buffer); slot table [TP_KB][32][4] uint16 (lane r reads the slots of rows r, 32+r, 64+r, 96+r as ONE 8-byte
// word); list of live offsets; descriptor.
__device__ unsigned long long g_tile_steps[8 * 16];     // dev aid (DBG & 256): time stamp of every step of the first 8 stages, workgroup 0 wave 0
__device__ unsigned long long g_tile_dbg[8];        // dev aid (DBG & 128): cycles per consumer phase, summed over workgroups
extern "C" int cg3d_tile_debug_steps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_steps), sizeof(g_tile_steps)) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
extern "C" int cg3d_tile_debug_read(unsigned long long *out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_dbg), sizeof(g_tile_dbg)) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_dbg), z, sizeof(z)) != hipSuccess) return CG3D_ERR_LAUNCH; }
    return CG3D_OK;
}
struct StageDesc {
    int32_t valid, first, last, kb, c, rows, yb, zi;
    int64_t row0, wslot0;
    int32_t abuf, pad_[3];       // which of the two row tiles this stage reads (sizeof stays a multiple of 16)
};
#define TP_NLV 16            // row granules a loader thread stages per chunk: 4 loader waves x 64 lanes x 16 = 4096 = 512 rows x 8

template <int NCO, int DBG>
__global__ __launch_bounds__(512, 2) void k_spconv_tile(
    const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wf, const uint16_t *__restrict__ slots,
    const uint8_t *__restrict__ live, const int32_t *__restrict__ pass_tab, const int32_t *__restrict__ npass,
    const int32_t *__restrict__ ulist, int32_t maxpass, int32_t ucap, const int32_t *__restrict__ tiles,
    const float *__restrict__ bias, float *__restrict__ Y, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
    int32_t nunit, int32_t ny, int32_t gz, int32_t maxk_dbg, int32_t wrev, float *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int KG = 4 / NCO;
    const int a_bytes = (ucap + 1) * 128 > 65536 ? (ucap + 1) * 128 : 65536;
    // LDS: row tile 0 | row tile 1 | stage block 0 | stage block 1 | flags.  A stage block = slot table of <= 32 offsets,
    // the list of live ones, the stage descriptor; it alternates every stage.  The ROW tile alternates only when a stage
    // brings new rows: the slot-table blocks of one pass of a single-chunk layer (K > 32: the 5^3 / 9^3 class
    // convolutions) all read the rows staged once for the pass.
    constexpr int s_tab = TP_KB * TP_TM * 2 + (TP_KB + 2) * 2 + 12;                                          // slot table + live list
    constexpr int s_bytes = s_tab + (int)sizeof(StageDesc);                                                  // multiple of 16
    uint8_t *const sblk = smem + 2 * a_bytes;
    volatile int32_t *xflag = reinterpret_cast<volatile int32_t *>(sblk + 2 * s_bytes);                      // [5][4] data / ack flags
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nt_total = cout >> 5, ks_total = cin >> 4, nchunk = cin >> 6;
    const int G = gridDim.x;
    if (tid < 24) const_cast<int32_t *>(xflag)[tid] = 0;
    // per-channel sum / sum of squares of the rows this workgroup stores (BatchNorm statistics of the layer's output,
    // accumulated by the loader waves while they drain a tile): [2][cout] floats behind the flags
    float *sacc = reinterpret_cast<float *>(sblk + 2 * s_bytes + 96);
    if (stats)
        for (int i = tid; i < 2 * cout; i += 512) sacc[i] = 0.f;
    __syncthreads();

    if (wave >= 4) {
        // =========================================================================================== loader waves
        // Software pipeline of the loaders: the pass descriptor (npass / pass_tab) is requested one PASS ahead, the row
        // indices (ulist) one STAGE ahead of the pass that needs them, so a stage costs one memory latency (its row
        // gathers, all in flight at once: 16 granules per thread) plus the LDS writes -- not a chain of three.
        const int lw = wave - 4, lt = lw * 64 + lane;     // 256 loader threads
        struct PassRec { int valid, np, k0, k1, uoff, ucnt; };
        // XCD-aware unit order: workgroup b runs on XCD b % 8 (own L2); position pos = b + i * gridDim of the round-robin
        // maps to unit  first unit of XCD (pos % 8) + pos / 8,  so an XCD walks ONE contiguous range of tiles and the
        // halo rows two neighbouring tiles share are found in its L2 (tile t, t+1 on different XCDs: both fetch them)
        const int u_lo = nunit >> 3, u_rem = nunit & 7;
        auto unit_of = [&](int pos) -> int {
            if (pos >= nunit) return nunit;
            const int x = pos & 7;
            return x * u_lo + (x < u_rem ? x : u_rem) + (pos >> 3);
        };
        auto load_pass = [&](int u, int p) -> PassRec {
            PassRec P = {0, 0, 0, 0, 0, 0};
            if (u < nunit) {
                const int64_t t = u / (ny * gz);
                const int32_t *pt = pass_tab + (t * maxpass + p) * 4;
                P.valid = 1; P.np = npass[t]; P.k0 = pt[0]; P.k1 = pt[1]; P.uoff = pt[2]; P.ucnt = pt[3];
            }
            return P;
        };
        int32_t idx[TP_NLV], idx_next[TP_NLV];
        auto issue_idx = [&](int32_t (&dst)[TP_NLV], int uoff, int ucnt) {
            // all TP_NLV requests unconditionally, on clamped positions (the surplus ones hit one cached address): a guard
            // per request becomes a branch per request, the requests serialise and the array they land in moves to scratch
            const int ngran = ucnt * 8;
            if (ngran > 0) {
#pragma unroll
                for (int j = 0; j < TP_NLV; j++) {
                    const int i = j * 256 + lt;
                    dst[j] = ulist[uoff + ((i < ngran ? i : ngran - 1) >> 3)];
                }
            }
        };
        // the output tile a unit's last stage left in its buffer (row-major per consumer wave) -> global memory
        // (two named records selected with ternaries, NOT an array indexed by the buffer number: a dynamically indexed local
        // array lives in scratch memory -- 16 KB of stores per stage and workgroup, 2.4 x the output bytes on the memory side)
        struct Hist { int32_t valid, last, yb, zi, rows; int64_t row0; };
        Hist h0 = {0, 0, 0, 0, 0, 0}, h1 = {0, 0, 0, 0, 0, 0};
        volatile int32_t *ldrain = xflag + 20;            // drains completed, summed over the loader waves
        int dseq = 0;
        auto drain = [&](int bufi) {
            Hist H;
            H.valid = bufi ? h1.valid : h0.valid; H.last = bufi ? h1.last : h0.last; H.yb = bufi ? h1.yb : h0.yb;
            H.zi = bufi ? h1.zi : h0.zi; H.rows = bufi ? h1.rows : h0.rows; H.row0 = bufi ? h1.row0 : h0.row0;
            if (!H.valid || !H.last || (DBG & 32)) return;
            constexpr int rows_per = TP_TM / KG;
            const int cw = lw, cg_ = cw / NCO, ch = cw % NCO;            // loader wave lw drains consumer wave lw
            const float *tb = reinterpret_cast<const float *>(smem + bufi * a_bytes) + (size_t)cw * 4096;
            const int c4 = (lane & 15) * 4, rq = lane >> 4;
            const int col0 = (H.yb * NCO + ch) * 64 + c4;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias && H.zi == 0) bv = *reinterpret_cast<const float4 *>(bias + col0);
            float *ybase = Y + (H.row0 + cg_ * rows_per + rq) * (int64_t)cout + col0;
            float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
#pragma unroll
            for (int i = 0; i < rows_per / 4; i++) {
                float4 v = *reinterpret_cast<const float4 *>(tb + (i * 4 + rq) * 64 + c4);
                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                if (cg_ * rows_per + i * 4 + rq < H.rows) {
                    float *dst = ybase + (int64_t)i * 4 * cout;
                    if (gz == 1) *reinterpret_cast<float4 *>(dst) = v;
                    else { unsafeAtomicAdd(dst, v.x); unsafeAtomicAdd(dst + 1, v.y); unsafeAtomicAdd(dst + 2, v.z); unsafeAtomicAdd(dst + 3, v.w); }
                    t0.x += v.x; t0.y += v.y; t0.z += v.z; t0.w += v.w;
                    t1.x += v.x * v.x; t1.y += v.y * v.y; t1.z += v.z * v.z; t1.w += v.w * v.w;
                }
            }
            if (stats) {                                  // 8 lanes x 2 waves share a column quad: LDS atomics
                float *a0 = sacc + col0, *a1 = sacc + cout + col0;
                unsafeAtomicAdd(a0, t0.x); unsafeAtomicAdd(a0 + 1, t0.y); unsafeAtomicAdd(a0 + 2, t0.z); unsafeAtomicAdd(a0 + 3, t0.w);
                unsafeAtomicAdd(a1, t1.x); unsafeAtomicAdd(a1 + 1, t1.y); unsafeAtomicAdd(a1 + 2, t1.z); unsafeAtomicAdd(a1 + 3, t1.w);
            }
            if (bufi) h1.valid = 0; else h0.valid = 0;
            // The four loader waves drain disjoint quarters of the buffer, but each one's refill requests are spread over
            // ALL of it: nobody may refill before everybody has read.  (Uniform: every loader wave drains the same tiles.)
            dseq++;
            __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): this wave's LDS reads have returned
            if (lane == 0) __hip_atomic_fetch_add(const_cast<int32_t *>(ldrain), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            while (*ldrain < 4 * dseq) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        int sidx = 0, abuf = 1;
        int pos = blockIdx.x;
        PassRec cur = load_pass(unit_of(pos), 0);
        if (cur.valid) issue_idx(idx, cur.uoff, cur.ucnt);
        int u = unit_of(pos), p = 0;
        while (cur.valid) {
            const int64_t tile = u / (ny * gz);
            const bool last_pass = p == cur.np - 1;
            const int un = last_pass ? unit_of(pos + G) : u, pn = last_pass ? 0 : p + 1;
            const PassRec nxt = load_pass(un, pn);       // requested now, needed at this pass's last stage
            StageDesc D;
            D.valid = 1;
            D.yb = (u / gz) % ny;
            D.zi = u % gz;
            D.row0 = tile * TP_TM;
            D.rows = (int)(n_out - D.row0 < TP_TM ? n_out - D.row0 : TP_TM);
            D.wslot0 = 0;
            if (tiles) { D.wslot0 = (int64_t)tiles[tile * 3] * K; D.row0 = tiles[tile * 3 + 1]; D.rows = tiles[tile * 3 + 2]; }
            const int ngran_pass = cur.ucnt * 8;
            for (int kb = cur.k0; kb < cur.k1 || kb == cur.k0; kb += TP_KB) {      // (an empty pass still is one stage)
                const int nk = cur.k1 - kb < TP_KB ? (cur.k1 - kb > 0 ? cur.k1 - kb : 0) : TP_KB;
                for (int c = 0; c < nchunk; c++) {
                    const bool last_stage = kb + TP_KB >= cur.k1 && c == nchunk - 1;
                    D.first = (p == 0 && kb == cur.k0 && c == 0) ? 1 : 0;
                    D.last = (last_pass && last_stage) ? 1 : 0;
                    D.kb = kb;
                    D.c = c;
                    const bool stage_rows = nchunk > 1 || kb == cur.k0;       // else: the rows of this pass are already there
                    const int ngran = stage_rows ? ngran_pass : 0;            // (no rows: every guard below is false)
                    if (stage_rows) {
                        abuf ^= 1;
                        drain(abuf);                    // the output tile an earlier unit's last stage left in this row tile
                    }
                    D.abuf = abuf;
                    uint8_t *As = smem + abuf * a_bytes;
                    uint8_t *Ss = sblk + (sidx & 1) * s_bytes;
                    uint16_t *slot_s = reinterpret_cast<uint16_t *>(Ss);
                    uint16_t *klist = slot_s + TP_KB * TP_TM;
                    {
                        const Hist hn = {D.valid, D.last, D.yb, D.zi, D.rows, D.row0};
                        if (abuf) h1 = hn; else h0 = hn;
                    }
                    // ---- requests: slot table (wave 4), the stage's rows, the next pass's row indices
                    uint4 sv[8];
                    int lv = 0;
                    if (lw == 0 && c == 0 && !((DBG & 64) && sidx > 1)) {
                        const uint16_t *slots_t = slots + (tile * K + kb) * (int64_t)TP_TM;
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int i = j * 64 + lane;
                            if (i < nk * (TP_TM / 8)) sv[j] = *reinterpret_cast<const uint4 *>(slots_t + (int64_t)(i >> 4) * TP_TM + (i & 15) * 8);
                        }
                        lv = lane < nk ? live[tile * (int64_t)K + kb + lane] : 0;
                    }
                    // rows: global memory -> LDS row tile by LDS-DMA (global_load_lds_dwordx4: no data registers, nothing to
                    // write back; a request fills wave-uniform base + lane * 16).  Lane i of request j owns PHYSICAL granule
                    // i & 7 of row slot (i >> 3) + 1, so it fetches the logical granule the swizzle maps there.  (The
                    // register-staged version kept its 16 uint4 per lane in scratch memory: hipcc did not promote the array
                    // next to the 256-register consumer path -- one request in flight per lane, and scratch write-backs worth
                    // 2.4 x the output bytes on the memory side.)
                    if (!(DBG & 8) && ngran > 0) {
#pragma unroll
                        for (int j = 0; j < TP_NLV; j++) {
                            if (j * 256 < ngran) {                               // uniform
                                const int i = j * 256 + lt;
                                const int sl = (i >> 3) + 1, gr = (i & 7) ^ ((sl >> 1) & 7);
                                const uint16_t *src = X + ((int64_t)idx[j] * cin + c * 64 + gr * 8);
                                uint8_t *dst = As + (size_t)(j * 256 + lw * 64 + 8) * 16;
                                if (i < ngran)
                                    __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)src,
                                                                     (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                            }
                        }
                    }
                    if (last_stage && nxt.valid) issue_idx(idx_next, nxt.uoff, nxt.ucnt);
                    // ---- LDS: descriptor, zero row, slot table / list of live offsets, rows
                    if (lt == 0) *reinterpret_cast<StageDesc *>(Ss + s_tab) = D;
                    if (lt < 8 && stage_rows) reinterpret_cast<uint4 *>(As)[lt] = make_uint4(0u, 0u, 0u, 0u);      // the zero row
                    if (lw == 0 && c == 0 && !((DBG & 64) && sidx > 1)) {
                        // slot table of offsets [kb, kb+nk): global [k][row] -> LDS [kk][r][m]; live offsets compacted with one ballot
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int i = j * 64 + lane;
                            if (i < nk * (TP_TM / 8)) {
                                const int kk = i >> 4, row8 = (i & 15) * 8;
                                const uint32_t w4[4] = {sv[j].x, sv[j].y, sv[j].z, sv[j].w};
#pragma unroll
                                for (int q = 0; q < 8; q++) {
                                    const int row = row8 + q;
                                    slot_s[kk * TP_TM + (row & 31) * 4 + (row >> 5)] = (uint16_t)(w4[q >> 1] >> ((q & 1) * 16));
                                }
                            }
                        }
                        const uint64_t bal = __ballot(lv != 0);
                        if (lv) klist[__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(lane | (lv << 8));
                        if (lane == 0) klist[TP_KB] = (uint16_t)__popcll(bal);
                    } else if (lw == 1 && (c > 0 || ((DBG & 64) && sidx > 1))) {
                        // same block of offsets as the previous stage (the other buffer): copy its slot table and list
                        const uint4 *src = reinterpret_cast<const uint4 *>(sblk + ((sidx + 1) & 1) * s_bytes);
                        uint4 *dst = reinterpret_cast<uint4 *>(Ss);
                        for (int i = lane; i < s_tab / 16; i += 64) dst[i] = src[i];
                    }
// (the rows were written by the LDS-DMA requests above; __syncthreads() below waits for them: vmcnt(0))
                    __syncthreads();                    // hand the buffer over; the consumers start on it
                    sidx++;
                }
            }
            cur = nxt;
#pragma unroll
            for (int j = 0; j < TP_NLV; j++) idx[j] = idx_next[j];
            u = un;
            if (last_pass) pos += G;
            p = pn;
        }
        // end marker
        drain(abuf ^ 1);
        if (lt == 0) {
            StageDesc E;
            E.valid = 0; E.first = E.last = 0; E.kb = E.c = E.rows = E.yb = E.zi = 0; E.row0 = E.wslot0 = 0; E.abuf = 0;
            *reinterpret_cast<StageDesc *>(sblk + (sidx & 1) * s_bytes + s_tab) = E;
        }
        __syncthreads();
        drain(abuf);                                    // the last unit's tile (its closing counter wait = all four loader waves are done)
        if (stats)
            for (int i = lt; i < 2 * cout; i += 256) stats[(int64_t)blockIdx.x * 2 * cout + i] = sacc[i];
        return;
    }

    // =============================================================================================== consumer waves
    const int r = lane & 31, kg = lane >> 5;
    const int g = wave / NCO, h = wave % NCO;
    const uint32_t cg[4] = {(uint32_t)kg, 2u + kg, 4u + kg, 6u + kg};        // channel granule of (ks, this lane's half)
    f32x16 acc[4][2];
    int seq = 0;                                        // units finished by this workgroup (exchange flag values)
    unsigned long long tph[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = (DBG & 128) ? __builtin_readcyclecounter() : 0;
    auto stamp = [&](int ph) {
        if (DBG & 128) { const unsigned long long t = __builtin_readcyclecounter(); tph[ph] += t - tprev; tprev = t; }
    };
    for (int sidx = 0;; sidx++) {
        __syncthreads();                                // the loader has filled buffer sidx & 1
        stamp(0);
        const uint8_t *Ss = sblk + (sidx & 1) * s_bytes;
        const uint16_t *slot_s = reinterpret_cast<const uint16_t *>(Ss);
        const uint16_t *klist = slot_s + TP_KB * TP_TM;
        const StageDesc D = *reinterpret_cast<const StageDesc *>(Ss + s_tab);
        if (!D.valid) break;
        uint8_t *As = smem + D.abuf * a_bytes;
        stamp(1);
        if (D.first) {
#pragma unroll
            for (int m = 0; m < 4; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int e = 0; e < 16; e++) acc[m][n][e] = 0.f;
        }
        const int nt0 = (D.yb * NCO + h) * 2;           // this wave's first 32-channel output block
        const int first = D.zi * KG + g, stride = gz * KG;   // this wave's share of the live offsets
        int nlive = klist[TP_KB];
        if (DBG & 16) nlive = nlive < maxk_dbg ? nlive : maxk_dbg;           // dev aid: time per step = slope over the step count
        const int nstep = first < nlive ? (nlive - first + stride - 1) / stride : 0;     // offsets of this wave
        if (nstep > 0) {
            // weight fragments of (offset kk, 16-channel group ks, output block n): Wf[slot][nt][ks][lane][8]
            // (wrev: offset k reads weight slot K-1-k -- the data gradient of a map onto itself walks the FORWARD plan)
            const uint16_t *wbase = Wf + ((D.wslot0 + (wrev ? K - 1 - D.kb : D.kb)) * nt_total + nt0) * (int64_t)ks_total * 512 + (int64_t)D.c * 4 * 512 + lane * 8;
            const int64_t wstride = (wrev ? -1 : 1) * (int64_t)nt_total * ks_total * 512;       // per offset
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
            auto kk_of = [&](int st) -> int { return klist[first + (st < nstep ? st : nstep - 1) * stride] & 0xff; };
            uint4 b[4][2];
            bf16x8 aA[4], aB[4];
            const int kcur = kk_of(0);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) load_b(b[ks], kcur, ks);
            Rows R = rows_of(kcur);
            read_a(aA, R, 0);
            if (DBG & 128) { __builtin_amdgcn_s_waitcnt(0); stamp(6); }      // prologue: first weights / slots / A fragments have arrived
            // One scheduling region per unit: its 8 MFMAs, the 4 LDS reads of the NEXT unit's A fragments and the 2
            // weight-fragment loads of the next offset, interleaved by rule (one wave per SIMD computes: whatever is not
            // issued between two MFMAs of the block is not overlapped with the matrix pipe).  Three fragment sets (requests
            // two units ahead) were tried: 13-20 spilled registers and no faster (111 vs 98 us on the 128 -> 128 layer).
#define TILE_UNIT_SCHED()                                                                       \
    do {                                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) {                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  /* MFMA */                      \
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  /* VALU (address) */            \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  /* DS read */                   \
        }                                                                                       \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) {                                      \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                  \
            __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);  /* VALU / SALU */               \
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  /* VMEM read */                 \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
            for (int st = 0; st < nstep; st++) {
                if ((DBG & 256) && blockIdx.x == 0 && tid == 0 && sidx < 8 && st < 15) g_tile_steps[sidx * 16 + st] = __builtin_readcyclecounter();
                const int knext = kk_of(st + 1);                              // past the end: this offset again, unused
                // unit 0
                read_a(aB, R, 1);
                mma(aA, b[0]);
                if (!(DBG & 1)) load_b(b[0], knext, 0);
                TILE_UNIT_SCHED();
                // unit 1
                read_a(aA, R, 2);
                const Rows Rn = rows_of(knext);
                mma(aB, b[1]);
                if (!(DBG & 1)) load_b(b[1], knext, 1);
                TILE_UNIT_SCHED();
                // unit 2
                read_a(aB, R, 3);
                mma(aA, b[2]);
                if (!(DBG & 1)) load_b(b[2], knext, 2);
                TILE_UNIT_SCHED();
                // unit 3
                read_a(aA, Rn, 0);
                mma(aB, b[3]);
                if (!(DBG & 1)) load_b(b[3], knext, 3);
                TILE_UNIT_SCHED();
                R = Rn;
            }
#undef TILE_UNIT_SCHED
        }
        if ((DBG & 256) && blockIdx.x == 0 && tid == 0 && sidx < 8) g_tile_steps[sidx * 16 + 15] = __builtin_readcyclecounter();
        stamp(2);
        if (!D.last || (DBG & 32)) continue;             // DBG 32: no exchange / stores

        // ---- end of the unit: the KG waves holding partial sums of the same 128 x 64 block exchange halves through
        // the buffer just consumed (monotonic per-wave flags, no workgroup barrier: the loader is busy with the other
        // buffer), level by level; wave g ends up owning 4 / KG of the 4 row blocks and stores them.
        seq++;
        float4 *xch = reinterpret_cast<float4 *>(As);
        volatile int32_t *f_done = xflag, *f_d0 = xflag + 4, *f_a0 = xflag + 8, *f_d1 = xflag + 12, *f_a1 = xflag + 16;
        (void)f_a1;
        // every consumer wave must have left the compute loop of this stage before the A tile is overwritten
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) f_done[wave] = seq;
        for (int w = 0; w < 4; w++)
            while (f_done[w] < seq) __builtin_amdgcn_s_sleep(1);
        stamp(3);
        auto level = [&](auto GIVE, auto KEEP, auto HALF, volatile int32_t *f_data, int partner) {
            constexpr int give = decltype(GIVE)::value, keep = decltype(KEEP)::value, half = decltype(HALF)::value;
            float4 *dst = xch + (size_t)wave * 16 * 64 + lane;               // 16 KB per wave
#pragma unroll
            for (int m = 0; m < half; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        dst[((m * 2 + n) * 4 + q) * 64] = make_float4(acc[give + m][n][q * 4], acc[give + m][n][q * 4 + 1],
                                                                      acc[give + m][n][q * 4 + 2], acc[give + m][n][q * 4 + 3]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) f_data[wave] = seq;
            while (f_data[partner] < seq) __builtin_amdgcn_s_sleep(1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float4 *src = xch + (size_t)partner * 16 * 64 + lane;
#pragma unroll
            for (int m = 0; m < half; m++)
#pragma unroll
                for (int n = 0; n < 2; n++) {
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float4 v = src[((m * 2 + n) * 4 + q) * 64];
                        acc[keep + m][n][q * 4] += v.x; acc[keep + m][n][q * 4 + 1] += v.y;
                        acc[keep + m][n][q * 4 + 2] += v.z; acc[keep + m][n][q * 4 + 3] += v.w;
                    }
                    __builtin_amdgcn_sched_barrier(0);                      // at most 16 registers of partner data in flight
                }
        };
        // Epilogue.  A dword store per accumulator register is store-ISSUE bound (128 store instructions per wave: ~13 us
        // per tile measured, the largest fixed cost of the first versions).  The wave leaves its final 32*cnt rows x 64
        // channels ROW-MAJOR in its own 16 KB exchange region (the partner has acknowledged reading it) and goes on to the
        // next stage; the loader waves store the tile 16 bytes per lane -- 4 rows x 256 contiguous bytes per instruction --
        // before they refill this buffer.
        auto store = [&](auto LO, auto CNT, volatile int32_t *f_ack, int partner) {
            constexpr int lo = decltype(LO)::value, cnt = decltype(CNT)::value;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) f_ack[wave] = seq;            // I have read the partner's region
            while (f_ack[partner] < seq) __builtin_amdgcn_s_sleep(1);
            float *tb = reinterpret_cast<float *>(xch) + (size_t)wave * 4096;
#pragma unroll
            for (int m = 0; m < cnt; m++)
#pragma unroll
                for (int n = 0; n < 2; n++)
#pragma unroll
                    for (int e = 0; e < 16; e++)
                        tb[(m * 32 + (e & 3) + 8 * (e >> 2) + 4 * kg) * 64 + n * 32 + r] = acc[lo + m][n][e];
        };
        using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
        if constexpr (KG == 2) {
            const int partner = (g ^ 1) * NCO + h;
            if (g == 0) { level(I2{}, I0{}, I2{}, f_d0, partner); stamp(4); store(I0{}, I2{}, f_a0, partner); }
            else        { level(I0{}, I2{}, I2{}, f_d0, partner); stamp(4); store(I2{}, I2{}, f_a0, partner); }
            stamp(5);
        } else {
            // KG == 4 (NCO == 1): level 0 between g and g ^ 2 (halves), level 1 between g and g ^ 1 (quarters)
            const int p0 = (g ^ 2), p1 = (g ^ 1);
            if ((g & 2) == 0) level(I2{}, I0{}, I2{}, f_d0, p0); else level(I0{}, I2{}, I2{}, f_d0, p0);
            // my level-0 region is reused at level 1: the level-0 partner must have read it
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) f_a0[wave] = seq;
            while (f_a0[p0] < seq) __builtin_amdgcn_s_sleep(1);
            if ((g & 2) == 0) {
                if ((g & 1) == 0) { level(I1{}, I0{}, I1{}, f_d1, p1); store(I0{}, I1{}, f_a1, p1); }
                else              { level(I0{}, I1{}, I1{}, f_d1, p1); store(I1{}, I1{}, f_a1, p1); }
            } else {
                if ((g & 1) == 0) { level(I3{}, I2{}, I1{}, f_d1, p1); store(I2{}, I1{}, f_a1, p1); }
                else              { level(I2{}, I3{}, I1{}, f_d1, p1); store(I3{}, I1{}, f_a1, p1); }
            }
        }
    }
    if ((DBG & 128) && wave == 0 && lane == 0) {
        for (int i = 0; i < 7; i++) atomicAdd(&g_tile_dbg[i], tph[i]);
        atomicAdd(&g_tile_dbg[7], 1ull);
    }
}

int64_t cg3d_tile_v1_lds_bytes(int32_t ucap) {
    const int64_t a = (int64_t)(ucap + 1) * 128;
    const int64_t buf = (a > 65536 ? a : 65536) + TP_KB * TP_TM * 2 + (TP_KB + 2) * 2 + 12 + (int64_t)sizeof(StageDesc);
    return 2 * buf + 96 + 4096;     // two stage buffers (the A tile doubles as the 64 KB exchange buffer) + flags + BN partial sums [2][<=512]
}

static int tile_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        ncu = prop.multiProcessorCount;
    }
    return ncu;
}
// number of (persistent) workgroups cg3d_spconv_tile_fwd launches = rows of its `stats` output
int32_t cg3d_tile_v1_grid(int64_t ntile, int32_t cout, int32_t ksplit) {
    const int ncu = tile_ncu();
    if (ncu <= 0 || ntile < 0 || cout < 64 || ksplit < 1) return -1;
    const int64_t nunit = ntile * (cout >= 128 ? cout / 128 : 1) * ksplit;
    return (int32_t)(nunit < ncu ? nunit : ncu);
}

int cg3d_tile_v1_fwd(const uint16_t *X, const uint16_t *Wf, const uint16_t *slots, const uint8_t *live,
                                    const int32_t *pass_tab, const int32_t *npass, const int32_t *ulist,
                                    int32_t maxpass, int32_t ucap, const int32_t *tiles, int64_t ntile, const float *bias,
                                    float *Y, int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                                    int32_t ksplit, int32_t wrev, float *stats, cg3d_stream_t stream) {
    if (stats && (ksplit != 1 || tiles || cout > 512)) return CG3D_ERR_ARG;
    if (n_out < 0 || n_in < 0 || K < 1 || cin < 64 || (cin & 63) || cout < 64 || (cout & 63) || (cout > 64 && (cout & 127)))
        return CG3D_ERR_ARG;
    if (ucap < TP_TM || ucap > 511 || ksplit < 1 || ksplit > 8 || ((uintptr_t)X & 15) || ((uintptr_t)Wf & 15)) return CG3D_ERR_ARG;
    if (ntile == 0) return CG3D_OK;
    hipStream_t s = cg3d_hs(stream);
    if (ksplit > 1 && hipMemsetAsync(Y, 0, (size_t)n_out * cout * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    static const int ldspad = getenv("CG3D_TILE_LDSPAD") ? atoi(getenv("CG3D_TILE_LDSPAD")) : 0;   // dev aid: occupancy experiments
    const size_t lds = (size_t)cg3d_tile_v1_lds_bytes(ucap) + (size_t)ldspad;
    const int ncu = tile_ncu();
    if (ncu <= 0) return CG3D_ERR_LAUNCH;
    // persistent workgroups, one per CU (two 64 KB stage buffers fill the LDS)
    const int32_t ny = cout >= 128 ? cout / 128 : 1;
    const int64_t nunit = ntile * ny * ksplit;
    if (nunit > 0x7fffffffll) return CG3D_ERR_ARG;
    const int64_t grid = nunit < ncu ? nunit : ncu;
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
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spconv_tile<NW, DBG>, 512, lds);                     \
                fprintf(stderr, "k_spconv_tile<%d,%d>: lds %zu B, occupancy %d workgroups/CU, %d CUs\n", NW, DBG, lds, nb, ncu); \
            }                                                                                                                  \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_spconv_tile<NW, DBG>), dim3((unsigned)grid), dim3(512), lds, s, X, Wf, slots, live, pass_tab,    \
                           npass, ulist, maxpass, ucap, tiles, bias, Y, n_out, K, cin, cout, (int32_t)nunit, ny, ksplit, maxk, wrev ? 1 : 0, stats); \
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
    case 128: TILE_LAUNCH_NW(128); break;
    case 256: TILE_LAUNCH_NW(256); break;
    case 257: TILE_LAUNCH_NW(257); break;
    case 258: TILE_LAUNCH_NW(258); break;
    case 259: TILE_LAUNCH_NW(259); break;
    case 260: TILE_LAUNCH_NW(260); break;
    case 264: TILE_LAUNCH_NW(264); break;
    case 129: TILE_LAUNCH_NW(129); break;
    case 130: TILE_LAUNCH_NW(130); break;
    case 136: TILE_LAUNCH_NW(136); break;
    case 56: TILE_LAUNCH_NW(56); break;
    case 88: TILE_LAUNCH_NW(88); break;
    case 120: TILE_LAUNCH_NW(120); break;
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
