// spconv_tile2.hip -- sparse convolution forward / data gradient on LDS-staged neighbour tiles, round 3 (gfx950).
//
// The convolution on the plans / operand layouts of spconv_tile.hip (cg3d_tile_plan_build, cg3d_spconv_prep_weights_frag);
// replaces round 2's persistent loader/consumer kernel (one workgroup per CU).  Call sites: pcdet/models/backbones_3d/biresnet.py:358-406 (every
// K > 1 convolution of BiResNet), dense_heads/cagroup_head.py:259-275 (the grouped class-branch convolutions);
// MinkowskiEngine's ConvolutionForwardGPU / ConvolutionBackwardGPU (un-vendored).
//
// Why a second kernel.  The round-2 kernel ran ONE persistent workgroup per CU (4 loader + 4 consumer waves, two 64 KB
// row tiles): one MFMA wave per SIMD, so every latency of that wave's chain -- weight fragments from L2, slot-indexed LDS
// reads, the stage hand-over barrier, the prologue of each stage, the exchange of partial sums at the end of a unit --
// was exposed: 0.79 us per (offset, 64 channels) step against 0.43 of MFMA issue, plus 12 us of fixed cost per 21 us
// unit (profiles/r03_tile_v1_knockout.txt: with the weight loads, the bank conflicts AND the MFMAs knocked out the 128 ->
// 128 layer still took 50 of its 82 us).  Here a workgroup is 4 waves that all do everything in turn -- stage the rows
// by LDS-DMA, multiply, exchange, store -- with ONE 64 KB row tile, so TWO independent workgroups live on a CU (2 x 73 KB
// of LDS, 2 waves per SIMD at <= 256 registers): while one waits for its gather, sits in a barrier or stores its tile,
// the other one's MFMAs own the matrix pipe, and inside the compute phase the two waves of a SIMD fill each other's
// LDS / L2 stalls.  Units (tile of 128 rows x 128 output channels x offset share) are plain workgroups, dispatched by
// the hardware as slots free up (no static assignment, no persistent tail beyond the last units).
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "cg3d_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define T2_TM 128            // output rows per tile
#define T2_KB 32             // offsets whose slot table is resident in LDS at a time (template parameter KBT of the unit: 32, or
#define T2_KB_BIG 128        // 128 for kernels of more than CG3D_TILE_KB_BIG_MIN_K offsets -- the 5^3 / 9^3 class convolutions)
#define T2_NLV 16            // row granules a thread stages per pass and chunk: 256 threads x 16 = 4096 = 512 rows x 8
#ifndef T2_STAT_SLOTS
#define T2_STAT_SLOTS CG3D_BN_SLOTS      // (dev: more slots than the table's consumers add up, to measure the atomics' same-address queueing)
#endif
#ifndef T2_DBG
#define T2_DBG 0             // dev knock-outs (tools/mb_tile_dbg.sh): 1 no weight loads in the loop, 2 conflict-free A reads, 4 no MFMA, 8 no row staging, 16 no exchange / store, 64 no global stores (exchange + row-major pass kept), 128 no exchange (rows + stores kept)
#endif

#ifdef CG3D_TILE_TRACE
// dev build only (CG3D_HIPCC_EXTRA=-DCG3D_TILE_TRACE): per workgroup {shader-clock start, end, 100 MHz start, end, HW_ID | XCC_ID << 32, unit}
__device__ unsigned long long *g_t2_trace;
extern "C" int cg3d_tile2_trace_set(unsigned long long *buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_t2_trace), &buf, sizeof(buf)) == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
#endif

// LDS: row tile (ucap + 1) x 128 B, >= 64 KB (it doubles as the exchange / output buffer) | slot table [T2_KB][32][4] uint16
// | list of live offsets | BatchNorm partial sums [2][128]
__host__ __device__ static inline int t2_a_bytes(int ucap) { return (ucap + 1) * 128 > 65536 ? (ucap + 1) * 128 : 65536; }
// row tile + slot table [KB][32][4] uint16 (see t2_unit)
__host__ __device__ static inline int t2_main_bytes(int ucap, int kb) {
    if (kb <= 32) return t2_a_bytes(ucap) + kb * 128 * 2;
    const int m = (ucap + 1) * 128 + kb * 128 * 2;
    return m > 65536 ? m : 65536;
}
#define T2_IDX_BYTES (512 * 4)                        // row indices of the current pass

// Workgroup = one unit = (tile of 128 output rows, block of 128 (NCO == 2) or 64 (NCO == 1) output channels, offset share
// zi of gz).  Wave (g, h): h = its 64-channel output block, g = its share of the live offsets (KG = 4 / NCO shares).
// Register tile of a wave: 128 rows x 64 channels = 4 x 2 MFMA blocks (128 accumulators) kept over all passes, offset
// blocks and input-channel chunks of the unit; at the end the KG waves holding partial sums of one block exchange halves
// through the row tile (free by then), leave their rows row-major in LDS and store them 16 bytes per lane.
// bx: index of the workgroup among the workgroups of ITS unit kind; tile_base: first tile of that kind (a launch may mix
// units of both kinds: k_spconv_tile2_mix)
template <int NCO, int KBT>
__device__ __forceinline__ void t2_unit(
    const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wf, const uint16_t *__restrict__ slots,
    const uint8_t *__restrict__ live, const int32_t *__restrict__ pass_tab, const int32_t *__restrict__ npass,
    const int32_t *__restrict__ ulist, int32_t maxpass, int32_t ucap, const int32_t *__restrict__ tiles,
    const int32_t *__restrict__ order, const float *__restrict__ bias, float *__restrict__ Y, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
    int32_t nunit, int32_t ny, int32_t gz, int32_t wrev, float *__restrict__ stats, int32_t stagger, const unsigned bx, const int32_t tile_base) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int KG = 4 / NCO;
    constexpr int NC = NCO * 64;                        // output channels of the unit
    // LDS layout (t2_main_bytes): [row tile | slot table] | klist | idx_s | sacc.  Blocks of 32 offsets: the row tile is at least the
    // 64 KB the exchange needs.  Blocks of 128: the exchange (after the last multiply: the table is dead by then) runs over
    // row tile AND table, so a plan of <= 255 rows per pass takes 32 + 32 KB and two workgroups fit a CU again.
    const int a_bytes = KBT > T2_KB ? (ucap + 1) * 128 : t2_a_bytes(ucap);
    const int main_bytes = t2_main_bytes(ucap, KBT);
    uint8_t *const As = smem;
    uint16_t *const slot_s = reinterpret_cast<uint16_t *>(smem + a_bytes);
    constexpr int T2_KL_BYTES = 16 + 2 * KBT;
    uint16_t *const klist = reinterpret_cast<uint16_t *>(smem + main_bytes);
    int32_t *const idx_s = reinterpret_cast<int32_t *>(smem + main_bytes + T2_KL_BYTES);
    float *const sacc = reinterpret_cast<float *>(smem + main_bytes + T2_KL_BYTES + T2_IDX_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform by construction: lets everything derived from it live in SGPRs
    const int nt_total = cout >> 5, ks_total = cin >> 4, nchunk = cin >> 6;

    // XCD-aware unit order: workgroup b runs on XCD b % 8 (own L2) and workgroups are dispatched in index order, so XCD x
    // walks ONE contiguous range of units: the halo rows two neighbouring tiles share meet in one L2
    int u;
    {
        const int u_lo = nunit >> 3, u_rem = nunit & 7, x = bx & 7;
        u = x * u_lo + (x < u_rem ? x : u_rem) + (bx >> 3);
        if ((int)(bx >> 3) >= u_lo + (x < u_rem ? 1 : 0)) return;      // (grid rounded up to a multiple of 8)
    }
    // Two workgroups share a CU and start together: left alone they stage, multiply and store in lockstep -- the matrix
    // pipe idles while both gather, and is contended while both multiply (profiles/r03_tile_trace.txt).  The workgroups
    // that land in a CU's second slot (dispatch fills every CU of an XCD once before it doubles up: 32 workgroups per XCD =
    // 256 per round) start a few microseconds late, so that one's gather / exchange / store falls into the other's multiply.
    if ((stagger & 255) > 0 && ((blockIdx.x >> 8) & 1))
        for (int i = 0; i < (stagger & 255); i++) __builtin_amdgcn_s_sleep(64);      // 64 x 64 cycles ~ 2.2 us each
    const int64_t tile = tile_base + u / (ny * gz);
    const int yb = (u / gz) % ny, zi = u % gz;
    int64_t row0 = tile * T2_TM, wslot0 = 0;
    int rows = (int)(n_out - row0 < T2_TM ? n_out - row0 : T2_TM);
    if (tiles) { wslot0 = (int64_t)tiles[tile * 3] * K; row0 = tiles[tile * 3 + 1]; rows = tiles[tile * 3 + 2]; }
    const int np = npass[tile];
    if (stats && tid < 2 * NC) sacc[tid] = 0.f;
#ifdef CG3D_TILE_TRACE
    const unsigned long long tr_c0 = __builtin_readcyclecounter(), tr_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tr_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int tr_n = 0;
#define T2_STAMP() do { if (tr_n < 8) tr_ph[tr_n] = __builtin_amdgcn_s_memrealtime(); tr_n++; } while (0)
#else
#define T2_STAMP() do { } while (0)
#endif

    // consumer identity
    const int r = lane & 31, kg = lane >> 5;
    const int g = wave / NCO, h = wave % NCO;
    const int nt0 = (yb * NCO + h) * 2;                 // this wave's first 32-channel output block
    const int first = zi * KG + g, stride = gz * KG;    // this wave's share of the live offsets
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
        for (int n = 0; n < 2; n++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[m][n][e] = 0.f;

    for (int p = 0; p < np; p++) {
        const int32_t *pt = pass_tab + (tile * maxpass + p) * 4;
        const int k0 = pt[0], k1 = pt[1], uoff = pt[2], ucnt = pt[3];
        const int ngran = ucnt * 8;
        // row indices of the pass -> LDS (two per thread; 16 registers per lane held over the multiply loop spilled): every
        // chunk of the pass reads its 16 per thread back right before it issues the row requests
        // (idx_s is only read by the staging code, before a stage's barrier: every wave is past the previous pass's reads here;
        // the barrier that opens the pass's first stage publishes the new indices)
        {
            const int i0 = tid * 2;
            if (i0 < ucnt) idx_s[i0] = ulist[uoff + i0];
            if (i0 + 1 < ucnt) idx_s[i0 + 1] = ulist[uoff + i0 + 1];
        }
        // Stage order.  Default: slot-table block OUTSIDE, channel chunk inside -- the table of a block is re-laid once and the rows
        // of the pass are staged again for every (block, chunk) of a pass of several chunks.  CG3D_TILE_CHUNK_OUTER=1 (stagger
        // bit 8) turns it round: rows of (pass, chunk) staged once, the table re-laid per (chunk, block).  Measured on the 9^3 / 5^3
        // class convolutions on split rows (192 channels; profiles/r06_ab_rocprof_tile.txt): with blocks of 128 offsets 298 us
        // per launch block-outside against 324 chunk-outside -- re-laying a 32 KB table through registers (64 two-byte LDS
        // stores per thread) costs more than another asynchronous LDS-DMA fill of the rows; with blocks of 32 offsets (23 blocks
        // per pass) chunk-outside wins by 8 %, but blocks of 128 beat both (404 -> 298 us).
        const int nblk = k1 - k0 > KBT ? (k1 - k0 + KBT - 1) / KBT : 1;      // (an empty pass still is one stage)
        const bool chunk_outer = (stagger & 256) != 0;
        {
            for (int sidx = 0; sidx < nblk * nchunk; sidx++) {
                const int c = chunk_outer ? sidx / nblk : sidx % nchunk, bi = chunk_outer ? sidx % nblk : sidx / nchunk;
                const int kb = k0 + bi * KBT;
                const int nk = k1 - kb < KBT ? (k1 - kb > 0 ? k1 - kb : 0) : KBT;
                // rows of (pass, chunk): staged at the chunk's first block and kept (block-outside order: at every stage of a
                // pass of several chunks); table of a block: resident over the chunks only when the pass is ONE block
                const bool stage_rows = chunk_outer ? bi == 0 : (nchunk > 1 || bi == 0);
                const bool stage_slots = chunk_outer ? (c == 0 || nblk > 1) : c == 0;
                // ------------------------------------------------------------------------------------ stage
                __syncthreads();                        // every wave has finished reading the row tile / slot table; idx_s is written
                // an opaque copy of the thread id for everything the staging code addresses: without it the compiler hoists
                // ~40 loop-invariant per-lane addresses out of the pass / block / chunk loops and keeps them alive across the
                // multiply loop, where every register is taken (64 spilled registers)
                int stid = tid;
                asm volatile("" : "+v"(stid));
                // slot table of offsets [kb, kb+nk): global [k][row] -> LDS [kk][r][m] (lane r reads the slots of rows r,
                // 32+r, 64+r, 96+r as ONE 8-byte word); requested before the rows so that it returns first
                constexpr int NSV = KBT / 16;            // 16-byte pieces of the block's table per thread
                uint4 sv[NSV];
                int lv = 0, lv2 = 0;
                if (stage_slots) {
                    const uint16_t *slots_t = slots + (tile * K + kb) * (int64_t)T2_TM;
#pragma unroll
                    for (int j = 0; j < NSV; j++) {
                        const int i = j * 256 + stid;
                        const int ic = i < nk * (T2_TM / 8) ? i : 0;
                        sv[j] = *reinterpret_cast<const uint4 *>(slots_t + (int64_t)(ic >> 4) * T2_TM + (ic & 15) * 8);
                    }
                    if (wave == 0) {
                        lv = lane < nk ? live[tile * (int64_t)K + kb + lane] : 0;
                        if (KBT > 64) lv2 = lane + 64 < nk ? live[tile * (int64_t)K + kb + 64 + lane] : 0;
                    }
                }
                // rows: global memory -> LDS row tile by LDS-DMA (global_load_lds_dwordx4: no data registers; a request fills
                // wave-uniform base + lane * 16).  Lane i of request j owns PHYSICAL granule i & 7 of row slot (i >> 3) + 1
                // and fetches the logical granule the XOR swizzle maps there (granule g of slot s sits at g ^ ((s >> 1) & 7):
                // 16 lanes reading one channel granule of 16 consecutive slots hit 16 different bank groups).
                if (stage_rows && ngran > 0 && !(T2_DBG & 8)) {
#pragma unroll
                    for (int j = 0; j < T2_NLV; j++) {
                        if (j * 256 < ngran) {                               // uniform
                            const int i = j * 256 + stid;
                            const int sl = (i >> 3) + 1, gr = (i & 7) ^ ((sl >> 1) & 7);
                            const int32_t row = idx_s[i < ngran ? (i >> 3) : 0];
                            const uint16_t *src = X + ((int64_t)row * cin + c * 64 + gr * 8);
                            uint8_t *dst = As + (size_t)(j * 256 + wave * 64 + 8) * 16;
                            if (i < ngran)
                                __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void *)src,
                                                                 (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
                        }
                    }
                }
                if (stid < 8 && stage_rows) reinterpret_cast<uint4 *>(As)[stid] = make_uint4(0u, 0u, 0u, 0u);      // the zero row
                if (stage_slots) {
#pragma unroll
                    for (int j = 0; j < NSV; j++) {
                        const int i = j * 256 + stid;
                        if (i < nk * (T2_TM / 8)) {
                            const int kk = i >> 4, row8 = (i & 15) * 8;
                            const uint32_t w4[4] = {sv[j].x, sv[j].y, sv[j].z, sv[j].w};
#pragma unroll
                            for (int q = 0; q < 8; q++) {
                                const int row = row8 + q;
                                // stored as the BYTE ADDRESS of the row's granule for (ks 0, kg 0): slot * 128 | swizzle << 4
                                // (<= 65520: ucap <= 511); the multiply loop then needs one XOR per fragment read
                                const uint32_t sl = (w4[q >> 1] >> ((q & 1) * 16)) & 0xffffu;
                                slot_s[kk * T2_TM + (row & 31) * 4 + (row >> 5)] = (uint16_t)((sl << 7) | (((sl >> 1) & 7u) << 4));
                            }
                        }
                    }
                    if (wave == 0) {                    // the live offsets of the block, compacted with one ballot (two: 128 offsets)
                        const uint64_t bal = __ballot(lv != 0);
                        if (lv) klist[__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(lane | (lv << 8));
                        int total = __popcll(bal);
                        if (KBT > 64) {
                            const uint64_t bal2 = __ballot(lv2 != 0);
                            if (lv2) klist[total + __popcll(bal2 & ((1ull << lane) - 1ull))] = (uint16_t)((lane + 64) | (lv2 << 8));
                            total += __popcll(bal2);
                        }
                        if (lane == 0) klist[KBT] = (uint16_t)total;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);          // the LDS-DMA requests of this wave have landed (vmcnt), its LDS writes too
                __syncthreads();
                T2_STAMP();                             // staged

                // ---------------------------------------------------------------------------------- multiply
                const int nlive = __builtin_amdgcn_readfirstlane(klist[KBT]);       // (uniform: keeps the step loop a scalar loop)
                const int nstep = first < nlive ? (nlive - first + stride - 1) / stride : 0;     // offsets of this wave
                if (nstep > 0) {
                    // weight fragments of (offset kk, 16-channel group ks, output block n): Wf[slot][nt][ks][lane][8]
                    // (wrev: offset k reads weight slot K-1-k -- the data gradient of a map onto itself walks the FORWARD plan)
                    const uint16_t *wbase = Wf + ((wslot0 + ((wrev & 1) ? K - 1 - kb : kb)) * nt_total + nt0) * (int64_t)ks_total * 512 +
                                            (int64_t)c * 4 * 512 + lane * 8;
                    const int64_t wstride = ((wrev & 1) ? -1 : 1) * (int64_t)nt_total * ks_total * 512;       // per offset
                    const int64_t wn = (int64_t)ks_total * 512;                        // per 32-channel output block
                    // LDS byte address of (row block m, ks = 0) for this lane's channel half; granule g of slot s sits at
                    // g ^ ((s >> 1) & 7) and g = 2 ks + kg, so the address of ks is  a0 ^ (kg << 4) ^ (ks << 5)
                    struct Rows { uint32_t a[4]; };
                    // this wave's offsets of the block, one per lane of ONE register (a single LDS read): the offset of step st
                    // is a v_readlane away -- a scalar, so the weight addresses are scalar arithmetic and no step starts with
                    // the LDS round trip list -> slot row -> A fragments
                    const int kreg = klist[first + (lane < nstep ? lane : nstep - 1) * stride];       // offset | live bits << 8
                    auto kk_of = [&](int st) -> int { return __builtin_amdgcn_readlane(kreg, st < nstep ? st : nstep - 1) & 0xff; };
                    const uint16_t *slot_lane = slot_s + r * 4;
                    const uint32_t kg16 = (uint32_t)kg << 4;
                    auto slot_raw = [&](int kk) -> uint2 { return *reinterpret_cast<const uint2 *>(slot_lane + kk * T2_TM); };
                    auto rows_from = [&](const uint2 s2) -> Rows {
                        Rows R;
                        R.a[0] = (s2.x & 0xffffu) ^ kg16; R.a[1] = (s2.x >> 16) ^ kg16;
                        R.a[2] = (s2.y & 0xffffu) ^ kg16; R.a[3] = (s2.y >> 16) ^ kg16;
#if T2_DBG & 2
#pragma unroll
                        for (int m = 0; m < 4; m++) {
                            const uint32_t sl = (uint32_t)(m * 32 + r + 1) + (R.a[m] & 1u);       // (bit 0 is always clear: keeps the table read alive)
                            R.a[m] = ((sl << 7) | (((sl >> 1) & 7u) << 4)) ^ kg16;
                        }
#endif
                        return R;
                    };
                    auto load_b = [&](uint4 (&b)[2], int kk, int ks) {
                        const uint16_t *wk = wbase + kk * wstride + ks * 512;
                        b[0] = *reinterpret_cast<const uint4 *>(wk);
                        b[1] = *reinterpret_cast<const uint4 *>(wk + wn);
                    };
                    uint4 b[4][2];
                    bf16x8 aA[4], aB[4];
                    int knext = kk_of(1);
                    {
                        const int kcur = kk_of(0);
                        // in THIS order, and the compiler must keep it: the step loop's first MFMA needs b[0]; with the prologue's
                        // requests reordered (it issued b[3] first) the loop header waits for vmcnt(0) on EVERY iteration --
                        // a full L2 round trip per step for the fragments requested at the end of the previous one
#pragma unroll
                        for (int ks = 0; ks < 4; ks++) { load_b(b[ks], kcur, ks); __builtin_amdgcn_sched_barrier(0); }
                        const Rows R0 = rows_from(slot_raw(kcur));
#pragma unroll
                        for (int m = 0; m < 4; m++) aA[m] = *reinterpret_cast<const bf16x8 *>(As + R0.a[m]);
                    }
                    Rows R = rows_from(slot_raw(kk_of(0)));
                    uint2 sraw_n = slot_raw(knext);      // the NEXT offset's slot row, requested a whole step before it is decoded
                    // One step = one offset x 64 input channels = 4 sub-steps of 16 channels.  The body is specialised on the
                    // offset's LIVE MASK (bit m: some row of the tile's 32-row block m has a neighbour at this offset; the plan's
                    // `live` bits ANDed with the row blocks this unit owns): a dead block costs neither its two MFMAs nor its
                    // fragment read.  Per sub-step: the MFMAs of the live blocks, the LDS reads of the NEXT sub-step's A fragments
                    // (the last sub-step requests all four blocks of the next offset: its mask selects among them) and the two
                    // weight-fragment loads of the next offset, interleaved by rule.
                    auto step = [&](auto MASK_, const int knn) {
                        constexpr int MASK = decltype(MASK_)::value;
                        constexpr int NM = ((MASK >> 0) & 1) + ((MASK >> 1) & 1) + ((MASK >> 2) & 1) + ((MASK >> 3) & 1);
                        auto read_live = [&](bf16x8 (&a)[4], const Rows &Rr, auto KS_) {
                            constexpr uint32_t kx = (uint32_t)decltype(KS_)::value << 5;
#pragma unroll
                            for (int m = 0; m < 4; m++)
                                if ((MASK >> m) & 1) a[m] = *reinterpret_cast<const bf16x8 *>(As + (Rr.a[m] ^ kx));
                        };
                        auto mma = [&](const bf16x8 (&a)[4], const uint4 (&bb)[2]) {
                            const bf16x8 b0 = __builtin_bit_cast(bf16x8, bb[0]), b1 = __builtin_bit_cast(bf16x8, bb[1]);
#pragma unroll
                            for (int m = 0; m < 4; m++)
                                if ((MASK >> m) & 1) {
#if T2_DBG & 4
                                    asm volatile("" :: "v"(a[m]), "v"(b0), "v"(b1));
#else
                                    acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b0, acc[m][0], 0, 0, 0);
                                    acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b1, acc[m][1], 0, 0, 0);
#endif
                                }
                        };
                        auto sched = [&](auto NDS_) {                          // 2 NM MFMAs, NDS LDS reads, 2 weight loads
                            constexpr int NDS = decltype(NDS_)::value, NMF = 2 * NM;
                            constexpr int per = NMF > 0 ? (NDS + NMF - 1) / NMF : 0;       // LDS reads behind each of the first MFMAs
                            constexpr int lead = NMF > 0 && per > 0 ? (NDS + per - 1) / per : 0;
#pragma unroll
                            for (int q_ = 0; q_ < lead; q_++) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // VALU (address)
                                __builtin_amdgcn_sched_group_barrier(0x100, per, 0);    // DS read
                            }
#pragma unroll
                            for (int q_ = lead; q_ < NMF; q_++) {
                                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                __builtin_amdgcn_sched_group_barrier(0x006, 3, 0);      // VALU / SALU
                                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
                        using K3 = std::integral_constant<int, 3>;
                        read_live(aB, R, K1{});
                        mma(aA, b[0]);
                        if (!(T2_DBG & 1)) load_b(b[0], knext, 0);
                        sched(std::integral_constant<int, NM>{});
                        read_live(aA, R, K2{});
                        mma(aB, b[1]);
                        if (!(T2_DBG & 1)) load_b(b[1], knext, 1);
                        sched(std::integral_constant<int, NM>{});
                        read_live(aB, R, K3{});
                        const Rows Rn = rows_from(sraw_n);
                        mma(aA, b[2]);
                        if (!(T2_DBG & 1)) load_b(b[2], knext, 2);
                        sched(std::integral_constant<int, NM>{});
#pragma unroll
                        for (int m = 0; m < 4; m++) aA[m] = *reinterpret_cast<const bf16x8 *>(As + Rn.a[m]);
                        sraw_n = slot_raw(knn);
                        mma(aB, b[3]);
                        if (!(T2_DBG & 1)) load_b(b[3], knext, 3);
                        sched(std::integral_constant<int, 5>{});
                        R = Rn;
                        knext = knn;
                    };
                    // Skipping the dead 32-row blocks of an offset (the plan's `live` bits; rows sorted by their live offsets inside
                    // the tile make 16 % of the blocks of a 3^3 map onto itself dead) was tried three ways and is not kept:
                    //  * one loop switching between 15 mask-specialised bodies, or C++ branches around each block's MFMAs: the
                    //    register allocator spills 700-900 registers at the joins;
                    //  * 15 plain loops in a row over the offsets grouped by mask: clean hot loops, ~120 scratch operations per
                    //    stage between them -- 7.6 vs 4.4 ms over the step's layers;
                    //  * the two MFMAs of a block behind s_bitcmp1 / s_cbranch_scc0 INSIDE one asm statement (straight-line code
                    //    to the compiler, no spills): correct, but a branch per 64 cycles of MFMA and no sched_group_barrier
                    //    interleave cost as much as the skipped blocks save (128 -> 128: 87 vs 88 us, 256 -> 256: 98 vs 88 us).
                    // `MASK` stays a template parameter of the body for that reason only.  
                    // (stagger bit 10, CG3D_TILE_SETPRIO=1: the multiply loop at wave priority 1 -- the other workgroup of the CU is
                    // usually in a phase of plain VALU / LDS work then; measured)
                    if (stagger & 1024) __builtin_amdgcn_s_setprio(1);
                    for (int st = 0; st < nstep; st++) step(std::integral_constant<int, 15>{}, kk_of(st + 2));
                    if (stagger & 1024) __builtin_amdgcn_s_setprio(0);
                }
                T2_STAMP();                             // multiplied
            }
        }
    }

    // ------------------------------------------------------------------------------------------------ exchange
    // The KG waves holding partial sums of the same 128 x 64 block exchange halves through the row tile, level by level;
    // wave g ends up owning 4 / KG of the 4 row blocks.  16 KB of LDS per wave.
    float4 *xch = reinterpret_cast<float4 *>(As);
    auto give = [&](auto GIVE, auto HALF) {
        constexpr int gv = decltype(GIVE)::value, half = decltype(HALF)::value;
        float4 *dst = xch + (size_t)wave * 16 * 64 + lane;
#pragma unroll
        for (int m = 0; m < half; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    dst[((m * 2 + n) * 4 + q) * 64] = make_float4(acc[gv + m][n][q * 4], acc[gv + m][n][q * 4 + 1],
                                                                  acc[gv + m][n][q * 4 + 2], acc[gv + m][n][q * 4 + 3]);
    };
    auto take = [&](auto KEEP, auto HALF, int partner) {
        constexpr int keep = decltype(KEEP)::value, half = decltype(HALF)::value;
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
    // the wave's final 32 * cnt rows x 64 channels, ROW-MAJOR in its own 16 KB region (a dword store per accumulator
    // register straight to global memory is store-issue bound: 128 store instructions per wave)
    auto to_rows = [&](auto LO, auto CNT) {
        constexpr int lo = decltype(LO)::value, cnt = decltype(CNT)::value;
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
    // BatchNorm statistics of the unit's output (per-channel sum / sum of squares of the rows this workgroup owns), EARLY: from
    // the accumulators right after the exchange, so that the workgroup's 2 NC global atomics are in flight while it lays its
    // rows out and stores them.  Issued at the very end (round 3-5: accumulated in the store loop) their round trip was the
    // tail of EVERY workgroup -- the slot it holds on its CU stays taken until they return: 64 -> 64 @ 155 773 rows 45 -> 56 us,
    // 128 -> 128 @ 82 107 rows 70 -> 78 us with statistics on, whatever the number of table slots (1 ... 1024 slots: the same
    // times, profiles/r06_tile2_stats_slots.txt -- not a same-address queue).  Rows past the end of the last tile multiply the
    // zero row: they add nothing.  With a bias the sums are those of acc + bias: the store loop's path below stays.
    const bool early_stats = stats != nullptr && bias == nullptr && (stagger & 512) && !(T2_DBG & 128);      // (bit 9 clear: CG3D_TILE_EARLY_STATS=0, A/B)
    auto stats_from_acc = [&](auto LO, auto CNT) {
        constexpr int lo = decltype(LO)::value, cnt = decltype(CNT)::value;
        float s0[2] = {0.f, 0.f}, s1[2] = {0.f, 0.f};
#pragma unroll
        for (int m = 0; m < cnt; m++)
#pragma unroll
            for (int n = 0; n < 2; n++)
#pragma unroll
                for (int e = 0; e < 16; e++) { const float v = acc[lo + m][n][e]; s0[n] += v; s1[n] += v * v; }
#pragma unroll
        for (int n = 0; n < 2; n++) {                   // lanes (r, kg = 0 / 1) hold different rows of channel n * 32 + r
            s0[n] += __shfl_xor(s0[n], 32);
            s1[n] += __shfl_xor(s1[n], 32);
            if (kg == 0) {                              // the KG waves of this channel block meet in LDS
                unsafeAtomicAdd(sacc + h * 64 + n * 32 + r, s0[n]);
                unsafeAtomicAdd(sacc + NC + h * 64 + n * 32 + r, s1[n]);
            }
        }
    };
    auto stats_flush = [&]() {
        if (tid < 2 * NC)
            unsafeAtomicAdd(&stats[((blockIdx.x % T2_STAT_SLOTS) * 2 + tid / NC) * (int64_t)cout + yb * NC + tid % NC], sacc[tid]);
    };
    __syncthreads();                                    // every wave has left the multiply loop: the row tile is free
    T2_STAMP();
    if ((T2_DBG & 16) && n_out >= 0) return;
    if (T2_DBG & 128) {
        if (g == 0) to_rows(I0{}, I1{}); else if (g == 1) to_rows(I1{}, I1{}); else if (g == 2) to_rows(I2{}, I1{}); else to_rows(I3{}, I1{});
    } else if constexpr (KG == 2) {
        const int partner = (g ^ 1) * NCO + h;
        if (g == 0) give(I2{}, I2{}); else give(I0{}, I2{});
        __syncthreads();
        if (g == 0) take(I0{}, I2{}, partner); else take(I2{}, I2{}, partner);
        if (early_stats) { if (g == 0) stats_from_acc(I0{}, I2{}); else stats_from_acc(I2{}, I2{}); }
        __syncthreads();                                // the partner has read my region (and the LDS sums are complete)
        if (early_stats) stats_flush();
        if (g == 0) to_rows(I0{}, I2{}); else to_rows(I2{}, I2{});
    } else {
        // KG == 4 (NCO == 1): level 0 between g and g ^ 2 (halves), level 1 between g and g ^ 1 (quarters)
        const int p0 = g ^ 2, p1 = g ^ 1;
        if ((g & 2) == 0) give(I2{}, I2{}); else give(I0{}, I2{});
        __syncthreads();
        if ((g & 2) == 0) take(I0{}, I2{}, p0); else take(I2{}, I2{}, p0);
        __syncthreads();
        if ((g & 2) == 0) { if ((g & 1) == 0) give(I1{}, I1{}); else give(I0{}, I1{}); }
        else              { if ((g & 1) == 0) give(I3{}, I1{}); else give(I2{}, I1{}); }
        __syncthreads();
        if ((g & 2) == 0) { if ((g & 1) == 0) take(I0{}, I1{}, p1); else take(I1{}, I1{}, p1); }
        else              { if ((g & 1) == 0) take(I2{}, I1{}, p1); else take(I3{}, I1{}, p1); }
        if (early_stats) {
            if (g == 0) stats_from_acc(I0{}, I1{}); else if (g == 1) stats_from_acc(I1{}, I1{});
            else if (g == 2) stats_from_acc(I2{}, I1{}); else stats_from_acc(I3{}, I1{});
        }
        __syncthreads();
        if (early_stats) stats_flush();
        if (g == 0) to_rows(I0{}, I1{}); else if (g == 1) to_rows(I1{}, I1{}); else if (g == 2) to_rows(I2{}, I1{}); else to_rows(I3{}, I1{});
    }
    // ------------------------------------------------------------------------------------------------ store
    // (the wave reads back what it wrote itself: LDS operations of one wave execute in order)
    {
        constexpr int rows_per = T2_TM / KG;
        const float *tb = reinterpret_cast<const float *>(As) + (size_t)wave * 4096;
        const int c4 = (lane & 15) * 4, rq = lane >> 4;
        const int col0 = (yb * NCO + h) * 64 + c4;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && zi == 0) bv = *reinterpret_cast<const float4 *>(bias + col0);
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
#pragma unroll
        for (int i = 0; i < rows_per / 4; i++) {
            float4 v = *reinterpret_cast<const float4 *>(tb + (i * 4 + rq) * 64 + c4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            const int pos = g * rows_per + i * 4 + rq;   // position in the tile -> output row (permuted tiles: `order`)
            if (pos < rows && !((T2_DBG & 64) && n_out >= 0)) {
                const int64_t orow = order ? (int64_t)order[row0 + pos] : row0 + pos;
                float *dst = Y + orow * cout + col0;
                if (wrev & CG3D_TILE_OUT_BF16) {          // Y holds bf16 rows: 8 bytes per lane, a row's 64 channels are one 128-byte line
                    typedef float t2_f32x2 __attribute__((ext_vector_type(2)));
                    typedef __bf16 t2_bf16x2 __attribute__((ext_vector_type(2)));
                    const t2_f32x2 p0 = {v.x, v.y}, p1 = {v.z, v.w};
                    *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(Y) + orow * cout + col0) =
                        make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(p0, t2_bf16x2)),
                                   __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, t2_bf16x2)));
                } else if (gz == 1) *reinterpret_cast<float4 *>(dst) = v;
                else { unsafeAtomicAdd(dst, v.x); unsafeAtomicAdd(dst + 1, v.y); unsafeAtomicAdd(dst + 2, v.z); unsafeAtomicAdd(dst + 3, v.w); }
                t0.x += v.x; t0.y += v.y; t0.z += v.z; t0.w += v.w;
                t1.x += v.x * v.x; t1.y += v.y * v.y; t1.z += v.z * v.z; t1.w += v.w * v.w;
            }
        }
        if (stats && !early_stats) {
            // per-channel sum / sum of squares of the rows this workgroup stored (BatchNorm statistics of the layer's
            // output): 4 lanes x KG waves share a column quad -> LDS atomics, then 2 NC global atomics per workgroup into
            // slot (workgroup % CG3D_BN_SLOTS) of the layer's zero-based table stats[slots][2][cout] (cg3d_bn_apply_sums
            // adds the slots up and derives mean / variance)
            float *a0 = sacc + h * 64 + c4, *a1 = sacc + NC + h * 64 + c4;
            unsafeAtomicAdd(a0, t0.x); unsafeAtomicAdd(a0 + 1, t0.y); unsafeAtomicAdd(a0 + 2, t0.z); unsafeAtomicAdd(a0 + 3, t0.w);
            unsafeAtomicAdd(a1, t1.x); unsafeAtomicAdd(a1 + 1, t1.y); unsafeAtomicAdd(a1 + 2, t1.z); unsafeAtomicAdd(a1 + 3, t1.w);
            __syncthreads();
            if (tid < 2 * NC)
                unsafeAtomicAdd(&stats[((blockIdx.x % T2_STAT_SLOTS) * 2 + tid / NC) * (int64_t)cout + yb * NC + tid % NC], sacc[tid]);
        }
    }
#ifdef CG3D_TILE_TRACE
    if (g_t2_trace && tid == 0) {
        unsigned long long *t = g_t2_trace + (size_t)blockIdx.x * 16;
#pragma unroll
        for (int i = 0; i < 8; i++) t[6 + i] = tr_ph[i];
        t[0] = tr_c0; t[1] = __builtin_readcyclecounter(); t[2] = tr_r0; t[3] = __builtin_amdgcn_s_memrealtime();
        t[4] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |      // HW_REG_HW_ID
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);   // HW_REG_XCC_ID
        t[5] = (unsigned long long)u;
    }
#endif
}

#define T2_PARAMS                                                                                                              \
    const uint16_t *__restrict__ X, const uint16_t *__restrict__ Wf, const uint16_t *__restrict__ slots,                       \
        const uint8_t *__restrict__ live, const int32_t *__restrict__ pass_tab, const int32_t *__restrict__ npass,             \
        const int32_t *__restrict__ ulist, int32_t maxpass, int32_t ucap, const int32_t *__restrict__ tiles,                   \
        const int32_t *__restrict__ order, const float *__restrict__ bias, float *__restrict__ Y, int64_t n_out, int32_t K,    \
        int32_t cin, int32_t cout
#define T2_ARGS X, Wf, slots, live, pass_tab, npass, ulist, maxpass, ucap, tiles, order, bias, Y, n_out, K, cin, cout

// KBT = T2_KB_BIG (kernels of more than 64 offsets): a pass of the 9^3 class convolution on split rows is 6 table blocks x 3
// chunks = 18 stages instead of 23 x 3 = 69, each one two barriers, a table re-lay and a weight-stream restart around what were
// ~4 offsets of work per wave; the 24 KB of extra table cost the second workgroup of a CU (101 KB of LDS) -- these launches
// have about one unit per CU anyway (240 tiles of 18 classes x 4 scenes).
template <int NCO, int KBT>
__global__ __launch_bounds__(256, 2) void k_spconv_tile2(T2_PARAMS, int32_t nunit, int32_t ny, int32_t gz, int32_t wrev,
                                                         float *__restrict__ stats, int32_t stagger) {
    t2_unit<NCO, KBT>(T2_ARGS, nunit, ny, gz, wrev, stats, stagger, blockIdx.x, 0);
}
// The tail of a launch in half units.  Workgroups are dispatched in index order onto 512 slots (two per CU): a launch of
// 642 full units (the 128 -> 128 layers at tensor stride 4) runs one full round and then 130 units on half-empty CUs -- 1.65
// rounds of time for 1.25 rounds of work (profiles/r03_tile_units_scaling.txt).  Here the tiles of the last, partial round
// are cut into units of 64 output channels instead of 128: twice as many, half as long, spread over all CUs; they come
// last in index order, the full units before them keep their XCD-contiguous ranges.
__global__ __launch_bounds__(256, 2) void k_spconv_tile2_mix(T2_PARAMS, int32_t grid2, int32_t nunit2, int32_t ny2, int32_t nunit1,
                                                             int32_t ny1, int32_t tile_split, int32_t wrev,
                                                             float *__restrict__ stats, int32_t stagger) {
    if ((int)blockIdx.x < grid2) t2_unit<2, T2_KB>(T2_ARGS, nunit2, ny2, 1, wrev, stats, stagger, blockIdx.x, 0);
    else t2_unit<1, T2_KB>(T2_ARGS, nunit1, ny1, 1, wrev, stats, stagger & ~255, blockIdx.x - grid2, tile_split);
}

static int64_t t2_lds_bytes(int32_t ucap, int kb) {
    return (int64_t)t2_main_bytes(ucap, kb) + 16 + 2 * kb + T2_IDX_BYTES + 2 * 128 * sizeof(float);
}
extern "C" int64_t cg3d_spconv_tile_lds_bytes(int32_t ucap) { return t2_lds_bytes(ucap, T2_KB); }

// rows of [2][cout] floats of the `stats` table of cg3d_spconv_tile_fwd (accumulated with atomics; the caller zero-fills it)
extern "C" int32_t cg3d_spconv_tile_grid(int64_t ntile, int32_t cout, int32_t ksplit) {
    if (ntile < 0 || ntile > 0x7fffffffll || cout < 64 || ksplit < 1) return -1;
    return CG3D_BN_SLOTS;
}

extern "C" int cg3d_spconv_tile_fwd(const uint16_t *X, const uint16_t *Wf, const uint16_t *slots, const uint8_t *live,
                                    const int32_t *pass_tab, const int32_t *npass, const int32_t *ulist,
                                    int32_t maxpass, int32_t ucap, const int32_t *tiles, int64_t ntile, const int32_t *order,
                                    const float *bias, float *Y, int64_t n_in, int64_t n_out, int32_t K, int32_t cin,
                                    int32_t cout, int32_t ksplit, int32_t wrev, float *stats, cg3d_stream_t stream) {
    if ((stats && (ksplit != 1 || tiles || cout > 512)) || (order && tiles)) return CG3D_ERR_ARG;
    if (n_out < 0 || n_in < 0 || K < 1 || cin < 64 || (cin & 63) || cout < 64 || (cout & 63))
        return CG3D_ERR_ARG;
    if (ucap < T2_TM || ucap > 511 || ksplit < 1 || ksplit > 8 || ((uintptr_t)X & 15) || ((uintptr_t)Wf & 15)) return CG3D_ERR_ARG;
    if ((wrev & CG3D_TILE_OUT_BF16) && (ksplit != 1 || ((uintptr_t)Y & 7))) return CG3D_ERR_ARG;      // bf16 rows are stored, not added
    wrev &= 3;
    if (ntile == 0) return CG3D_OK;
    hipStream_t s = cg3d_hs(stream);
    if (ksplit > 1 && hipMemsetAsync(Y, 0, (size_t)n_out * cout * sizeof(float), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    static const int kb_big_min_k = getenv("CG3D_TILE_KB_BIG_MIN_K") ? atoi(getenv("CG3D_TILE_KB_BIG_MIN_K")) : 64;
    const bool big = K > kb_big_min_k;                // table blocks of 128 offsets (one workgroup per CU): the 5^3 / 9^3 kernels
    const size_t lds = (size_t)t2_lds_bytes(ucap, big ? T2_KB_BIG : T2_KB);
    // units of 128 output channels, or of 64 when the count is not a multiple of 128 (64, 192: the 64 -> 3 x 64 feature-offset
    // convolution of the yaw datasets, cagroup_head.py:170-172)
    // A launch of at most CG3D_TILE_NARROW units of 128 channels (fewer than one per CU: the 512-channel layers at tensor stride
    // 16, 5 330 rows = 42 tiles x 4) is cut into twice as many units of 64 channels instead.
    static const int narrow_env = getenv("CG3D_TILE_NARROW") ? atoi(getenv("CG3D_TILE_NARROW")) : 0;
    const bool wide = (cout & 127) == 0 && !(ksplit == 1 && ntile * (cout / 128) <= narrow_env);
    const int32_t ny = wide ? cout / 128 : cout / 64;
    const int64_t nunit = ntile * ny * ksplit;
    if (nunit > 0x7ffffff0ll) return CG3D_ERR_ARG;
    const unsigned grid = (unsigned)((nunit + 7) / 8 * 8);
    static const int stagger_env = getenv("CG3D_TILE_STAGGER") ? atoi(getenv("CG3D_TILE_STAGGER")) : 2;
    static const int chunk_outer_env = getenv("CG3D_TILE_CHUNK_OUTER") ? atoi(getenv("CG3D_TILE_CHUNK_OUTER")) : 0;
    static const int early_stats_env = getenv("CG3D_TILE_EARLY_STATS") ? atoi(getenv("CG3D_TILE_EARLY_STATS")) : 1;
    static const int setprio_env = getenv("CG3D_TILE_SETPRIO") ? atoi(getenv("CG3D_TILE_SETPRIO")) : 0;
    const int32_t stagger = (nunit > 256 ? (stagger_env & 255) : 0) | (chunk_outer_env ? 256 : 0) | (early_stats_env ? 512 : 0) | (setprio_env ? 1024 : 0);
#define T2_LAUNCH(NW, KB)                                                                                                      \
    do {                                                                                                                       \
        static bool attr = false;                                                                                              \
        if (!attr) {                                                                                                           \
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spconv_tile2<NW, KB>),                                   \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (KB > T2_KB ? 104 : 80) * 1024) != hipSuccess) \
                return CG3D_ERR_LAUNCH;                                                                                        \
            attr = true;                                                                                                       \
            if (getenv("CG3D_TILE_INFO")) {                                                                                    \
                int nb = -1;                                                                                                   \
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_spconv_tile2<NW, KB>, 256, lds);                     \
                fprintf(stderr, "k_spconv_tile2<%d, %d>: lds %zu B, occupancy %d workgroups/CU\n", NW, KB, lds, nb);          \
            }                                                                                                                  \
        }                                                                                                                      \
        hipLaunchKernelGGL((k_spconv_tile2<NW, KB>), dim3(grid), dim3(256), lds, s, X, Wf, slots, live, pass_tab, npass, ulist, \
                           maxpass, ucap, tiles, order, bias, Y, n_out, K, cin, cout, (int32_t)nunit, ny, ksplit, wrev, stats, stagger); \
    } while (0)
    // tail in half units (see k_spconv_tile2_mix): only when the last round is at most half full
    static const int tail_env = getenv("CG3D_TILE_TAIL") ? atoi(getenv("CG3D_TILE_TAIL")) : 1;
    const int64_t SLOTS = 512, rem = nunit % SLOTS;
    const int64_t tail_tiles = (tail_env && !big && wide && ksplit == 1 && nunit > SLOTS && rem > 0 && 2 * rem <= SLOTS) ? rem / ny : 0;
    if (tail_tiles > 0) {
        static bool attr_mix = false;
        if (!attr_mix) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spconv_tile2_mix), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    80 * 1024) != hipSuccess)
                return CG3D_ERR_LAUNCH;
            attr_mix = true;
        }
        const int32_t tile_split = (int32_t)(ntile - tail_tiles), ny1 = cout / 64;
        const int32_t nunit2 = tile_split * ny, nunit1 = (int32_t)tail_tiles * ny1;
        const int32_t grid2 = (nunit2 + 7) / 8 * 8;
        hipLaunchKernelGGL(k_spconv_tile2_mix, dim3((unsigned)(grid2 + (nunit1 + 7) / 8 * 8)), dim3(256), lds, s, X, Wf, slots, live,
                           pass_tab, npass, ulist, maxpass, ucap, tiles, order, bias, Y, n_out, K, cin, cout, grid2, nunit2, ny, nunit1,
                           ny1, tile_split, wrev, stats, stagger);
    } else if (big) {
        if (wide) T2_LAUNCH(2, T2_KB_BIG); else T2_LAUNCH(1, T2_KB_BIG);
    } else if (wide) T2_LAUNCH(2, T2_KB); else T2_LAUNCH(1, T2_KB);
#undef T2_LAUNCH
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
