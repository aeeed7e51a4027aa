/*
 * oracle_tile.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Host restatement of the tile-plan sparse convolution of cagroup3d_amd/csrc/spconv_tile.hip: the same data
 * contract (plan = per tile of 128 output rows: passes, list of distinct input rows, slot table, liveness), the
 * same arithmetic (bf16 operands, fp32 accumulate) -- the MinkowskiEngine ConvolutionForward algorithm
 * (ME v0.5.4 un-vendored: PARITY UNPINNED against ME, pinned against dense conv3d through
 * oracle_conv.c in tests/).  The convolution here reads its neighbours ONLY through the plan, so comparing it with
 * cg3d_spconv_fwd on the dense map checks that a plan is a faithful re-encoding of the kernel map.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/cagroup3d_hip.h"

#define TP_TM 128

static inline float ot_bf16_bits(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static inline uint16_t ot_bf16(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline int64_t ot_frag_index(int n_idx, int k_idx, int kdim) {
    return ((((int64_t)(n_idx >> 5) * (kdim >> 4) + (k_idx >> 4)) * 64) + ((k_idx >> 3) & 1) * 32 + (n_idx & 31)) * 8 + (k_idx & 7);
}

int64_t cg3d_spconv_tile_lds_bytes(int32_t ucap) { return (int64_t)(ucap + 1) * 128 + 32 * TP_TM * 2 + 34 * 2; }

int cg3d_spconv_prep_weights_frag(const float *W0, const float *const *Ws, uint16_t *Wf_t, uint16_t *Wf, int32_t G,
                                  int64_t slots_per, int32_t cin, int32_t cout, cg3d_stream_t s) {
    (void)s;
    if (G < 1 || slots_per < 0 || cin < 1 || cout < 1 || (!W0 && !Ws) || (!Wf_t && !Wf)) return CG3D_ERR_ARG;
    if (Wf_t && ((cin & 15) || (cout & 31))) return CG3D_ERR_ARG;
    if (Wf && ((cout & 15) || (cin & 31))) return CG3D_ERR_ARG;
    const int64_t per = (int64_t)cin * cout;
    for (int64_t slot = 0; slot < (int64_t)G * slots_per; slot++) {
        const float *src = Ws ? Ws[slot / slots_per] + (slot % slots_per) * per : W0 + slot * per;
        for (int32_t ci = 0; ci < cin; ci++)
            for (int32_t co = 0; co < cout; co++) {
                const uint16_t b = ot_bf16(src[(int64_t)ci * cout + co]);
                if (Wf_t) Wf_t[slot * per + ot_frag_index(co, ci, cin)] = b;
                if (Wf) Wf[slot * per + ot_frag_index(ci, co, cout)] = b;
            }
    }
    return CG3D_OK;
}

/* Rows sorted by (set of live offsets, row) inside windows of CG3D_TILE_WINDOW rows (spconv_tile.hip, k_tile_row_order). */
typedef struct { uint32_t sig; int32_t row; } ot_key;
static int ot_key_cmp(const void *a, const void *b) {
    const ot_key *x = (const ot_key *)a, *y = (const ot_key *)b;
    if (x->sig != y->sig) return x->sig < y->sig ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row);
}
int cg3d_tile_row_order(const int32_t *nbr, int32_t K, int64_t n_out, int32_t window, int32_t *order, cg3d_stream_t s) {
    (void)s;
    if (K < 1 || K > 32 || n_out < 0 || window < 128 || window > CG3D_TILE_WINDOW || (window & (window - 1))) return CG3D_ERR_ARG;
    ot_key *keys = (ot_key *)malloc(sizeof(ot_key) * CG3D_TILE_WINDOW);
    for (int64_t w0 = 0; w0 < n_out; w0 += window) {
        const int n = (int)(n_out - w0 < window ? n_out - w0 : window);
        for (int i = 0; i < n; i++) {
            uint32_t sig = 0;
            for (int k = 0; k < K; k++) sig |= (nbr[(int64_t)k * n_out + w0 + i] >= 0 ? 1u : 0u) << k;
            keys[i].sig = sig; keys[i].row = (int32_t)(w0 + i);
        }
        qsort(keys, (size_t)n, sizeof(ot_key), ot_key_cmp);
        for (int i = 0; i < n; i++) order[w0 + i] = keys[i].row;
    }
    free(keys);
    return CG3D_OK;
}

/* Greedy passes in offset order; a row's slot = 1 + its rank of first appearance within the pass. */
int cg3d_tile_plan_build(const int32_t *nbr, int32_t K, int64_t n_out, const int32_t *tiles, int64_t ntile, int32_t ucap,
                         int32_t maxpass, uint16_t *slots, uint8_t *live, int32_t *pass_tab, int32_t *npass,
                         int32_t *ulist, int64_t ulist_cap, int32_t *cursor, const int32_t *order, cg3d_stream_t s) {
    (void)s;
    if (K < 1 || n_out < 0 || ntile < 0 || ucap < TP_TM || ucap > 1023 || maxpass < 1 || (order && tiles)) return CG3D_ERR_ARG;
    if (!tiles && ntile != (n_out + TP_TM - 1) / TP_TM) return CG3D_ERR_ARG;
    cursor[0] = cursor[1] = 0;
    int32_t *ul = (int32_t *)malloc(sizeof(int32_t) * 1024);
    for (int64_t t = 0; t < ntile; t++) {
        int64_t row0 = tiles ? tiles[t * 3 + 1] : t * TP_TM;
        int rows = tiles ? tiles[t * 3 + 2] : (int)(n_out - row0 < TP_TM ? n_out - row0 : TP_TM);
        uint16_t *st = slots + t * (int64_t)K * TP_TM;
        int32_t *pt = pass_tab + t * (int64_t)maxpass * 4;
        int ucount = 0, k0 = 0, np = 0;
        for (int k = 0; k <= K; k++) {
            int newc = 0;
            if (k < K)
                for (int r = 0; r < rows; r++) {
                    const int32_t g = nbr[(int64_t)k * n_out + (order ? order[row0 + r] : row0 + r)];
                    if (g < 0) continue;
                    int found = 0;
                    for (int u = 0; u < ucount && !found; u++) found = ul[u] == g;
                    newc += !found;
                }
            if (k == K || ucount + newc > ucap) {          /* close the pass [k0, k) */
                if ((int64_t)cursor[0] + ucount > ulist_cap || np >= maxpass) { cursor[1] = 1; free(ul); return CG3D_OK; }
                pt[np * 4] = k0; pt[np * 4 + 1] = k; pt[np * 4 + 2] = cursor[0]; pt[np * 4 + 3] = ucount;
                memcpy(ulist + cursor[0], ul, sizeof(int32_t) * (size_t)ucount);
                cursor[0] += ucount;
                np++;
                ucount = 0;
                k0 = k;
                if (k == K) break;
            }
            int lv = 0;
            for (int r = 0; r < TP_TM; r++) {
                const int32_t g = r < rows ? nbr[(int64_t)k * n_out + (order ? order[row0 + r] : row0 + r)] : -1;
                int sl = 0;
                if (g >= 0) {
                    lv |= 1 << (r >> 5);
                    for (int u = 0; u < ucount && !sl; u++) if (ul[u] == g) sl = u + 1;
                    if (!sl) { ul[ucount] = g; sl = ++ucount; }
                }
                st[(int64_t)k * TP_TM + r] = (uint16_t)sl;
            }
            live[t * (int64_t)K + k] = (uint8_t)lv;
        }
        npass[t] = np;
    }
    free(ul);
    return CG3D_OK;
}

int cg3d_spconv_tile_fwd(const uint16_t *X, const uint16_t *Wf, const uint16_t *slots, const uint8_t *live,
                         const int32_t *pass_tab, const int32_t *npass, const int32_t *ulist, int32_t maxpass,
                         int32_t ucap, const int32_t *tiles, int64_t ntile, const int32_t *order, const float *bias, float *Y,
                         int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout, int32_t ksplit, int32_t wrev,
                         float *stats, cg3d_stream_t s) {
    (void)s; (void)n_in; (void)ucap;
    if ((stats && (ksplit != 1 || tiles || cout > 512)) || (order && tiles)) return CG3D_ERR_ARG;
    /* CG3D_TILE_OUT_BF16: Y is uint16 [n_out, cout] = bf16(fp32 sums + bias); the statistics are those of the fp32 sums */
    uint16_t *Y16 = NULL;
    if (wrev & CG3D_TILE_OUT_BF16) {
        if (ksplit != 1) return CG3D_ERR_ARG;
        Y16 = (uint16_t *)Y;
        Y = (float *)malloc(sizeof(float) * (size_t)(n_out > 0 ? n_out : 1) * cout);
        /* rows no tile covers keep what the caller left there */
        for (int64_t i = 0; i < n_out * cout; i++) Y[i] = ot_bf16_bits(Y16[i]);
    }
    wrev &= 1;
    if (n_out < 0 || K < 1 || cin < 64 || (cin & 63) || cout < 64 || (cout & 63)) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t t = 0; t < ntile; t++) {
        const int64_t row0 = tiles ? tiles[t * 3 + 1] : t * TP_TM;
        const int rows = tiles ? tiles[t * 3 + 2] : (int)(n_out - row0 < TP_TM ? n_out - row0 : TP_TM);
        const int64_t wslot0 = tiles ? (int64_t)tiles[t * 3] * K : 0;
        float *xq = (float *)malloc(sizeof(float) * (size_t)cin);
        for (int r = 0; r < rows; r++) {
            float *y = Y + (order ? (int64_t)order[row0 + r] : row0 + r) * cout;       /* position r of the tile -> output row */
            for (int c = 0; c < cout; c++) y[c] = bias ? bias[c] : 0.f;
        }
        for (int p = 0; p < npass[t]; p++) {
            const int32_t *pt = pass_tab + (t * (int64_t)maxpass + p) * 4;
            for (int k = pt[0]; k < pt[1]; k++) {
                const int lv = live[t * (int64_t)K + k];
                const uint16_t *wk = Wf + (wslot0 + (wrev ? K - 1 - k : k)) * (int64_t)cin * cout;
                for (int r = 0; r < rows; r++) {
                    const int sl = slots[(t * (int64_t)K + k) * TP_TM + r];
                    if (!sl || !((lv >> (r >> 5)) & 1)) continue;     /* a dead block is skipped by the kernel: its slots must be 0 */
                    const int64_t g = ulist[pt[2] + sl - 1];
                    for (int a = 0; a < cin; a++) xq[a] = ot_bf16_bits(X[g * cin + a]);
                    float *y = Y + (order ? (int64_t)order[row0 + r] : row0 + r) * cout;
                    for (int c = 0; c < cout; c++) {
                        float acc = 0.f;
                        for (int a = 0; a < cin; a++) acc += xq[a] * ot_bf16_bits(wk[ot_frag_index(c, a, cin)]);
                        y[c] += acc;
                    }
                }
            }
        }
        free(xq);
    }
    if (stats) {        /* the zero-based table [2][cout]: += sum and sum of squares over all rows (fp64, then added) */
        double *acc = (double *)calloc(2 * (size_t)cout, sizeof(double));
        for (int64_t o = 0; o < n_out; o++)
            for (int c = 0; c < cout; c++) {
                const double v = Y[o * cout + c];
                acc[c] += v;
                acc[cout + c] += v * v;
            }
        for (int c = 0; c < 2 * cout; c++) stats[c] += (float)acc[c];
        free(acc);
    }
    if (Y16) {
        for (int64_t i = 0; i < n_out * cout; i++) Y16[i] = ot_bf16(Y[i]);
        free(Y);
    }
    return CG3D_OK;
}
int32_t cg3d_spconv_tile_grid(int64_t ntile, int32_t cout, int32_t ksplit) {
    if (ntile < 0 || cout < 64 || ksplit < 1) return -1;
    return 1;
}

/* Y = X @ W (+ bias) on bf16 rows, fp32 accumulate: the 1x1x1 convolutions (reference call sites biresnet.py:270-280,
 * 308-315, cagroup_head.py:163-188; ME's kernel-size-1 convolution is a plain matrix product).  Wf = the weights in MFMA
 * fragment order (cg3d_spconv_prep_weights_frag, one slot): element (output n, contraction k) at ot_frag_index(n, k, cin).
 * stats: += sum / sum of squares per output channel into slot 0 of the statistics table. */
int cg3d_linear_fwd(const uint16_t *X, const uint16_t *Wf, const float *bias, float *Y, int64_t n, int32_t cin, int32_t cout,
                    int32_t ksplit, float *stats, float *partials, cg3d_stream_t s) {
    (void)s; (void)partials;          /* scratch of the device's split contraction: the restatement sums in one pass */
    /* CG3D_LINEAR_OUT_BF16 (a bit of `ksplit`, with ksplit == 1): Y is uint16 [n, cout] = bf16(fp32 sums + bias) */
    uint16_t *Y16 = NULL;
    if (ksplit & CG3D_LINEAR_OUT_BF16) {
        ksplit &= ~CG3D_LINEAR_OUT_BF16;
        if (ksplit != 1) return CG3D_ERR_ARG;
        Y16 = (uint16_t *)Y;
        Y = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * cout);
    }
    if (n < 0 || cin < 64 || (cin & 63) || cout < 64 || (cout & 63) || ksplit < 1 || ksplit > 256 || ksplit > (cin >> 6)) return CG3D_ERR_ARG;
    if (stats && ksplit != 1) return CG3D_ERR_ARG;
    float *w = (float *)malloc(sizeof(float) * (size_t)cin * cout);          /* [cout][cin] */
    for (int c = 0; c < cout; c++)
        for (int a = 0; a < cin; a++) w[(size_t)c * cin + a] = ot_bf16_bits(Wf[ot_frag_index(c, a, cin)]);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float xq[1024];
        for (int a = 0; a < cin && a < 1024; a++) xq[a] = ot_bf16_bits(X[i * cin + a]);
        for (int c = 0; c < cout; c++) {
            float acc = 0.f;
            const float *wc = w + (size_t)c * cin;
            if (cin <= 1024)
                for (int a = 0; a < cin; a++) acc += xq[a] * wc[a];
            else
                for (int a = 0; a < cin; a++) acc += ot_bf16_bits(X[i * cin + a]) * wc[a];
            Y[i * cout + c] = acc + (bias ? bias[c] : 0.f);
        }
    }
    free(w);
    if (stats) {
        double *acc = (double *)calloc(2 * (size_t)cout, sizeof(double));
        for (int64_t o = 0; o < n; o++)
            for (int c = 0; c < cout; c++) {
                const double v = Y[o * cout + c];
                acc[c] += v; acc[cout + c] += v * v;
            }
        for (int c = 0; c < 2 * cout; c++) stats[c] += (float)acc[c];
        free(acc);
    }
    if (Y16) {
        for (int64_t i = 0; i < n * cout; i++) Y16[i] = ot_bf16(Y[i]);
        free(Y);
    }
    return CG3D_OK;
}
