/*
 * oracle_conv.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Sparse convolution forward and weight gradient as gather -> multiply-accumulate over the kernel
 * map, the MinkowskiEngine ConvolutionForward / ConvolutionBackward algorithm (SURVEY.md 3.3;
 * ME v0.5.4 is not vendored -- PARITY UNPINNED against ME, pinned against dense conv3d in tests/).
 * Split from oracle_sparse.c only so this file can be built with FMA contraction and OpenMP for
 * the bench.py cpu_baseline leg; results are compared to the HIP kernels with a tolerance.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/cagroup3d_hip.h"

/* bf16 round-to-nearest-even of an fp32 value, returned as fp32 (precision==1 operands) */
static inline float os_bf16(float v) {
    uint32_t u; memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return v; /* NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    float r; memcpy(&r, &u, 4);
    return r;
}

static inline int64_t os_frag_index(int n_idx, int k_idx, int kdim) {       /* cg3d_spconv_prep_weights_frag layout */
    return ((((int64_t)(n_idx >> 5) * (kdim >> 4) + (k_idx >> 4)) * 64) + ((k_idx >> 3) & 1) * 32 + (n_idx & 31)) * 8 + (k_idx & 7);
}
int cg3d_spconv_prep_weights_bf16_multi(const float *W0, const float *const *Ws, uint16_t *Wb_t, uint16_t *Wb, int32_t G,
                                        int64_t slots_per, int32_t cin, int32_t cout, cg3d_stream_t s) {
    (void)s;
    if (G < 1 || slots_per < 0 || cin < 1 || cout < 1 || (!W0 && !Ws) || (!Wb_t && !Wb)) return CG3D_ERR_ARG;
    int64_t per = (int64_t)cin * cout;
    for (int64_t slot = 0; slot < (int64_t)G * slots_per; slot++) {
        const float *src = Ws ? Ws[slot / slots_per] + (slot % slots_per) * per : W0 + slot * per;
        for (int32_t ci = 0; ci < cin; ci++)
            for (int32_t co = 0; co < cout; co++) {
                float v = os_bf16(src[(int64_t)ci * cout + co]);
                uint32_t u; memcpy(&u, &v, 4);
                if (Wb) Wb[slot * per + (int64_t)ci * cout + co] = (uint16_t)(u >> 16);
                if (Wb_t) Wb_t[slot * per + (int64_t)co * cin + ci] = (uint16_t)(u >> 16);
            }
    }
    return CG3D_OK;
}
static inline float os_bf16_bits(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
/* element a of a gathered row: fp32 storage (precision 0: as is, 1: rounded to bf16) or bf16 storage (2) */
static inline float os_row(const void *base, int64_t i, int32_t precision) {
    if (precision == 2) return os_bf16_bits(((const uint16_t *)base)[i]);
    float v = ((const float *)base)[i];
    return precision == 1 ? os_bf16(v) : v;
}

int cg3d_to_bf16(const float *X, uint16_t *Xb, int64_t n, cg3d_stream_t s) {
    (void)s;
    if (n < 0 || (n & 3)) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float v = os_bf16(X[i]);
        uint32_t u; memcpy(&u, &v, 4);
        Xb[i] = (uint16_t)(u >> 16);
    }
    return CG3D_OK;
}

int cg3d_from_bf16(const uint16_t *Xb, float *X, int64_t n, cg3d_stream_t s) {
    (void)s;
    if (n < 0) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) X[i] = os_bf16_bits(Xb[i]);
    return CG3D_OK;
}

/* Split operands (include/cagroup3d_hip.h, "bf16x3"): hi = bf16(v), lo = bf16(v - hi); inf / NaN keep lo = 0. */
static inline void os_split(float v, uint16_t *hi, uint16_t *lo) {
    float h = os_bf16(v);
    uint32_t u; memcpy(&u, &h, 4);
    *hi = (uint16_t)(u >> 16);
    if ((*hi & 0x7f80u) == 0x7f80u) { *lo = 0; return; }
    float r = os_bf16(v - h);
    memcpy(&u, &r, 4);
    *lo = (uint16_t)(u >> 16);
}
int cg3d_to_bf16_split(const float *X, uint16_t *Xs, int64_t n_rows, int32_t c, cg3d_stream_t s) {
    (void)s;
    if (n_rows < 0 || c < 4 || (c & 3)) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rows; r++)
        for (int32_t a = 0; a < c; a++) {
            uint16_t hi, lo;
            os_split(X[r * c + a], &hi, &lo);
            Xs[r * 3 * c + a] = hi;
            Xs[r * 3 * c + c + a] = lo;
            Xs[r * 3 * c + 2 * c + a] = hi;
        }
    return CG3D_OK;
}
int cg3d_spconv_prep_weights_split(const float *W0, const float *const *Ws, uint16_t *W_t, uint16_t *W_p, int32_t G,
                                   int64_t slots_per, int32_t cin, int32_t cout, int32_t frag, cg3d_stream_t s) {
    (void)s;
    if (G < 1 || slots_per < 0 || cin < 1 || cout < 1 || (!W0 && !Ws) || (!W_t && !W_p)) return CG3D_ERR_ARG;
    if (frag && W_t && ((cin & 15) || (cout & 31))) return CG3D_ERR_ARG;
    if (frag && W_p && ((cout & 15) || (cin & 31))) return CG3D_ERR_ARG;
    int64_t per = (int64_t)cin * cout;
    for (int64_t slot = 0; slot < (int64_t)G * slots_per; slot++) {
        const float *src = Ws ? Ws[slot / slots_per] + (slot % slots_per) * per : W0 + slot * per;
        for (int32_t ci = 0; ci < cin; ci++)
            for (int32_t co = 0; co < cout; co++) {
                uint16_t hi, lo;
                os_split(src[(int64_t)ci * cout + co], &hi, &lo);
                for (int part = 0; part < 3; part++) {
                    uint16_t b = part < 2 ? hi : lo;
                    if (W_t) W_t[slot * 3 * per + (frag ? os_frag_index(co, part * cin + ci, 3 * cin) : (int64_t)co * 3 * cin + part * cin + ci)] = b;
                    if (W_p) W_p[slot * 3 * per + (frag ? os_frag_index(ci, part * cout + co, 3 * cout) : (int64_t)ci * 3 * cout + part * cout + co)] = b;
                }
            }
    }
    return CG3D_OK;
}

/* `tiles` (may be NULL): int32 [ntile,3] = (group, first row, row count): the output rows of a tile use the weights of
 * that group (W holds G stacked weight sets of K slots each); rows not covered by a tile are not written. */
int cg3d_spconv_fwd_tiled(const float *X, const float *W, const int32_t *nbr, const int32_t *tiles, int64_t ntile,
                          const float *bias, float *Y, int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout,
                          int32_t precision, cg3d_stream_t s) {
    (void)s; (void)n_in;
    if (n_out < 0 || K < 1 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    if (tiles && (precision == 0 || ntile < 0)) return CG3D_ERR_ARG;
    int32_t G = 1;
    int32_t *grp = NULL;                       /* group of every output row, -1 = not covered */
    if (tiles) {
        grp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_out > 0 ? n_out : 1));
        for (int64_t o = 0; o < n_out; o++) grp[o] = -1;
        for (int64_t t = 0; t < ntile; t++) {
            if (tiles[t * 3] + 1 > G) G = tiles[t * 3] + 1;
            for (int32_t r = 0; r < tiles[t * 3 + 2]; r++) grp[tiles[t * 3 + 1] + r] = tiles[t * 3];
        }
    }
    float *Wq = NULL;
    if (precision >= 1) {
        int64_t nw = (int64_t)G * K * cin * cout;
        K = G * K;                              /* widen all G*K slots; restored below */
        /* precision 1: W is the prepared bf16 [K][cout][cin] buffer; widen it back to fp32 [K][cin][cout] */
        const uint16_t *wb = (const uint16_t *)W;
        Wq = (float *)malloc((size_t)nw * sizeof(float));
        for (int32_t k = 0; k < K; k++)
            for (int32_t c = 0; c < cout; c++)
                for (int32_t a = 0; a < cin; a++) {
                    union { uint32_t u; float f; } v;
                    v.u = (uint32_t)wb[((int64_t)k * cout + c) * cin + a] << 16;
                    Wq[((int64_t)k * cin + a) * cout + c] = v.f;
                }
        W = Wq;
        K = K / G;
    }
#pragma omp parallel
    {
        float *xq = (float *)malloc((size_t)cin * sizeof(float));
#pragma omp for schedule(dynamic, 64)
        for (int64_t o = 0; o < n_out; o++) {
            if (grp && grp[o] < 0) continue;
            const int64_t slot0 = grp ? (int64_t)grp[o] * K : 0;
            float *y = Y + o * cout;
            for (int32_t c = 0; c < cout; c++) y[c] = bias ? bias[c] : 0.f;
            for (int32_t k = 0; k < K; k++) {
                int32_t i = nbr[(int64_t)k * n_out + o];
                if (i < 0) continue;
                for (int32_t a = 0; a < cin; a++) xq[a] = os_row(X, (int64_t)i * cin + a, precision);
                const float *x = xq;
                const float *w = W + (slot0 + k) * cin * cout;
                for (int32_t a = 0; a < cin; a++) {
                    float xa = x[a];
                    const float *wr = w + (int64_t)a * cout;
                    for (int32_t c = 0; c < cout; c++) y[c] += xa * wr[c];
                }
            }
        }
        free(xq);
    }
    free(Wq);
    free(grp);
    return CG3D_OK;
}
int cg3d_spconv_fwd(const float *X, const float *W, const int32_t *nbr, const float *bias, float *Y,
                    int64_t n_in, int64_t n_out, int32_t K, int32_t cin, int32_t cout, int32_t precision,
                    cg3d_stream_t s) {
    return cg3d_spconv_fwd_tiled(X, W, nbr, NULL, 0, bias, Y, n_in, n_out, K, cin, cout, precision, s);
}

int cg3d_spconv_wgrad(const float *X, const float *dY, const int32_t *nbr, float *dW, int64_t n_in,
                      int64_t n_out, int32_t K, int32_t cin, int32_t cout, int32_t precision,
                      cg3d_stream_t s) {
    (void)s; (void)n_in;
    if (n_out < 0 || K < 1 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    const int32_t AB = 16; /* cin block owned by one task */
    int32_t nab = (cin + AB - 1) / AB;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int32_t k = 0; k < K; k++)
        for (int32_t ab = 0; ab < nab; ab++) {
            int32_t a0 = ab * AB, a1 = a0 + AB < cin ? a0 + AB : cin;
            float *w = dW + (int64_t)k * cin * cout;
            for (int32_t a = a0; a < a1; a++)
                for (int32_t c = 0; c < cout; c++) w[(int64_t)a * cout + c] = 0.f;
            float *dq = precision == 1 ? (float *)malloc((size_t)cout * sizeof(float)) : NULL;
            for (int64_t o = 0; o < n_out; o++) {
                int32_t i = nbr[(int64_t)k * n_out + o];
                if (i < 0) continue;
                const float *x = X + (int64_t)i * cin;
                const float *d = dY + o * cout;
                if (precision == 1) { for (int32_t c = 0; c < cout; c++) dq[c] = os_bf16(d[c]); d = dq; }
                for (int32_t a = a0; a < a1; a++) {
                    float xa = precision == 1 ? os_bf16(x[a]) : x[a];
                    float *wr = w + (int64_t)a * cout;
                    for (int32_t c = 0; c < cout; c++) wr[c] += xa * d[c];
                }
            }
            free(dq);
        }
    return CG3D_OK;
}


/* table form (include/cagroup3d_hip.h): one row per 64 x 64 tile of one slot */
int cg3d_spconv_prep_weights_bf16_table(const int64_t *table, int64_t nrows, cg3d_stream_t s) {
    (void)s;
    if (nrows < 0 || (nrows > 0 && !table)) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nrows; b++) {
        const int64_t *row = table + b * 6;
        const float *src = (const float *)(uintptr_t)row[0];
        uint16_t *wt = (uint16_t *)(uintptr_t)row[1], *wp = (uint16_t *)(uintptr_t)row[2];
        const int cin = (int)row[3], cout = (int)row[4], tile = (int)(row[5] & 0xfffffff);
        const int frag_t = (int)((row[5] >> 30) & 1), frag_p = (int)((row[5] >> 29) & 1);   /* MFMA fragment order */
        const int split = (int)((row[5] >> 28) & 1);                                        /* three-part split operands */
        const int np = split ? 3 : 1;
        const int co_tiles = (cout + 63) / 64;
        const int ci0 = (tile / co_tiles) * 64, co0 = (tile % co_tiles) * 64;
        for (int r = ci0; r < ci0 + 64 && r < cin; r++)
            for (int c = co0; c < co0 + 64 && c < cout; c++) {
                uint16_t hi, lo = 0;
                if (split) os_split(src[(int64_t)r * cout + c], &hi, &lo);
                else {
                    const float f = os_bf16(src[(int64_t)r * cout + c]);
                    uint32_t u; memcpy(&u, &f, 4);
                    hi = (uint16_t)(u >> 16);
                }
                for (int part = 0; part < np; part++) {
                    const uint16_t v = part < 2 ? hi : lo;
                    const int kc_p = part * cout + c, kc_t = part * cin + r;
                    if (wp) wp[frag_p ? os_frag_index(r, kc_p, np * cout) : (int64_t)r * np * cout + kc_p] = v;
                    if (wt) wt[frag_t ? os_frag_index(c, kc_t, np * cin) : (int64_t)c * np * cin + kc_t] = v;
                }
            }
    }
    return CG3D_OK;
}
int cg3d_spconv_prep_weights_bf16(const float *W, uint16_t *Wb, int64_t slots, int32_t cin, int32_t cout, cg3d_stream_t s) {
    (void)s;
    for (int64_t t = 0; t < slots; t++)
        for (int32_t co = 0; co < cout; co++)
            for (int32_t ci = 0; ci < cin; ci++) {
                float v = os_bf16(W[(t * cin + ci) * cout + co]);
                uint32_t u; memcpy(&u, &v, 4);
                Wb[(t * cout + co) * cin + ci] = (uint16_t)(u >> 16);
            }
    return CG3D_OK;
}

/* Pair-list form (ME's in/out kernel maps): per offset k, Y[out] += X[in] W[k]. */
int cg3d_spconv_pairs_fwd(const float *X, const float *W, const int32_t *pin, const int32_t *pout, const int32_t *seg,
                          int64_t nseg, const float *bias, float *Y, int64_t n_out, int32_t cin, int32_t cout,
                          int32_t precision, int32_t accumulate, cg3d_stream_t s) {
    (void)s;
    if (n_out < 0 || nseg < 0 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    if (!accumulate)
        for (int64_t o = 0; o < n_out; o++)
            for (int32_t c = 0; c < cout; c++) Y[o * cout + c] = bias ? bias[c] : 0.f;
    /* segments of one offset never share an output row, segments of different offsets may:
       run the offsets one after the other, the pairs of an offset in parallel */
    for (int64_t g = 0; g < nseg; g++) {
        int32_t k = seg[g * 3], start = seg[g * 3 + 1], count = seg[g * 3 + 2];
        const float *w = W + (int64_t)k * cin * cout;
        const uint16_t *wb = (const uint16_t *)W + (int64_t)k * cin * cout;   /* precision 1: prepared [cout][cin] bf16 */
#pragma omp parallel for schedule(static)
        for (int32_t p = start; p < start + count; p++) {
            const float *x = X + (int64_t)pin[p] * cin;
            float *y = Y + (int64_t)pout[p] * cout;
            if (precision >= 1) {
                for (int32_t c = 0; c < cout; c++) {
                    float acc = 0.f;
                    for (int32_t a = 0; a < cin; a++)
                        acc += os_row(X, (int64_t)pin[p] * cin + a, precision) * os_bf16_bits(wb[(int64_t)c * cin + a]);
                    y[c] += acc;
                }
            } else {
                for (int32_t a = 0; a < cin; a++) {
                    float xa = x[a];
                    const float *wr = w + (int64_t)a * cout;
                    for (int32_t c = 0; c < cout; c++) y[c] += xa * wr[c];
                }
            }
        }
    }
    return CG3D_OK;
}
int cg3d_spconv_pairs_wgrad(const float *X, const float *dY, const int32_t *pin, const int32_t *pout,
                            const int32_t *seg, int64_t nseg, float *dW, int32_t K, int32_t cin, int32_t cout,
                            int32_t precision, cg3d_stream_t s) {
    (void)s;
    if (nseg < 0 || K < 1 || cin < 1 || cout < 1) return CG3D_ERR_ARG;
    const int accumulate = (precision & CG3D_WGRAD_ACCUMULATE) != 0;
    precision &= ~CG3D_WGRAD_ACCUMULATE;
    if (precision < 0 || precision > 3) return CG3D_ERR_ARG;
    if (!accumulate) memset(dW, 0, (size_t)K * cin * cout * sizeof(float));
    if (precision == 3) {
        /* split rows [hi | lo | hi]: dW = Xhi^T dYhi + Xlo^T dYhi + Xhi^T dYlo, each product of two bf16 values exact in fp32 */
        const uint16_t *Xs = (const uint16_t *)X, *Ds = (const uint16_t *)dY;
        const int32_t AB3 = 16;
        int32_t nab3 = (cin + AB3 - 1) / AB3;
#pragma omp parallel for schedule(dynamic, 1)
        for (int32_t ab = 0; ab < nab3; ab++) {
            int32_t a0 = ab * AB3, a1 = a0 + AB3 < cin ? a0 + AB3 : cin;
            for (int64_t g = 0; g < nseg; g++) {
                int32_t k = seg[g * 3], start = seg[g * 3 + 1], count = seg[g * 3 + 2];
                float *w = dW + (int64_t)k * cin * cout;
                for (int32_t p = start; p < start + count; p++) {
                    const uint16_t *xr = Xs + (int64_t)pin[p] * 3 * cin, *dr = Ds + (int64_t)pout[p] * 3 * cout;
                    for (int32_t a = a0; a < a1; a++) {
                        const float xh = os_bf16_bits(xr[a]), xl = os_bf16_bits(xr[cin + a]);
                        float *wr = w + (int64_t)a * cout;
                        for (int32_t c = 0; c < cout; c++) {
                            const float dh = os_bf16_bits(dr[c]), dl = os_bf16_bits(dr[cout + c]);
                            wr[c] += xh * dh + xl * dh + xh * dl;
                        }
                    }
                }
            }
        }
        return CG3D_OK;
    }
    const int32_t AB = 16;
    int32_t nab = (cin + AB - 1) / AB;
#pragma omp parallel for schedule(dynamic, 1)
    for (int32_t ab = 0; ab < nab; ab++) {
        int32_t a0 = ab * AB, a1 = a0 + AB < cin ? a0 + AB : cin;
        for (int64_t g = 0; g < nseg; g++) {
            int32_t k = seg[g * 3], start = seg[g * 3 + 1], count = seg[g * 3 + 2];
            float *w = dW + (int64_t)k * cin * cout;
            for (int32_t p = start; p < start + count; p++) {
                const float *d = dY + (int64_t)pout[p] * cout;
                for (int32_t a = a0; a < a1; a++) {
                    float xa = os_row(X, (int64_t)pin[p] * cin + a, precision);
                    float *wr = w + (int64_t)a * cout;
                    if (precision >= 1) {
                        for (int32_t c = 0; c < cout; c++) wr[c] += xa * os_row(dY, (int64_t)pout[p] * cout + c, precision);
                    } else { for (int32_t c = 0; c < cout; c++) wr[c] += xa * d[c]; }
                }
            }
        }
    }
    return CG3D_OK;
}
