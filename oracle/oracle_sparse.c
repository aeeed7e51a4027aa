/*
 * oracle_sparse.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the sparse-voxel engine the reference gets from MinkowskiEngine v0.5.4
 * (README.md:45; NOT vendored under /root/reference and not installable here, so its arithmetic
 * is restated from the published algorithm and anchored on the reference's own call sites --
 * SURVEY.md section 2.3).  PARITY UNPINNED against ME itself: the reference holds no golden
 * vectors for these operators.  The restatement is instead pinned by
 * independent dense references in tests/ (torch.nn.functional.conv3d on an occupancy grid,
 * explicit trilinear / pooling loops).
 *
 * Exports the same C-ABI as the HIP library (include/cagroup3d_hip.h) on HOST pointers.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Semantics fixed here (each is what the reference call sites rely on):
 *  - coordinate map: duplicates merged, representative = first occurrence, output rows in
 *    ascending representative order (cagroup3d.py:18-25; cagroup_roi_head.py:70-72 relies on
 *    "already unique coordinates keep their order").
 *  - stride map: floor(c / s) * s, de-duplicated (biresnet.py strided convs).
 *  - kernel map: nbr[k, o] = row of (o + offset_k) (ME kernel_map).
 *  - convolution: Y[o] = sum_k X[nbr[k,o]] W[k]  (ConvolutionForward gather-GEMM-scatter).
 *  - interpolation: 8-corner trilinear on the source lattice, absent corners contribute 0
 *    (features_at_coordinates, biresnet.py:182-197,376,389,394).
 *  - average pooling / quantise-average: mean over PRESENT inputs only.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/cagroup3d_hip.h"

int cg3d_is_device_library(void) { return 0; }
int cg3d_abi_version(void) { return 2; }
int cg3d_h2d_async(void *dst, const void *src, int64_t nbytes, cg3d_stream_t stream) {
    (void)stream;
    if (nbytes < 0 || (nbytes > 0 && (!dst || !src))) return CG3D_ERR_ARG;
    if (nbytes > 0) memcpy(dst, src, (size_t)nbytes);
    return CG3D_OK;
}

#define OS_EMPTY (~0ULL)

static inline int os_pack(int32_t b, int32_t x, int32_t y, int32_t z, uint64_t *key) {
    if (b < 0 || b >= CG3D_BATCH_LIMIT) return 0;
    if (x < -CG3D_COORD_LIMIT || x >= CG3D_COORD_LIMIT || y < -CG3D_COORD_LIMIT || y >= CG3D_COORD_LIMIT ||
        z < -CG3D_COORD_LIMIT || z >= CG3D_COORD_LIMIT)
        return 0;
    *key = ((uint64_t)b << 45) | ((uint64_t)(x + CG3D_COORD_LIMIT) << 30) |
           ((uint64_t)(y + CG3D_COORD_LIMIT) << 15) | (uint64_t)(z + CG3D_COORD_LIMIT);
    return 1;
}
static inline uint64_t os_hash(uint64_t k) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}
static inline int32_t os_floordiv(int32_t a, int32_t s) {
    int32_t q = a / s;
    if ((a % s != 0) && ((a < 0) != (s < 0))) q--;
    return q;
}
static inline int32_t os_lookup(const uint64_t *keys, const int32_t *vals, int64_t cap, uint64_t key) {
    uint64_t slot = os_hash(key) & (uint64_t)(cap - 1);
    for (;;) {
        uint64_t k = keys[slot];
        if (k == key) return vals[slot];
        if (k == OS_EMPTY) return -1;
        slot = (slot + 1) & (uint64_t)(cap - 1);
    }
}

int64_t cg3d_hash_capacity(int64_t n) {
    int64_t cap = 64;
    while (cap < 2 * n) cap <<= 1;
    return cap;
}
int64_t cg3d_coord_map_ws_bytes(int64_t n) { return (2 * n + n / 1024 + 64) * (int64_t)sizeof(int32_t); }

int cg3d_coord_map_build(const int32_t *coords, int64_t n, int32_t qstride, uint64_t *keys, int32_t *vals,
                         int64_t cap, void *ws, int32_t *out_coords, int32_t *unique_index,
                         int32_t *inverse, int32_t *n_out, cg3d_stream_t s) {
    (void)s; (void)ws;
    if (n < 0 || qstride < 1 || cap < 2 * n || (cap & (cap - 1))) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < cap; i++) keys[i] = OS_EMPTY;
    int32_t m = 0;
    for (int64_t i = 0; i < n; i++) {
        int32_t b = coords[i * 4], x = coords[i * 4 + 1], y = coords[i * 4 + 2], z = coords[i * 4 + 3];
        if (qstride > 1) {
            x = os_floordiv(x, qstride) * qstride;
            y = os_floordiv(y, qstride) * qstride;
            z = os_floordiv(z, qstride) * qstride;
        }
        uint64_t key;
        if (!os_pack(b, x, y, z, &key)) { n_out[0] = 0; n_out[1] = CG3D_ERR_RANGE; return CG3D_OK; }   /* as the device: status word */
        uint64_t slot = os_hash(key) & (uint64_t)(cap - 1);
        for (;;) {
            if (keys[slot] == key) { inverse[i] = vals[slot]; break; }
            if (keys[slot] == OS_EMPTY) {
                keys[slot] = key; vals[slot] = m;
                out_coords[m * 4] = b; out_coords[m * 4 + 1] = x; out_coords[m * 4 + 2] = y; out_coords[m * 4 + 3] = z;
                unique_index[m] = (int32_t)i; inverse[i] = m; m++;
                break;
            }
            slot = (slot + 1) & (uint64_t)(cap - 1);
        }
    }
    n_out[0] = m;
    n_out[1] = 0;
    return CG3D_OK;
}

/* Morton row order (the row order the host engine builds every inserted map in): order[i] = input row of the i-th row
 * in (batch, Morton(x,y,z)) order, ties (rows of one voxel) in input order. */
static uint64_t os_spread3(uint32_t v) {
    uint64_t x = v & 0x7fffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
typedef struct { uint64_t key; int32_t row; } os_mrow;
static int os_mrow_cmp(const void *a, const void *b) {
    const os_mrow *x = (const os_mrow *)a, *y = (const os_mrow *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->row < y->row ? -1 : (x->row > y->row ? 1 : 0);
}
int64_t cg3d_morton_order_ws_bytes(int64_t n) { return (n > 0 ? n : 1) * (int64_t)sizeof(os_mrow); }
int cg3d_morton_order(const int32_t *coords, int64_t n, int32_t *order, void *ws, cg3d_stream_t s) {
    (void)s;
    if (n < 0) return CG3D_ERR_ARG;
    os_mrow *rows = (os_mrow *)ws;
    for (int64_t i = 0; i < n; i++) {
        const int32_t b = coords[i * 4];
        const uint32_t ux = (uint32_t)(coords[i * 4 + 1] + CG3D_COORD_LIMIT), uy = (uint32_t)(coords[i * 4 + 2] + CG3D_COORD_LIMIT),
                       uz = (uint32_t)(coords[i * 4 + 3] + CG3D_COORD_LIMIT);
        uint64_t key = ~0ull;
        if ((uint32_t)b < (uint32_t)CG3D_BATCH_LIMIT && (ux | uy | uz) < (uint32_t)(2 * CG3D_COORD_LIMIT))
            key = ((uint64_t)b << 45) | (os_spread3(ux) << 2) | (os_spread3(uy) << 1) | os_spread3(uz);
        rows[i].key = key;
        rows[i].row = (int32_t)i;
    }
    qsort(rows, (size_t)n, sizeof(os_mrow), os_mrow_cmp);
    for (int64_t i = 0; i < n; i++) order[i] = rows[i].row;
    return CG3D_OK;
}

int cg3d_kernel_map(const int32_t *q, int64_t nq, const int32_t *off, int32_t K, const uint64_t *keys,
                    const int32_t *vals, int64_t cap, int32_t *nbr, cg3d_stream_t s) {
    (void)s;
    if (nq < 0 || K < 1) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++)
        for (int32_t k = 0; k < K; k++) {
            uint64_t key;
            int32_t r = -1;
            if (os_pack(q[i * 4], q[i * 4 + 1] + off[k * 3], q[i * 4 + 2] + off[k * 3 + 1],
                        q[i * 4 + 3] + off[k * 3 + 2], &key))
                r = os_lookup(keys, vals, cap, key);
            nbr[(int64_t)k * nq + i] = r;
        }
    return CG3D_OK;
}

int cg3d_kernel_map_self(const int32_t *q, int64_t n, const int32_t *off, int32_t K, const uint64_t *keys,
                         const int32_t *vals, int64_t cap, int32_t *nbr, cg3d_stream_t s) {
    if (!(K & 1)) return CG3D_ERR_ARG;
    return cg3d_kernel_map(q, n, off, K, keys, vals, cap, nbr, s);      /* the plain lookups: the definition of the result */
}

int cg3d_kernel_map_transpose(const int32_t *nbr, int32_t K, int64_t n_out, int64_t n_in, int32_t *nbrT, cg3d_stream_t s) {
    (void)s;
    if (K < 1 || n_out < 0 || n_in < 0) return CG3D_ERR_ARG;
    for (int64_t t = 0; t < (int64_t)K * n_in; t++) nbrT[t] = -1;
    for (int32_t k = 0; k < K; k++)
        for (int64_t o = 0; o < n_out; o++) {
            const int32_t i = nbr[(int64_t)k * n_out + o];
            if (i >= 0) nbrT[(int64_t)k * n_in + i] = (int32_t)o;
        }
    return CG3D_OK;
}

int cg3d_interp_map(const float *q, int64_t nq, int32_t ts, const uint64_t *keys, const int32_t *vals,
                    int64_t cap, int32_t *idx, float *w, cg3d_stream_t s) {
    (void)s;
    if (nq < 0 || ts < 1) return CG3D_ERR_ARG;
    const float fts = (float)ts;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) {
        int32_t b = (int32_t)q[i * 4];
        int32_t base[3]; float w0[3], w1[3];
        for (int d = 0; d < 3; d++) {
            float v = q[i * 4 + 1 + d];
            float fl = floorf(v / fts);
            float lo = fl * fts;
            float r = (v - lo) / fts;
            base[d] = (int32_t)lo; w0[d] = 1.0f - r; w1[d] = r;
        }
        for (int j = 0; j < 8; j++) {
            int dx = (j >> 2) & 1, dy = (j >> 1) & 1, dz = j & 1;
            float wt = ((dx ? w1[0] : w0[0]) * (dy ? w1[1] : w0[1])) * (dz ? w1[2] : w0[2]);
            uint64_t key; int32_t r = -1;
            if (os_pack(b, base[0] + dx * ts, base[1] + dy * ts, base[2] + dz * ts, &key))
                r = os_lookup(keys, vals, cap, key);
            idx[i * 8 + j] = r; w[i * 8 + j] = wt;
        }
    }
    return CG3D_OK;
}
int cg3d_interp_fwd(const float *F, const int32_t *idx, const float *w, float *out, int64_t nq, int32_t c,
                    cg3d_stream_t s) {
    (void)s;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nq; i++) {
        float *o = out + i * c;
        for (int32_t a = 0; a < c; a++) o[a] = 0.f;
        for (int j = 0; j < 8; j++) {
            int32_t r = idx[i * 8 + j];
            if (r < 0) continue;
            float wt = w[i * 8 + j];
            const float *f = F + (int64_t)r * c;
            for (int32_t a = 0; a < c; a++) o[a] += wt * f[a];
        }
    }
    return CG3D_OK;
}
int cg3d_interp_bwd(const float *dout, const int32_t *idx, const float *w, float *dF, int64_t nq, int32_t c,
                    cg3d_stream_t s) {
    (void)s;
    for (int64_t i = 0; i < nq; i++)
        for (int j = 0; j < 8; j++) {
            int32_t r = idx[i * 8 + j];
            if (r < 0) continue;
            float wt = w[i * 8 + j];
            float *f = dF + (int64_t)r * c;
            const float *d = dout + i * c;
            for (int32_t a = 0; a < c; a++) f[a] += wt * d[a];
        }
    return CG3D_OK;
}

int cg3d_pool_map(const int32_t *in, int64_t n_in, int32_t out_stride, int32_t half_extent,
                  const uint64_t *keys, const int32_t *vals, int64_t cap, int32_t *pmap, cg3d_stream_t s) {
    (void)s;
    if (n_in < 0 || out_stride < 1 || half_extent < 0) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_in; i++) {
        int32_t b = in[i * 4], c[3] = {in[i * 4 + 1], in[i * 4 + 2], in[i * 4 + 3]};
        int32_t base[3];
        for (int d = 0; d < 3; d++) base[d] = os_floordiv(c[d], out_stride);
        for (int j = 0; j < 27; j++) {
            int32_t dd[3] = {j / 9 - 1, (j / 3) % 3 - 1, j % 3 - 1};
            int32_t o[3]; int ok = 1;
            for (int d = 0; d < 3; d++) {
                o[d] = (base[d] + dd[d]) * out_stride;
                int32_t diff = o[d] - c[d];
                if (diff < 0) diff = -diff;
                if (diff > half_extent) ok = 0;
            }
            int32_t r = -1; uint64_t key;
            if (ok && os_pack(b, o[0], o[1], o[2], &key)) r = os_lookup(keys, vals, cap, key);
            pmap[(int64_t)j * n_in + i] = r;
        }
    }
    return CG3D_OK;
}
int cg3d_scatter_mean_fwd(const float *F, const int32_t *map, int32_t J, float *out, float *cnt, int64_t n_in,
                          int64_t n_out, int32_t c, cg3d_stream_t s) {
    (void)s;
    memset(out, 0, (size_t)(n_out * c) * sizeof(float));
    memset(cnt, 0, (size_t)n_out * sizeof(float));
    for (int32_t j = 0; j < J; j++)
        for (int64_t i = 0; i < n_in; i++) {
            int32_t m = map[(int64_t)j * n_in + i];
            if (m < 0) continue;
            cnt[m] += 1.0f;
            float *o = out + (int64_t)m * c;
            const float *f = F + i * c;
            for (int32_t a = 0; a < c; a++) o[a] += f[a];
        }
    for (int64_t m = 0; m < n_out; m++)
        if (cnt[m] > 0.f) {
            float *o = out + m * c;
            for (int32_t a = 0; a < c; a++) o[a] = o[a] / cnt[m];
        }
    return CG3D_OK;
}
int cg3d_scatter_mean_bwd(const float *dout, const float *cnt, const int32_t *map, int32_t J, float *dF,
                          int64_t n_in, int64_t n_out, int32_t c, cg3d_stream_t s) {
    (void)s; (void)n_out;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_in; i++) {
        float *f = dF + i * c;
        for (int32_t a = 0; a < c; a++) f[a] = 0.f;
        for (int32_t j = 0; j < J; j++) {
            int32_t m = map[(int64_t)j * n_in + i];
            if (m < 0) continue;
            const float *d = dout + (int64_t)m * c;
            float inv = cnt[m];
            for (int32_t a = 0; a < c; a++) f[a] += d[a] / inv;
        }
    }
    return CG3D_OK;
}

/* ---------------------------------------------------------------- pair-compacted kernel maps */
int64_t cg3d_pairs_ws_bytes(int64_t total) { return (total + total / 1024 + 64) * (int64_t)sizeof(int32_t); }
int cg3d_pairs_count(const int32_t *nbr, int32_t K, int64_t n_out, const int32_t *row_bounds, int32_t G, void *ws,
                     int32_t *pair_off, cg3d_stream_t s) {
    (void)s;
    if (K < 1 || n_out < 0 || G < 1) return CG3D_ERR_ARG;
    int32_t *pos = (int32_t *)ws;
    int32_t run = 0;
    for (int32_t k = 0; k < K; k++)
        for (int64_t o = 0; o < n_out; o++) {
            int64_t t = (int64_t)k * n_out + o;
            pos[t] = run;
            if (nbr[t] >= 0) run++;
        }
    for (int32_t k = 0; k < K; k++)
        for (int32_t g = 0; g < G; g++) {
            int64_t r = row_bounds ? row_bounds[g] : 0;
            int64_t t = (int64_t)k * n_out + r;
            pair_off[k * G + g] = (t < (int64_t)K * n_out) ? pos[t] : run;
        }
    pair_off[K * G] = run;
    return CG3D_OK;
}
int cg3d_pairs_fill(const int32_t *nbr, int32_t K, int64_t n_out, const void *ws, int32_t *pair_in, int32_t *pair_out,
                    cg3d_stream_t s) {
    (void)s;
    const int32_t *pos = (const int32_t *)ws;
    for (int64_t t = 0; t < (int64_t)K * n_out; t++)
        if (nbr[t] >= 0) { pair_in[pos[t]] = nbr[t]; pair_out[pos[t]] = (int32_t)(t % n_out); }
    return CG3D_OK;
}

/* ---------------------------------------------------------------- row gather / scatter-add */
int cg3d_gather_rows(const float *F, const int32_t *idx, float *out, int64_t n, int32_t c, cg3d_stream_t s) {
    (void)s;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) memcpy(out + i * c, F + (int64_t)idx[i] * c, (size_t)c * sizeof(float));
    return CG3D_OK;
}
int cg3d_scatter_add_rows(const float *dout, const int32_t *idx, float *dF, int64_t n, int32_t c, cg3d_stream_t s) {
    (void)s;
    for (int64_t i = 0; i < n; i++) {
        float *f = dF + (int64_t)idx[i] * c;
        const float *d = dout + i * c;
        for (int32_t a = 0; a < c; a++) f[a] += d[a];
    }
    return CG3D_OK;
}

/* ---------------------------------------------------------------- grouped BatchNorm (+res) (+act)
 * plain restatement of torch.nn.BatchNorm1d -> ReLU/ELU -> residual add as the reference chains them
 * (biresnet.py:33-50; cagroup_head.py:117-127). */
static inline float os_act_fwd(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    return v;
}
static inline float os_act_bwd(float y, int act) {
    if (act == 1) return y > 0.f ? 1.f : 0.f;
    if (act == 2) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}
/* fp32 -> bf16 bits, round-to-nearest-even (the copy the next convolution gathers from; cg3d_to_bf16's rounding) */
static inline uint16_t os_bf16_rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
/* CG3D_BN_STORE_BF16 (a bit of the `act` argument): the row matrices of the call are stored as bf16 (uint16) instead of
 * float32 -- element o of such a matrix, read / written through these two */
static inline float os_ld(const void *base, int64_t o, int s16) {
    if (!s16) return ((const float *)base)[o];
    uint32_t u = (uint32_t)((const uint16_t *)base)[o] << 16; float f; memcpy(&f, &u, 4); return f;
}
static inline void os_st(void *base, int64_t o, float v, int s16) {
    if (s16) ((uint16_t *)base)[o] = os_bf16_rne(v); else ((float *)base)[o] = v;
}
static void os_bn_sums(int bwd, const float *A, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk,
                       int32_t G, int32_t c, const float *mean, const float *var, float eps, int32_t act, double *sums) {
    const int s16 = (act & CG3D_BN_STORE_BF16) != 0;
    act &= ~CG3D_BN_STORE_BF16;
    const size_t ns = 2 * (size_t)G * c;
    memset(sums, 0, sizeof(double) * ns);
#pragma omp parallel
    {
        double *loc = (double *)calloc(ns, sizeof(double));
#pragma omp for schedule(static)
        for (int64_t k = 0; k < nchunk; k++) {
            int32_t g = chunks[k * 3], r0 = chunks[k * 3 + 1], nr = chunks[k * 3 + 2];
            for (int32_t r = r0; r < r0 + nr; r++)
                for (int32_t a = 0; a < c; a++) {
                    int64_t o = (int64_t)r * c + a, p = (int64_t)g * c + a;
                    if (!bwd) {
                        double v = os_ld(A, o, s16);
                        loc[p] += v;
                        loc[(int64_t)(G + g) * c + a] += v * v;
                    } else {
                        float d = os_ld(A, o, s16) * (act ? os_act_bwd(os_ld(Y, o, s16), act) : 1.f);
                        loc[p] += d;
                        loc[(int64_t)(G + g) * c + a] += (double)(d * (os_ld(X, o, s16) - mean[p]) * (1.0f / sqrtf(var[p] + eps)));
                    }
                }
        }
#pragma omp critical
        for (size_t i = 0; i < ns; i++) sums[i] += loc[i];
        free(loc);
    }
}
/* The statistics tables are float32 [CG3D_BN_SLOTS][2][G][C] (include/cagroup3d_hip.h); this restatement adds everything to
 * slot 0 and the consumers add the slots up.
 * sums[0][g][:] += sum x, sums[1][g][:] += sum x^2 over the rows of group g (the caller zero-fills `sums`) */
int cg3d_bn_sums(const float *X, const int32_t *chunks, int64_t nchunk, int32_t G, int32_t c, float *sums, cg3d_stream_t s) {
    (void)s;
    if (nchunk < 0 || G < 1 || c < 1 || !sums) return CG3D_ERR_ARG;
    double *d = (double *)malloc(sizeof(double) * 2 * (size_t)G * c);
    os_bn_sums(0, X, NULL, NULL, chunks, nchunk, G, c, NULL, NULL, 0.f, 0, d);
    for (int64_t i = 0; i < 2 * (int64_t)G * c; i++) sums[i] += (float)d[i];
    free(d);
    return CG3D_OK;
}
int cg3d_bn_bwd_sums(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk, int32_t G, int32_t c,
                     const float *mean, const float *var, float eps, int32_t act, float *dsums, cg3d_stream_t s) {
    (void)s;
    if (nchunk < 0 || G < 1 || c < 1 || !dsums) return CG3D_ERR_ARG;
    double *d = (double *)malloc(sizeof(double) * 2 * (size_t)G * c);
    os_bn_sums(1, dY, X, Y, chunks, nchunk, G, c, mean, var, eps, act, d);
    for (int64_t i = 0; i < 2 * (int64_t)G * c; i++) dsums[i] += (float)d[i];
    free(d);
    return CG3D_OK;
}
int cg3d_bn_apply(const float *X, const float *R, const int32_t *chunks, int64_t nchunk, int32_t c, const float *mean,
                  const float *var, float eps, const float *gamma, const float *beta, int32_t act, float *Y,
                  uint16_t *Y16, cg3d_stream_t s) {
    (void)s;
    const int s16 = (act & CG3D_BN_STORE_BF16) != 0;
    act &= ~CG3D_BN_STORE_BF16;
    if (s16 && Y16) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nchunk; k++) {
        int32_t g = chunks[k * 3], r0 = chunks[k * 3 + 1], nr = chunks[k * 3 + 2];
        for (int32_t r = r0; r < r0 + nr; r++)
            for (int32_t a = 0; a < c; a++) {
                int64_t o = (int64_t)r * c + a, p = (int64_t)g * c + a;
                float y = (os_ld(X, o, s16) - mean[p]) * (1.0f / sqrtf(var[p] + eps)) * gamma[p] + beta[p];
                if (R) y += os_ld(R, o, s16);
                y = os_act_fwd(y, act);
                os_st(Y, o, y, s16);
                if (Y16) Y16[o] = os_bf16_rne(y);
            }
    }
    return CG3D_OK;
}
/* cg3d_bn_apply with mean / variance derived from the statistics table (fp64), written to mean / var, running statistics
 * updated like nn.BatchNorm1d */
int cg3d_bn_apply_sums(const float *X, const float *R, const int32_t *chunks, int64_t nchunk, int32_t G, int32_t c,
                       const float *sums, const float *group_n, float eps, const float *gamma, const float *beta, int32_t act,
                       float *Y, uint16_t *Y16, float *mean, float *var, float *running_mean, float *running_var,
                       int64_t *num_batches_tracked, float momentum, cg3d_stream_t s) {
    if (nchunk < 0 || G < 1 || c < 1 || !sums || !group_n || !mean || !var) return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    for (int32_t g = 0; g < G; g++) {
        const double n = group_n[g] > 0.f ? (double)group_n[g] : 1.0;
        for (int32_t a = 0; a < c; a++) {
            double s0 = 0, s1 = 0;
            for (int sl = 0; sl < CG3D_BN_SLOTS; sl++) {
                s0 += sums[((int64_t)(sl * 2) * G + g) * c + a];
                s1 += sums[((int64_t)(sl * 2 + 1) * G + g) * c + a];
            }
            const double m = s0 / n, v = s1 / n - m * m;
            mean[(int64_t)g * c + a] = (float)m;
            var[(int64_t)g * c + a] = (float)(v > 0 ? v : 0);
            if (running_mean && running_var) {
                const float unb = (float)(n / (n > 1 ? n - 1 : 1));
                float *rm = running_mean + (int64_t)g * c + a, *rv = running_var + (int64_t)g * c + a;
                *rm = (1.f - momentum) * *rm + momentum * (float)m;
                *rv = (1.f - momentum) * *rv + momentum * ((float)(v > 0 ? v : 0) * unb);
            }
        }
        if (num_batches_tracked) num_batches_tracked[g] += 1;
    }
    return cg3d_bn_apply(X, R, chunks, nchunk, c, mean, var, eps, gamma, beta, act, Y, Y16, s);
}
int cg3d_bn_bwd_apply(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk, int32_t c,
                      const float *mean, const float *var, float eps, const float *gamma, const float *dbeta,
                      const float *dgamma, const float *group_n, int32_t act, int32_t use_batch, float *dX,
                      uint16_t *dX16, float *dR, cg3d_stream_t s);
int cg3d_bn_bwd_apply_sums(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk, int32_t G,
                           int32_t c, const float *mean, const float *var, float eps, const float *gamma, const float *dsums,
                           const float *group_n, int32_t act, int32_t use_batch, float *dX, uint16_t *dX16, float *dR,
                           float *dbeta, float *dgamma, cg3d_stream_t s) {
    if (nchunk < 0 || G < 1 || c < 1 || !dsums || !dbeta || !dgamma) return CG3D_ERR_ARG;
    if (nchunk == 0) return CG3D_OK;
    for (int64_t i = 0; i < (int64_t)G * c; i++) {
        double b = 0, gm = 0;
        for (int sl = 0; sl < CG3D_BN_SLOTS; sl++) {
            b += dsums[(int64_t)(sl * 2) * G * c + i];
            gm += dsums[(int64_t)(sl * 2 + 1) * G * c + i];
        }
        dbeta[i] = (float)b; dgamma[i] = (float)gm;
    }
    return cg3d_bn_bwd_apply(dY, X, Y, chunks, nchunk, c, mean, var, eps, gamma, dbeta, dgamma, group_n, act, use_batch, dX, dX16, dR, s);
}
int cg3d_bn_bwd_apply(const float *dY, const float *X, const float *Y, const int32_t *chunks, int64_t nchunk, int32_t c,
                      const float *mean, const float *var, float eps, const float *gamma, const float *dbeta,
                      const float *dgamma, const float *group_n, int32_t act, int32_t use_batch, float *dX,
                      uint16_t *dX16, float *dR, cg3d_stream_t s) {
    (void)s;
    const int s16 = (act & CG3D_BN_STORE_BF16) != 0;
    act &= ~CG3D_BN_STORE_BF16;
    if (s16 && dX16) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nchunk; k++) {
        int32_t g = chunks[k * 3], r0 = chunks[k * 3 + 1], nr = chunks[k * 3 + 2];
        float inv_n = use_batch ? 1.f / group_n[g] : 0.f;
        for (int32_t r = r0; r < r0 + nr; r++)
            for (int32_t a = 0; a < c; a++) {
                int64_t o = (int64_t)r * c + a, p = (int64_t)g * c + a;
                float d = os_ld(dY, o, s16) * (act ? os_act_bwd(os_ld(Y, o, s16), act) : 1.f);
                if (dR) os_st(dR, o, d, s16);
                float is = 1.0f / sqrtf(var[p] + eps);
                float xh = (os_ld(X, o, s16) - mean[p]) * is;
                float dx = gamma[p] * is * (d - (dbeta[p] + xh * dgamma[p]) * inv_n);
                os_st(dX, o, dx, s16);
                if (dX16) dX16[o] = os_bf16_rne(dx);
            }
    }
    return CG3D_OK;
}
