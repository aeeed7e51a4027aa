/*
 * oracle_geom.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's box-geometry / NMS / kNN / sort_vertices algorithms,
 * exporting the same C-ABI as the HIP library (include/cagroup3d_hip.h) on HOST pointers.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Bit-exactness contract with the HIP kernels: every float expression below is evaluated in
 * the same order with plain IEEE fp32 add/mul/div (compile with -ffp-contract=off; no
 * fast-math).  sin/cos/atan2 are NOT taken from libm (device and host libms differ in ULPs):
 * both sides use the same Cephes-style single-precision polynomials (og_sinf/og_cosf/og_atan2f),
 * so IoU values, suppression masks, keep lists, kNN indices and vertex orders compare bit-exact.
 * Parity pin: the rotated BEV IoU here is checked against the reference's own
 * pcdet/ops/iou3d_nms/src/iou3d_cpu.cpp built under oracle/_ref (libm trig) to 1e-5 absolute.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/cagroup3d_hip.h"

/* ---------------------------------------------------------------- deterministic fp32 trig */
static const float OG_FOPI = 1.27323954473516f; /* 4/pi */
static const float OG_DP1 = 0.78515625f, OG_DP2 = 2.4187564849853515625e-4f,
                   OG_DP3 = 3.77489497744594108e-8f;

static inline float og_poly_sin(float x, float z) {
    float p = -1.9515295891E-4f;
    p = p * z + 8.3321608736E-3f;
    p = p * z + -1.6666654611E-1f;
    return p * z * x + x;
}
static inline float og_poly_cos(float z) {
    float p = 2.443315711809948E-005f;
    p = p * z + -1.388731625493765E-003f;
    p = p * z + 4.166664568298827E-002f;
    return p * z * z - 0.5f * z + 1.0f;
}
float og_sinf(float x) {
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    int j = (int)(OG_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sign = -sign; j -= 4; }
    x = ((x - y * OG_DP1) - y * OG_DP2) - y * OG_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? og_poly_cos(z) : og_poly_sin(x, z);
    return sign * r;
}
float og_cosf(float x) {
    float sign = 1.0f;
    if (x < 0.0f) x = -x;
    int j = (int)(OG_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sign = -sign; j -= 4; }
    if (j > 1) sign = -sign;
    x = ((x - y * OG_DP1) - y * OG_DP2) - y * OG_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? og_poly_sin(x, z) : og_poly_cos(z);
    return sign * r;
}
static inline float og_atanf(float x) {
    float sign = 1.0f, y;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    float p = 8.05374449538e-2f;
    p = p * z - 1.38776856032E-1f;
    p = p * z + 1.99777106478E-1f;
    p = p * z - 3.33329491539E-1f;
    y += p * z * x + x;
    return sign * y;
}
float og_atan2f(float y, float x) {
    const float PIF = 3.141592653589793f, PIO2F = 1.5707963267948966f;
    int code = 0;
    if (x < 0.0f) code = 2;
    if (y < 0.0f) code |= 1;
    if (x == 0.0f) {
        if (code & 1) return -PIO2F;
        if (y == 0.0f) return 0.0f;
        return PIO2F;
    }
    if (y == 0.0f) return (code & 2) ? PIF : 0.0f;
    float w = (code == 2) ? PIF : ((code == 3) ? -PIF : 0.0f);
    return w + og_atanf(y / x);
}

/* ---------------------------------------------------------------- rotated BEV overlap
 * follows pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:104-225 (== iou3d_cpu.cpp:38-229). */
#define OG_EPS 1e-8f
typedef struct { float x, y; } og_pt;

static inline float og_cross3(og_pt p1, og_pt p2, og_pt p0) { /* .cu:32-34 */
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
static inline float og_cross2(og_pt a, og_pt b) { return a.x * b.y - a.y * b.x; } /* .cu:28-30 */
static inline float og_min(float a, float b) { return a > b ? b : a; } /* cpu.cpp:30-36 */
static inline float og_max(float a, float b) { return a > b ? a : b; }

static inline int og_rect_cross(og_pt p1, og_pt p2, og_pt q1, og_pt q2) { /* .cu:36-42 */
    return og_min(p1.x, p2.x) <= og_max(q1.x, q2.x) && og_min(q1.x, q2.x) <= og_max(p1.x, p2.x) &&
           og_min(p1.y, p2.y) <= og_max(q1.y, q2.y) && og_min(q1.y, q2.y) <= og_max(p1.y, p2.y);
}
static inline int og_in_box2d(const float *box, og_pt p) { /* .cu:44-55 */
    const float MARGIN = 1e-2f;
    float cx = box[0], cy = box[1];
    float ac = og_cosf(-box[6]), as = og_sinf(-box[6]);
    float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
    float ry = (p.x - cx) * as + (p.y - cy) * ac;
    return (fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN);
}
static inline int og_intersection(og_pt p1, og_pt p0, og_pt q1, og_pt q0, og_pt *ans) { /* .cu:57-87 */
    if (!og_rect_cross(p0, p1, q0, q1)) return 0;
    float s1 = og_cross3(q0, p1, p0);
    float s2 = og_cross3(p1, q1, p0);
    float s3 = og_cross3(p0, q1, q0);
    float s4 = og_cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
    float s5 = og_cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > OG_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return 1;
}
static inline og_pt og_rot(og_pt c, float ac, float as, og_pt p) { /* .cu:89-93 */
    og_pt r;
    r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    return r;
}

float og_box_overlap(const float *a, const float *b) { /* .cu:99-225 */
    float a_ang = a[6], b_ang = b[6];
    float adx = a[3] / 2, bdx = b[3] / 2, ady = a[4] / 2, bdy = b[4] / 2;
    float ax1 = a[0] - adx, ay1 = a[1] - ady, ax2 = a[0] + adx, ay2 = a[1] + ady;
    float bx1 = b[0] - bdx, by1 = b[1] - bdy, bx2 = b[0] + bdx, by2 = b[1] + bdy;
    og_pt ca = {a[0], a[1]}, cb = {b[0], b[1]};
    og_pt A[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}, {0, 0}};
    og_pt B[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}, {0, 0}};
    float aco = og_cosf(a_ang), asi = og_sinf(a_ang), bco = og_cosf(b_ang), bsi = og_sinf(b_ang);
    for (int k = 0; k < 4; k++) { A[k] = og_rot(ca, aco, asi, A[k]); B[k] = og_rot(cb, bco, bsi, B[k]); }
    A[4] = A[0]; B[4] = B[0];

    og_pt cp[16], ctr = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            og_pt t;
            if (og_intersection(A[i + 1], A[i], B[j + 1], B[j], &t)) {
                cp[cnt] = t; ctr.x = ctr.x + t.x; ctr.y = ctr.y + t.y; cnt++;
            }
        }
    for (int k = 0; k < 4; k++) {
        if (og_in_box2d(a, B[k])) { ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y; cp[cnt++] = B[k]; }
        if (og_in_box2d(b, A[k])) { ctr.x = ctr.x + A[k].x; ctr.y = ctr.y + A[k].y; cp[cnt++] = A[k]; }
    }
    ctr.x /= cnt; ctr.y /= cnt; /* cnt == 0 -> NaN centre, empty loops, area 0 (as the reference) */

    /* bubble sort by polar angle around the centre (.cu:95-97,198-208); angles cached per point
     * (atan2 of the same operands gives the same value, so caching does not change the order) */
    float ang[16];
    for (int i = 0; i < cnt; i++) ang[i] = og_atan2f(cp[i].y - ctr.y, cp[i].x - ctr.x);
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++)
            if (ang[i] > ang[i + 1]) {
                og_pt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
                float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; k++) {
        og_pt u = {cp[k].x - cp[0].x, cp[k].y - cp[0].y};
        og_pt v = {cp[k + 1].x - cp[0].x, cp[k + 1].y - cp[0].y};
        area += og_cross2(u, v);
    }
    return fabsf(area) / 2.0f;
}

float og_iou_bev(const float *a, const float *b) { /* .cu:227-234 */
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float so = og_box_overlap(a, b);
    return so / fmaxf(sa + sb - so, OG_EPS);
}
float og_iou_normal(const float *a, const float *b) { /* .cu:314-325 */
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    float inter = width * height;
    float Sa = a[3] * a[4], Sb = b[3] * b[4];
    return inter / fmaxf(Sa + Sb - inter, OG_EPS);
}

int cg3d_boxes_overlap_bev(const float *A, int64_t na, const float *B, int64_t nb, float *out,
                           cg3d_stream_t s) { /* .cu:236-249 */
    (void)s;
    if (na < 0 || nb < 0) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < na; i++)
        for (int64_t j = 0; j < nb; j++) out[i * nb + j] = og_box_overlap(A + i * 7, B + j * 7);
    return CG3D_OK;
}
int cg3d_boxes_iou_bev(const float *A, int64_t na, const float *B, int64_t nb, float *out,
                       cg3d_stream_t s) { /* .cu:251-265 */
    (void)s;
    if (na < 0 || nb < 0) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < na; i++)
        for (int64_t j = 0; j < nb; j++) out[i * nb + j] = og_iou_bev(A + i * 7, B + j * 7);
    return CG3D_OK;
}

int cg3d_boxes_iou_bev_cpu(const float *A, int64_t na, const float *B, int64_t nb, float *out) { /* iou3d_cpu.cpp:232-252 */
    return cg3d_boxes_iou_bev(A, na, B, nb, out, (cg3d_stream_t)0);
}

/* mask tiles (.cu:267-311 / 328-372) + greedy scan (iou3d_nms.cpp:117-132 / 167-182) */
static void og_nms_one(const float *boxes, int64_t n, float thr, int rotated, uint64_t *mask,
                       int64_t *keep, int32_t *num_keep) {
    int64_t cb = (n + 63) / 64;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        for (int64_t c = 0; c < cb; c++) {
            uint64_t t = 0;
            int64_t csz = n - c * 64 < 64 ? n - c * 64 : 64;
            int64_t start = (i / 64 == c) ? (i % 64) + 1 : 0;
            for (int64_t j = start; j < csz; j++) {
                const float *bj = boxes + (c * 64 + j) * 7;
                float iou = rotated ? og_iou_bev(boxes + i * 7, bj) : og_iou_normal(boxes + i * 7, bj);
                if (iou > thr) t |= 1ULL << j;
            }
            mask[i * cb + c] = t;
        }
    }
    uint64_t *remv = (uint64_t *)calloc((size_t)(cb > 0 ? cb : 1), sizeof(uint64_t));
    int32_t nk = 0;
    for (int64_t i = 0; i < n; i++) {
        int64_t nb = i / 64, ib = i % 64;
        if (!(remv[nb] & (1ULL << ib))) {
            keep[nk++] = i;
            const uint64_t *p = mask + i * cb;
            for (int64_t j = nb; j < cb; j++) remv[j] |= p[j];
        }
    }
    free(remv);
    *num_keep = nk;
}
int cg3d_nms(const float *boxes, int64_t n, float thr, int32_t rotated, uint64_t *mask_ws,
             int64_t *keep, int32_t *num_keep, cg3d_stream_t s) {
    (void)s;
    if (n < 0) return CG3D_ERR_ARG;
    og_nms_one(boxes, n, thr, rotated, mask_ws, keep, num_keep);
    return CG3D_OK;
}
/* the reference's literal forms (iou3d_nms.h:9-12): keep list handed back, count returned */
int64_t cg3d_nms_gpu_ws_bytes(int64_t n) {
    if (n < 0) n = 0;
    return n * ((n + 63) / 64) * 8 + n * 8 + 16;
}
static int og_nms_host_keep(const float *boxes, int64_t n, int64_t *keep_host, float thr, int rotated, void *ws) {
    if (n < 0 || (n > 0 && (!boxes || !keep_host)) || !ws) return CG3D_ERR_ARG;
    if (n == 0) return 0;
    int32_t nk = 0;
    og_nms_one(boxes, n, thr, rotated, (uint64_t *)ws, keep_host, &nk);
    return (int)nk;
}
int cg3d_nms_gpu(const float *boxes, int64_t n, int64_t *keep_host, float thr, void *ws, cg3d_stream_t s) {
    (void)s;
    return og_nms_host_keep(boxes, n, keep_host, thr, 1, ws);
}
int cg3d_nms_normal_gpu(const float *boxes, int64_t n, int64_t *keep_host, float thr, void *ws, cg3d_stream_t s) {
    (void)s;
    return og_nms_host_keep(boxes, n, keep_host, thr, 0, ws);
}
int cg3d_nms_batched(const float *boxes, const int64_t *seg_off, const int64_t *mask_off, int32_t nseg,
                     int64_t max_seg, float thr, int32_t rotated, uint64_t *mask_ws, int64_t *keep,
                     int32_t *num_keep, cg3d_stream_t s) {
    (void)s; (void)max_seg;
    for (int32_t g = 0; g < nseg; g++) {
        int64_t o = seg_off[g], n = seg_off[g + 1] - o;
        og_nms_one(boxes + o * 7, n, thr, rotated, mask_ws + mask_off[g], keep + o, num_keep + g);
    }
    return CG3D_OK;
}

/* ---------------------------------------------------------------- kNN, knn_cuda.cu:9-94 */
static void og_reheap(float *d, int *ix, int k) { /* knn_cuda.cu:25-40 */
    int root = 0, child = 1;
    while (child < k) {
        if (child + 1 < k && d[child + 1] > d[child]) child++;
        if (d[root] > d[child]) return;
        float td = d[root]; d[root] = d[child]; d[child] = td;
        int ti = ix[root]; ix[root] = ix[child]; ix[child] = ti;
        root = child; child = root * 2 + 1;
    }
}
int64_t cg3d_knn_ws_bytes(int32_t b, int32_t n, int32_t m, int32_t k) { (void)b; (void)n; (void)m; (void)k; return 0; }
int cg3d_knn(int32_t b, int32_t n, int32_t m, int32_t k, const float *xyz, const float *new_xyz,
             int32_t *idx, float *dist2, void *ws, cg3d_stream_t s) {
    (void)s; (void)ws;
    if (k < 1 || k > 100 || b < 0 || n < 0 || m < 0) return CG3D_ERR_ARG;
#pragma omp parallel for collapse(2) schedule(static)
    for (int32_t bi = 0; bi < b; bi++)
        for (int32_t q = 0; q < m; q++) {
            const float *p = new_xyz + ((int64_t)bi * m + q) * 3;
            const float *X = xyz + (int64_t)bi * n * 3;
            float bd[100]; int bx[100];
            for (int i = 0; i < k; i++) { bd[i] = 1e10f; bx[i] = 0; } /* .cu:74-77 */
            for (int i = 0; i < n; i++) {
                float x = X[i * 3 + 0], y = X[i * 3 + 1], z = X[i * 3 + 2];
                float d2 = (p[0] - x) * (p[0] - x) + (p[1] - y) * (p[1] - y) + (p[2] - z) * (p[2] - z);
                if (d2 < bd[0]) { bd[0] = d2; bx[0] = i; og_reheap(bd, bx, k); } /* strict <, .cu:83 */
            }
            for (int i = k - 1; i > 0; i--) { /* heap_sort, .cu:43-52 */
                float td = bd[0]; bd[0] = bd[i]; bd[i] = td;
                int ti = bx[0]; bx[0] = bx[i]; bx[i] = ti;
                og_reheap(bd, bx, i);
            }
            int64_t o = ((int64_t)bi * m + q) * k;
            for (int i = 0; i < k; i++) { idx[o + i] = bx[i]; dist2[o + i] = bd[i]; }
        }
    return CG3D_OK;
}

/* ---------------------------------------------------------------- ball query
 * follows pcdet/ops/pointnet2/pointnet2_batch/src/ball_query_gpu.cu:15-52 literally (one query = one loop over the
 * reference points in index order); a query without a hit keeps the zeros its caller initialised (here: written). */
int cg3d_ball_query(int32_t b, int32_t n, int32_t m, float radius, int32_t nsample, const float *new_xyz, const float *xyz,
                    int32_t *idx, cg3d_stream_t s) {
    (void)s;
    if (b < 0 || n < 0 || m < 0 || nsample < 1) return CG3D_ERR_ARG;
    const float radius2 = radius * radius;                      /* .cu:28 */
#pragma omp parallel for collapse(2) schedule(static)
    for (int32_t bi = 0; bi < b; bi++)
        for (int32_t q = 0; q < m; q++) {
            const float *p = new_xyz + ((int64_t)bi * m + q) * 3;
            const float *X = xyz + (int64_t)bi * n * 3;
            int32_t *out = idx + ((int64_t)bi * m + q) * nsample;
            for (int l = 0; l < nsample; l++) out[l] = 0;
            int cnt = 0;
            for (int k = 0; k < n; ++k) {                       /* .cu:34-49 */
                float x = X[k * 3 + 0], y = X[k * 3 + 1], z = X[k * 3 + 2];
                float d2 = (p[0] - x) * (p[0] - x) + (p[1] - y) * (p[1] - y) + (p[2] - z) * (p[2] - z);
                if (d2 < radius2) {
                    if (cnt == 0) for (int l = 0; l < nsample; ++l) out[l] = k;
                    out[cnt] = k;
                    ++cnt;
                    if (cnt >= nsample) break;
                }
            }
        }
    return CG3D_OK;
}

/* ---------------------------------------------------------------- sort_vertices
 * follows pcdet/ops/rotated_iou/cuda_op/sort_vert_kernel.cu:15-134 */
/* EPSILON is a double literal in the reference (:8), so comparisons against it and the `+ EPSILON`
 * in the norms promote to double; restated with the same promotions. */
#define OG_SV_EPS 1e-8
static int og_cmp_vert(float x1, float y1, float x2, float y2) { /* :15-40 */
    if ((double)fabsf(x1 - x2) < OG_SV_EPS && (double)fabsf(y2 - y1) < OG_SV_EPS) return 0;
    if (y1 > 0 && y2 < 0) return 1;
    if (y1 < 0 && y2 > 0) return 0;
    float n1 = (float)((double)(x1 * x1 + y1 * y1) + OG_SV_EPS);
    float n2 = (float)((double)(x2 * x2 + y2 * y2) + OG_SV_EPS);
    if (y1 > 0 && y2 > 0) return ((double)(fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2) > OG_SV_EPS) ? 1 : 0;
    if (y1 < 0 && y2 < 0) return ((double)(fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2) < OG_SV_EPS) ? 1 : 0;
    return 0; /* the reference falls off the end here (UB); we define it as false */
}
int cg3d_sort_vertices(int32_t b, int32_t n, int32_t m, const float *vertices, const uint8_t *mask,
                       const int32_t *num_valid, int32_t *idx, cg3d_stream_t s) {
    (void)s;
    if (b < 0 || n < 0 || m < 9) return CG3D_ERR_ARG;
    const int NV = 9, IOFF = 8;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < (int64_t)b * n; p++) {
        const float *v = vertices + p * m * 2;
        const uint8_t *mk = mask + p * m;
        int32_t *out = idx + p * NV;
        int nv = num_valid[p];
        int pad = 0; /* the reference leaves `pad` uninitialised when every intersection slot is valid */
        for (int j = IOFF; j < m; ++j) if (!mk[j]) { pad = j; break; }
        if (nv < 3) { for (int j = 0; j < NV; ++j) out[j] = pad; continue; }
        for (int j = 0; j < NV; ++j) out[j] = 0;
        for (int j = 0; j < nv && j < NV; ++j) { /* nv > 8 overruns idx[] in the reference (UB); clamped */
            float x_min = 1, y_min = (float)(-OG_SV_EPS);
            int i_take = 0;
            for (int k = 0; k < m; ++k) {
                float x = v[k * 2], y = v[k * 2 + 1];
                if (j == 0) {
                    if (mk[k] && og_cmp_vert(x, y, x_min, y_min)) { x_min = x; y_min = y; i_take = k; }
                } else {
                    int i2 = out[j - 1];
                    float x2 = v[i2 * 2], y2 = v[i2 * 2 + 1];
                    if (mk[k] && og_cmp_vert(x, y, x_min, y_min) && og_cmp_vert(x2, y2, x, y)) {
                        x_min = x; y_min = y; i_take = k;
                    }
                }
            }
            out[j] = i_take;
        }
        if (nv < NV) out[nv] = out[0];
        for (int j = nv + 1; j < NV; ++j) out[j] = pad;
        if (nv == 8) { /* identical boxes: duplicate corners (:113-128) */
            int counter = 0;
            for (int j = 0; j < 4; ++j) {
                int check = out[j];
                for (int k = 4; k < IOFF; ++k) if (out[k] == check) counter++;
            }
            if (counter == 4) { out[4] = out[0]; for (int j = 5; j < NV; ++j) out[j] = pad; }
        }
    }
    return CG3D_OK;
}


/* find_points_in_boxes (cagroup3d_assigner.py:9-36): strictly inside the rotated box; same operation order as the
 * reference's tensor expression (and as k_points_in_boxes). */
int cg3d_points_in_boxes(const float *pts, int64_t n, const float *boxes, int32_t g, const int32_t *pseg, const int32_t *bseg,
                         uint8_t *out, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || g < 0) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        for (int32_t j = 0; j < g; j++) {
            const float *b = boxes + (int64_t)j * 7;
            float px = pts[i * 3], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
            float cx = b[0], cy = b[1], cz = b[2];
            float sx = px - cx, sy = py - cy, sz = pz - cz;
            float c = og_cosf(-b[6]), s = og_sinf(-b[6]);
            float rx = sx * c + sy * s, ry = sy * c - sx * s;
            float qx = cx + rx, qy = cy + ry, qz = cz + sz;
            float hx = b[3] / 2, hy = b[4] / 2, hz = b[5] / 2;
            float m = qx - cx + hx;
            m = fminf(m, cx + hx - qx);
            m = fminf(m, qy - cy + hy);
            m = fminf(m, cy + hy - qy);
            m = fminf(m, qz - cz + hz);
            m = fminf(m, cz + hz - qz);
            int in = m > 0.f;
            if (pseg && bseg) in = in && pseg[i] == bseg[j];
            out[i * g + j] = (uint8_t)in;
        }
    return CG3D_OK;
}

/* FCOS-style assignment of the class-map points (cagroup3d_assigner.py:62-130, compute_centerness :39-46), restated per
 * (point, box) pair in the operation order of the reference's tensor expressions (face distances as find_points_in_boxes
 * above; centerness = sqrt(x.min / x.max * y.min / y.max * z.min / z.max) evaluated left to right), compiled with
 * -ffp-contract=off like the rest of this file.  Pinned through CAGroup3DAssigner.assign_all_classes against the fixture the
 * reference's own assigner produced (tests/test_golden.py::test_assigner_matches_reference). */
static float og_fcos_centerness(const float *p, const float *b, int *inside) {
    float cx = b[0], cy = b[1], cz = b[2];
    float sx = p[0] - cx, sy = p[1] - cy, sz = p[2] - cz;
    float c = og_cosf(-b[6]), s = og_sinf(-b[6]);
    float rx = sx * c + sy * s, ry = sy * c - sx * s;
    float qx = cx + rx, qy = cy + ry, qz = cz + sz;
    float hx = b[3] / 2, hy = b[4] / 2, hz = b[5] / 2;
    float x0 = qx - cx + hx, x1 = cx + hx - qx, y0 = qy - cy + hy, y1 = cy + hy - qy, z0 = qz - cz + hz, z1 = cz + hz - qz;
    float m = x0;
    m = fminf(m, x1); m = fminf(m, y0); m = fminf(m, y1); m = fminf(m, z0); m = fminf(m, z1);
    *inside = m > 0.f;
    float v = fminf(x0, x1) / fmaxf(x0, x1);
    v = v * fminf(y0, y1);
    v = v / fmaxf(y0, y1);
    v = v * fminf(z0, z1);
    v = v / fmaxf(z0, z1);
    return sqrtf(v);
}
int cg3d_fcos_centerness(const float *points, const int64_t *pt_cls, const int64_t *pt_scene, int64_t n, const float *gt,
                         const int64_t *gt_cls, const int64_t *gt_scene, int32_t m, float *cness, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || m < 0 || (!pt_scene) != (!gt_scene)) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        for (int32_t j = 0; j < m; j++) {
            int inside;
            const float v = og_fcos_centerness(points + i * 3, gt + (int64_t)j * 7, &inside);
            const int compete = inside && pt_cls[i] == gt_cls[j] && (!pt_scene || pt_scene[i] == gt_scene[j]);
            cness[i * m + j] = compete ? v : -1.f;
        }
    return CG3D_OK;
}
int cg3d_fcos_assign(const float *cness, const float *kth, int64_t n, const float *gt, const int64_t *gt_cls, int32_t m,
                     float *ctr_t, float *box_t, int64_t *labels, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || m < 1) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        float best = 1e8f;
        int bj = 0;
        for (int32_t j = 0; j < m; j++) {
            const float c = cness[i * m + j];
            if (c > kth[j] && c >= 0.f) {
                const float *b = gt + (int64_t)j * 7;
                const float vol = b[3] * b[4] * b[5];
                if (vol < best) { best = vol; bj = j; }
            }
        }
        labels[i] = best != 1e8f ? gt_cls[bj] : -1;
        ctr_t[i] = cness[i * m + bj];
        for (int q = 0; q < 7; q++) box_t[i * 7 + q] = gt[(int64_t)bj * 7 + q];
    }
    return CG3D_OK;
}
