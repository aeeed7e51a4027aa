// dg_geom.h -- deterministic fp32 trigonometry and the rotated BEV overlap shared by iou3d_nms.hip and stages.hip
// (reference pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:28-234; the oracle's twin is oracle/oracle_geom.c).  Translation units
// that include it are built with -ffp-contract=off: results are bit-identical to the CPU oracle.
#pragma once
#include "cg3d_common.h"

// ---------------------------------------------------------------- deterministic fp32 trig
// Same Cephes-style polynomials as the oracle (oracle/oracle_geom.c); device libm is avoided so
// that host and device agree bit-for-bit.
#define DG_FOPI 1.27323954473516f
#define DG_DP1 0.78515625f
#define DG_DP2 2.4187564849853515625e-4f
#define DG_DP3 3.77489497744594108e-8f

__host__ __device__ static inline float dg_poly_sin(float x, float z) {
    float p = -1.9515295891E-4f;
    p = p * z + 8.3321608736E-3f;
    p = p * z + -1.6666654611E-1f;
    return p * z * x + x;
}
__host__ __device__ static inline float dg_poly_cos(float z) {
    float p = 2.443315711809948E-005f;
    p = p * z + -1.388731625493765E-003f;
    p = p * z + 4.166664568298827E-002f;
    return p * z * z - 0.5f * z + 1.0f;
}
__host__ __device__ static inline float dg_sinf(float x) {
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    int j = (int)(DG_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sign = -sign; j -= 4; }
    x = ((x - y * DG_DP1) - y * DG_DP2) - y * DG_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? dg_poly_cos(z) : dg_poly_sin(x, z);
    return sign * r;
}
__host__ __device__ static inline float dg_cosf(float x) {
    float sign = 1.0f;
    if (x < 0.0f) x = -x;
    int j = (int)(DG_FOPI * x);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sign = -sign; j -= 4; }
    if (j > 1) sign = -sign;
    x = ((x - y * DG_DP1) - y * DG_DP2) - y * DG_DP3;
    float z = x * x;
    float r = (j == 1 || j == 2) ? dg_poly_sin(x, z) : dg_poly_cos(z);
    return sign * r;
}
__host__ __device__ static inline float dg_atanf(float x) {
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
__host__ __device__ static inline float dg_atan2f(float y, float x) {
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
    return w + dg_atanf(y / x);
}

// ---------------------------------------------------------------- rotated BEV overlap
#define DG_EPS 1e-8f
struct dpt { float x, y; };

__host__ __device__ static inline float d_cross3(dpt p1, dpt p2, dpt p0) {
    return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}
__host__ __device__ static inline float d_min(float a, float b) { return a > b ? b : a; }
__host__ __device__ static inline float d_max(float a, float b) { return a > b ? a : b; }
__host__ __device__ static inline bool d_rect_cross(dpt p1, dpt p2, dpt q1, dpt q2) {
    return d_min(p1.x, p2.x) <= d_max(q1.x, q2.x) && d_min(q1.x, q2.x) <= d_max(p1.x, p2.x) &&
           d_min(p1.y, p2.y) <= d_max(q1.y, q2.y) && d_min(q1.y, q2.y) <= d_max(p1.y, p2.y);
}
// corner-in-box test with the box's cos(-h), sin(-h) hoisted by the caller
__host__ __device__ static inline bool d_in_box2d(const float *box, float ac, float as, dpt p) {
    const float MARGIN = 1e-2f;
    float cx = box[0], cy = box[1];
    float rx = (p.x - cx) * ac + (p.y - cy) * (-as);
    float ry = (p.x - cx) * as + (p.y - cy) * ac;
    return (fabsf(rx) < box[3] / 2 + MARGIN && fabsf(ry) < box[4] / 2 + MARGIN);
}
__host__ __device__ static inline bool d_intersection(dpt p1, dpt p0, dpt q1, dpt q0, dpt *ans) {
    if (!d_rect_cross(p0, p1, q0, q1)) return false;
    float s1 = d_cross3(q0, p1, p0);
    float s2 = d_cross3(p1, q1, p0);
    float s3 = d_cross3(p0, q1, q0);
    float s4 = d_cross3(q1, p1, q0);
    if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
    float s5 = d_cross3(q1, p1, p0);
    if (fabsf(s5 - s1) > DG_EPS) {
        ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
        ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
    } else {
        float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
        float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
        float D = a0 * b1 - a1 * b0;
        ans->x = (b0 * c1 - b1 * c0) / D;
        ans->y = (a1 * c0 - a0 * c1) / D;
    }
    return true;
}
__host__ __device__ static inline dpt d_rot(dpt c, float ac, float as, dpt p) {
    dpt r;
    r.x = (p.x - c.x) * ac + (p.y - c.y) * (-as) + c.x;
    r.y = (p.x - c.x) * as + (p.y - c.y) * ac + c.y;
    return r;
}

__host__ __device__ static float d_box_overlap(const float *a, const float *b) {
    float adx = a[3] / 2, bdx = b[3] / 2, ady = a[4] / 2, bdy = b[4] / 2;
    float ax1 = a[0] - adx, ay1 = a[1] - ady, ax2 = a[0] + adx, ay2 = a[1] + ady;
    float bx1 = b[0] - bdx, by1 = b[1] - bdy, bx2 = b[0] + bdx, by2 = b[1] + bdy;
    dpt ca = {a[0], a[1]}, cb = {b[0], b[1]};
    dpt A[5] = {{ax1, ay1}, {ax2, ay1}, {ax2, ay2}, {ax1, ay2}, {0.f, 0.f}};
    dpt B[5] = {{bx1, by1}, {bx2, by1}, {bx2, by2}, {bx1, by2}, {0.f, 0.f}};
    float aco = dg_cosf(a[6]), asi = dg_sinf(a[6]), bco = dg_cosf(b[6]), bsi = dg_sinf(b[6]);
#pragma unroll
    for (int k = 0; k < 4; k++) { A[k] = d_rot(ca, aco, asi, A[k]); B[k] = d_rot(cb, bco, bsi, B[k]); }
    A[4] = A[0]; B[4] = B[0];

    dpt cp[16];
    float ang[16];
    dpt ctr = {0.f, 0.f};
    int cnt = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            dpt t;
            if (d_intersection(A[i + 1], A[i], B[j + 1], B[j], &t)) {
                cp[cnt] = t; ctr.x = ctr.x + t.x; ctr.y = ctr.y + t.y; cnt++;
            }
        }
    // cos(-h) / sin(-h) evaluated once per box (same operands as the per-corner calls)
    float nac = dg_cosf(-a[6]), nas = dg_sinf(-a[6]), nbc = dg_cosf(-b[6]), nbs = dg_sinf(-b[6]);
    for (int k = 0; k < 4; k++) {
        if (d_in_box2d(a, nac, nas, B[k])) { ctr.x = ctr.x + B[k].x; ctr.y = ctr.y + B[k].y; cp[cnt++] = B[k]; }
        if (d_in_box2d(b, nbc, nbs, A[k])) { ctr.x = ctr.x + A[k].x; ctr.y = ctr.y + A[k].y; cp[cnt++] = A[k]; }
    }
    ctr.x /= cnt; ctr.y /= cnt;
    for (int i = 0; i < cnt; i++) ang[i] = dg_atan2f(cp[i].y - ctr.y, cp[i].x - ctr.x);
    for (int j = 0; j < cnt - 1; j++)
        for (int i = 0; i < cnt - j - 1; i++)
            if (ang[i] > ang[i + 1]) {
                dpt t = cp[i]; cp[i] = cp[i + 1]; cp[i + 1] = t;
                float ta = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = ta;
            }
    float area = 0.f;
    for (int k = 0; k < cnt - 1; k++) {
        float ux = cp[k].x - cp[0].x, uy = cp[k].y - cp[0].y;
        float vx = cp[k + 1].x - cp[0].x, vy = cp[k + 1].y - cp[0].y;
        area += ux * vy - uy * vx;
    }
    return fabsf(area) / 2.0f;
}
__host__ __device__ static inline float d_iou_bev(const float *a, const float *b) {
    float sa = a[3] * a[4], sb = b[3] * b[4];
    float so = d_box_overlap(a, b);
    return so / fmaxf(sa + sb - so, DG_EPS);
}
__host__ __device__ static inline float d_iou_normal(const float *a, const float *b) {
    float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    float inter = width * height;
    float Sa = a[3] * a[4], Sb = b[3] * b[4];
    return inter / fmaxf(Sa + Sb - inter, DG_EPS);
}

