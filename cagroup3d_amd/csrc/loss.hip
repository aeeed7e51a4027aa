// loss.hip -- fused sigmoid focal loss with per-row weights for gfx950.
//
// Replaces the element-wise chain of `py_sigmoid_focal_loss` (reference pcdet/utils/loss_utils.py:903-961, called by
// FocalLoss :964-1040 from cagroup_head.py:520-531 for the semantic and the classification scores): sigmoid, the
// (1-p)t + p(1-t) blend, pow(gamma), BCE-with-logits, weight broadcast and the sum -- ~15 launches forward and ~25
// backward on [N, 18] tensors -- as one pass each way.  HBM-bound: pred read once per pass, dpred written once.
//   label[i] in [0, C) = foreground class, anything else (the reference rewrites -1 to C) = background row.
//   loss_i,c = row_w[i] * BCE(x, t) * (alpha t + (1-alpha)(1-t)) * pt^gamma,  pt = (1-p) t + p (1-t),  p = sigmoid(x)
// Forward: fp32 per-block partial sums (plain stores, summed by the caller in a fixed order: deterministic).
#include "cg3d_common.h"

__device__ static inline void focal_terms(float x, float t, float gamma, float alpha, float &loss, float &grad) {
    const float p = 1.f / (1.f + expf(-x));
    const float pt = (1.f - p) * t + p * (1.f - t);
    const float at = alpha * t + (1.f - alpha) * (1.f - t);
    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    const float ptg = powf(pt, gamma);
    loss = bce * at * ptg;
    // d pt / dx = (1 - 2t) p (1 - p);  d bce / dx = p - t
    const float dptg = pt > 0.f ? gamma * powf(pt, gamma - 1.f) * (1.f - 2.f * t) * p * (1.f - p) : 0.f;
    grad = at * ((p - t) * ptg + bce * dptg);
}

__global__ __launch_bounds__(256) void k_focal_fwd(const float *__restrict__ pred, const int32_t *__restrict__ label,
                                                   const float *__restrict__ row_w, int64_t n, int32_t c, float gamma,
                                                   float alpha, float *__restrict__ partial) {
    __shared__ float red[256];
    const int64_t total = n * c;
    float s = 0.f;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t i = t / c;
        const int a = (int)(t - i * c);
        float l, g;
        focal_terms(pred[t], label[i] == a ? 1.f : 0.f, gamma, alpha, l, g);
        s += l * row_w[i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_focal_bwd(const float *__restrict__ pred, const int32_t *__restrict__ label,
                                                   const float *__restrict__ row_w, const float *__restrict__ gscale,
                                                   int64_t n, int32_t c, float gamma, float alpha,
                                                   float *__restrict__ dpred) {
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (t >= n * c) return;
    const int64_t i = t / c;
    const int a = (int)(t - i * c);
    float l, g;
    focal_terms(pred[t], label[i] == a ? 1.f : 0.f, gamma, alpha, l, g);
    dpred[t] = g * row_w[i] * gscale[0];
}

extern "C" int32_t cg3d_focal_loss_nblocks(int64_t n, int32_t c) {
    const int64_t b = cg3d_divup(n * (int64_t)c, 256 * 4);
    return (int32_t)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
extern "C" int cg3d_focal_loss_fwd(const float *pred, const int32_t *label, const float *row_w, int64_t n, int32_t c,
                                   float gamma, float alpha, float *partial, cg3d_stream_t stream) {
    if (n < 0 || c < 1) return CG3D_ERR_ARG;
    const int32_t nb = cg3d_focal_loss_nblocks(n, c);
    hipLaunchKernelGGL(k_focal_fwd, dim3((unsigned)nb), dim3(256), 0, cg3d_hs(stream), pred, label, row_w, n, c, gamma,
                       alpha, partial);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_focal_loss_bwd(const float *pred, const int32_t *label, const float *row_w, const float *gscale,
                                   int64_t n, int32_t c, float gamma, float alpha, float *dpred, cg3d_stream_t stream) {
    if (n < 0 || c < 1) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_focal_bwd, dim3((unsigned)cg3d_divup(n * (int64_t)c, 256)), dim3(256), 0, cg3d_hs(stream), pred,
                       label, row_w, gscale, n, c, gamma, alpha, dpred);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ positives of the class maps
// Centerness BCE + axis-aligned IoU loss over the positive points (reference dense_heads/cagroup_head.py:532-546: the
// centerness loss :532-536, the box loss :537-546 through `_bbox_pred_to_bbox` :654-668 and `axis_aligned_bbox_overlaps_3d`,
// pcdet/utils/loss_utils.py:419-538, `IoU3DLoss` iou3d_loss.py:14-95) -- ~45 element-wise launches forward and ~80 backward
// (index / select / slice / max / min / clamp / mul / div nodes) on a few thousand rows -- as one pass each way.
//   prediction box of a point p with face distances (dx-,dx+,dy-,dy+,dz-,dz+):  lo = p - d-,  hi = p + d+
//   (the reference goes through centre = p + (d+ - d-)/2 and size = d- + d+ and back to corners: the same numbers up to rounding)
struct PosTerms { float bce, dbce, loss_b, dlo[3], dhi[3]; };
__device__ static inline PosTerms pos_terms(float pc, float ct, const float *p, const float *d, const float *t) {
    PosTerms r;
    const float sg = 1.f / (1.f + expf(-pc));
    r.bce = fmaxf(pc, 0.f) - pc * ct + log1pf(expf(-fabsf(pc)));
    r.dbce = sg - ct;
    float lo[3], hi[3], tlo[3], thi[3], wh[3], s[3];
    float a1 = 1.f, a2 = 1.f, ov = 1.f;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        lo[a] = p[a] - d[2 * a]; hi[a] = p[a] + d[2 * a + 1];
        tlo[a] = t[a] - t[3 + a] / 2; thi[a] = t[a] + t[3 + a] / 2;
        s[a] = hi[a] - lo[a];
        wh[a] = fmaxf(fminf(hi[a], thi[a]) - fmaxf(lo[a], tlo[a]), 0.f);
        a1 *= s[a]; a2 *= thi[a] - tlo[a]; ov *= wh[a];
    }
    const float ub = a1 + a2 - ov;
    const bool clampu = ub < 1e-6f;
    const float un = clampu ? 1e-6f : ub;
    r.loss_b = 1.f - ov / un;
    // d iou = (d ov * un - ov * d un) / un^2,  d un = d a1 - d ov (0 when clamped)
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float ov_o = wh[(a + 1) % 3] * wh[(a + 2) % 3], a1_o = s[(a + 1) % 3] * s[(a + 2) % 3];
        const float dov_hi = (wh[a] > 0.f && hi[a] < thi[a]) ? ov_o : 0.f;       // min(hi, thi) follows hi
        const float dov_lo = (wh[a] > 0.f && lo[a] > tlo[a]) ? -ov_o : 0.f;      // max(lo, tlo) follows lo
        const float dun_hi = clampu ? 0.f : a1_o - dov_hi, dun_lo = clampu ? 0.f : -a1_o - dov_lo;
        r.dhi[a] = -(dov_hi * un - ov * dun_hi) / (un * un);                     // d (1 - iou)
        r.dlo[a] = -(dov_lo * un - ov * dun_lo) / (un * un);
    }
    return r;
}
__global__ __launch_bounds__(256) void k_pos_loss_fwd(const float *__restrict__ cent, const float *__restrict__ bbox,
                                                      const float *__restrict__ points, const float *__restrict__ ctr_t,
                                                      const float *__restrict__ bbox_t, int32_t tstride,
                                                      const int64_t *__restrict__ scene, const float *__restrict__ n_pos,
                                                      const float *__restrict__ ctr_den, const int64_t *__restrict__ pos,
                                                      int64_t npos, float wc, float wb, float eps, float *__restrict__ partial) {
    __shared__ float r0[256], r1[256];
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < npos; i += (int64_t)gridDim.x * 256) {
        const int64_t r = pos[i];
        const PosTerms T = pos_terms(cent[r], ctr_t[r], points + r * 3, bbox + r * 6, bbox_t + r * tstride);
        const int64_t sc = scene[r];
        s0 += T.bce * (wc / (n_pos[sc] + eps));
        s1 += T.loss_b * (wb * ctr_t[r] / ctr_den[sc]);
    }
    r0[threadIdx.x] = s0; r1[threadIdx.x] = s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { r0[threadIdx.x] += r0[threadIdx.x + o]; r1[threadIdx.x] += r1[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = r0[0]; partial[blockIdx.x * 2 + 1] = r1[0]; }
}
__global__ __launch_bounds__(256) void k_pos_loss_bwd(const float *__restrict__ cent, const float *__restrict__ bbox,
                                                      const float *__restrict__ points, const float *__restrict__ ctr_t,
                                                      const float *__restrict__ bbox_t, int32_t tstride,
                                                      const int64_t *__restrict__ scene, const float *__restrict__ n_pos,
                                                      const float *__restrict__ ctr_den, const int64_t *__restrict__ pos,
                                                      int64_t npos, float wc, float wb, float eps,
                                                      const float *__restrict__ gscale, float *__restrict__ dcent,
                                                      float *__restrict__ dbbox) {
    const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (i >= npos) return;
    const int64_t r = pos[i];
    const PosTerms T = pos_terms(cent[r], ctr_t[r], points + r * 3, bbox + r * 6, bbox_t + r * tstride);
    const int64_t sc = scene[r];
    dcent[r] = gscale[0] * T.dbce * (wc / (n_pos[sc] + eps));
    const float gb = gscale[1] * (wb * ctr_t[r] / ctr_den[sc]);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        dbbox[r * 6 + 2 * a] = -gb * T.dlo[a];           // lo = p - d-
        dbbox[r * 6 + 2 * a + 1] = gb * T.dhi[a];        // hi = p + d+
    }
}
extern "C" int32_t cg3d_pos_loss_nblocks(int64_t npos) {
    const int64_t b = cg3d_divup(npos, 256);
    return (int32_t)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}
extern "C" int cg3d_pos_loss_fwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                                 const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                                 const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                                 float *partial, cg3d_stream_t stream) {
    if (npos < 0 || tstride < 6) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_pos_loss_fwd, dim3((unsigned)cg3d_pos_loss_nblocks(npos)), dim3(256), 0, cg3d_hs(stream), centerness,
                       bbox_pred, points, ctr_t, bbox_t, tstride, scene, n_pos, ctr_denorm, pos, npos, wc, wb, eps, partial);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_pos_loss_bwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                                 const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                                 const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                                 const float *gscale, float *dcenterness, float *dbbox_pred, cg3d_stream_t stream) {
    if (npos < 0 || tstride < 6) return CG3D_ERR_ARG;
    if (npos == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_pos_loss_bwd, dim3((unsigned)cg3d_divup(npos, 256)), dim3(256), 0, cg3d_hs(stream), centerness, bbox_pred,
                       points, ctr_t, bbox_t, tstride, scene, n_pos, ctr_denorm, pos, npos, wc, wb, eps, gscale, dcenterness,
                       dbbox_pred);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ smooth-L1 with row weights
// sum_i w[i] * sum_j smooth_l1(pred[i,j] - target[i,j]; beta)  (the vote loss, cagroup_head.py:512-519 / SmoothL1Loss
// loss_utils.py:1042-1123 with reduction 'sum'): abs, where, mul, sum and their five backward nodes in one pass each way.
__global__ __launch_bounds__(256) void k_sl1_fwd(const float *__restrict__ pred, const float *__restrict__ tgt,
                                                 const float *__restrict__ w, int64_t n, int32_t d, float beta,
                                                 float *__restrict__ partial) {
    __shared__ float red[256];
    float s = 0.f;
    for (int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x; t < n * d; t += (int64_t)gridDim.x * 256) {
        const float e = fabsf(pred[t] - tgt[t]);
        s += (e < beta ? 0.5f * e * e / beta : e - 0.5f * beta) * w[t / d];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void k_sl1_bwd(const float *__restrict__ pred, const float *__restrict__ tgt,
                                                 const float *__restrict__ w, const float *__restrict__ gscale, int64_t n,
                                                 int32_t d, float beta, float *__restrict__ dpred) {
    const int64_t t = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (t >= n * d) return;
    const float df = pred[t] - tgt[t], e = fabsf(df);
    const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
    dpred[t] = gscale[0] * w[t / d] * (e < beta ? df / beta : sgn);
}
extern "C" int cg3d_smooth_l1_rows_fwd(const float *pred, const float *target, const float *w, int64_t n, int32_t d, float beta,
                                       float *partial, cg3d_stream_t stream) {
    if (n < 0 || d < 1 || !(beta > 0.f)) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_sl1_fwd, dim3((unsigned)cg3d_focal_loss_nblocks(n, d)), dim3(256), 0, cg3d_hs(stream), pred, target, w, n, d,
                       beta, partial);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_smooth_l1_rows_bwd(const float *pred, const float *target, const float *w, const float *gscale, int64_t n,
                                       int32_t d, float beta, float *dpred, cg3d_stream_t stream) {
    if (n < 0 || d < 1 || !(beta > 0.f)) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_sl1_bwd, dim3((unsigned)cg3d_divup(n * (int64_t)d, 256)), dim3(256), 0, cg3d_hs(stream), pred, target, w,
                       gscale, n, d, beta, dpred);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
