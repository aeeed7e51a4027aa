/* oracle_loss.c -- CPU restatement of the fused focal loss (TEST INFRASTRUCTURE ONLY: tests/, smoke() and bench.py's
 * cpu_baseline may load this; the product path is cagroup3d_amd/csrc/loss.hip).
 * Follows py_sigmoid_focal_loss, reference pcdet/utils/loss_utils.py:903-961 (sigmoid; pt = (1-p)t + p(1-t);
 * focal weight (alpha t + (1-alpha)(1-t)) pt^gamma; BCE-with-logits; weight broadcast over the class axis; sum),
 * with the -1 -> background rewrite of FocalLoss.forward :1024.  Pinned by tests/test_losses_cpu.py against the
 * torch expression and the golden vectors generated from the reference's loss_utils.py. */
#include <math.h>
#include <stdint.h>
#include "../include/cagroup3d_hip.h"

static void focal_terms(float x, float t, float gamma, float alpha, float *loss, float *grad) {
    const float p = 1.f / (1.f + expf(-x));
    const float pt = (1.f - p) * t + p * (1.f - t);
    const float at = alpha * t + (1.f - alpha) * (1.f - t);
    const float bce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
    const float ptg = powf(pt, gamma);
    *loss = bce * at * ptg;
    const float dptg = pt > 0.f ? gamma * powf(pt, gamma - 1.f) * (1.f - 2.f * t) * p * (1.f - p) : 0.f;
    *grad = at * ((p - t) * ptg + bce * dptg);
}

int32_t cg3d_focal_loss_nblocks(int64_t n, int32_t c) {
    (void)n; (void)c;
    return 1;
}
int cg3d_focal_loss_fwd(const float *pred, const int32_t *label, const float *row_w, int64_t n, int32_t c, float gamma,
                        float alpha, float *partial, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || c < 1) return CG3D_ERR_ARG;
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < c; a++) {
            float l, g;
            focal_terms(pred[i * c + a], label[i] == a ? 1.f : 0.f, gamma, alpha, &l, &g);
            s += (double)(l * row_w[i]);
        }
    partial[0] = (float)s;
    return CG3D_OK;
}
int cg3d_focal_loss_bwd(const float *pred, const int32_t *label, const float *row_w, const float *gscale, int64_t n,
                        int32_t c, float gamma, float alpha, float *dpred, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || c < 1) return CG3D_ERR_ARG;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        for (int a = 0; a < c; a++) {
            float l, g;
            focal_terms(pred[i * c + a], label[i] == a ? 1.f : 0.f, gamma, alpha, &l, &g);
            dpred[i * c + a] = g * row_w[i] * gscale[0];
        }
    return CG3D_OK;
}

/* ---- optimiser step (test infrastructure): plain-C statement of cg3d_adamw_step, the per-element arithmetic of torch's
 * fused AdamW kernel (torch/optim/adamw.py `_fused_adamw` -> aten `_fused_adamw_`), gradient pre-scaled by *clip. */
int cg3d_adamw_step(const int64_t *table, const int32_t *pid, int64_t nrows, const int64_t *grads, const float *clip,
                    float lr, float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                    float bias_correction2, cg3d_stream_t stream) {
    (void)stream;
    if (nrows < 0 || (nrows > 0 && (!table || !pid || !grads))) return CG3D_ERR_ARG;
    if (!(bias_correction1 > 0.f) || !(bias_correction2 > 0.f)) return CG3D_ERR_ARG;
    const float cs = clip ? *clip : 1.f;
    const float step_size = lr / bias_correction1, omb1 = 1.f - beta1, omb2 = 1.f - beta2, lrwd = lr * weight_decay;
    const float bc2_sqrt = sqrtf(bias_correction2);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nrows; r++) {
        const int64_t *row = table + r * 5;
        float *p = (float *)(intptr_t)row[0] + row[3], *m = (float *)(intptr_t)row[1] + row[3], *v = (float *)(intptr_t)row[2] + row[3];
        const float *g = (const float *)(intptr_t)grads[pid[r]] + row[3];
        for (int64_t i = 0; i < row[4]; i++) {
            const float gg = g[i] * cs;
            p[i] -= lrwd * p[i];
            m[i] += (gg - m[i]) * omb1;
            v[i] = beta2 * v[i] + omb2 * gg * gg;
            p[i] -= step_size * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
        }
    }
    return CG3D_OK;
}

/* clip_grad_norm_ over the chunk table: norm = sqrt(sum g^2) (double accumulation), coef = min(max_norm / (norm + 1e-6), 1)
 * (torch/nn/utils/clip_grad.py: total_norm, clip_coef_clamped) */
int cg3d_grad_norm_clip(const int64_t *table, const int32_t *pid, int64_t nrows, const int64_t *grads, float max_norm,
                        double *scratch, float *norm, float *coef, cg3d_stream_t stream) {
    (void)stream;
    if (nrows < 0 || (nrows > 0 && (!table || !pid || !grads)) || !scratch || !norm || !coef) return CG3D_ERR_ARG;
    double s = 0;
    for (int64_t r = 0; r < nrows; r++) {
        const int64_t *row = table + r * 5;
        const float *g = (const float *)(intptr_t)grads[pid[r]] + row[3];
        for (int64_t i = 0; i < row[4]; i++) s += (double)g[i] * g[i];
    }
    *scratch = s;
    *norm = (float)sqrt(s);
    const float c = max_norm / (*norm + 1e-6f);
    *coef = (c < 1.f || c != c) ? c : 1.f;     /* torch.clamp(max=1) propagates NaN */
    return CG3D_OK;
}

/* ---- positives of the class maps (test infrastructure): centerness BCE + axis-aligned IoU loss, restated in the
 * REFERENCE's parametrisation -- `_bbox_pred_to_bbox` (dense_heads/cagroup_head.py:654-668: centre = p + (d+ - d-)/2,
 * size = d- + d+), corners = centre -/+ size/2 (iou3d_loss.py:_corners), axis_aligned_bbox_overlaps_3d
 * (loss_utils.py:419-538: max / min of the corners, clamp(min=0), union clamped at eps 1e-6), loss 1 - IoU weighted by the
 * centerness target / (sum of centerness targets of the scene * B) (cagroup_head.py:537-546), centerness loss
 * BCE-with-logits / ((positives of the scene + eps) * B) (:532-536) -- in double precision, derivatives by the chain rule
 * through (centre, size).  Pinned against the torch mirror of those lines (itself pinned by the reference's fixtures) in
 * tests/test_fused_losses.py. */
typedef struct { double bce, dbce, loss_b, dd[6]; } ol_pos;
static ol_pos ol_pos_terms(float pc, float ct, const float *p, const float *d, const float *t) {
    ol_pos r;
    const double x = pc, sg = 1.0 / (1.0 + exp(-x));
    r.bce = fmax(x, 0.0) - x * ct + log1p(exp(-fabs(x)));
    r.dbce = sg - ct;
    double lo[3], hi[3], tlo[3], thi[3], wh[3], s[3], a1 = 1, a2 = 1, ov = 1;
    for (int a = 0; a < 3; a++) {
        const double c = (double)p[a] + ((double)d[2 * a + 1] - (double)d[2 * a]) / 2, sz = (double)d[2 * a] + (double)d[2 * a + 1];
        lo[a] = c - sz / 2; hi[a] = c + sz / 2;
        tlo[a] = (double)t[a] - (double)t[3 + a] / 2; thi[a] = (double)t[a] + (double)t[3 + a] / 2;
        s[a] = hi[a] - lo[a];
        const double w = fmin(hi[a], thi[a]) - fmax(lo[a], tlo[a]);
        wh[a] = w > 0 ? w : 0;
        a1 *= s[a]; a2 *= thi[a] - tlo[a]; ov *= wh[a];
    }
    const double ub = a1 + a2 - ov, un = ub < 1e-6 ? 1e-6 : ub;
    r.loss_b = 1.0 - ov / un;
    for (int a = 0; a < 3; a++) {
        const double ov_o = wh[(a + 1) % 3] * wh[(a + 2) % 3], a1_o = s[(a + 1) % 3] * s[(a + 2) % 3];
        const double dov_hi = (wh[a] > 0 && hi[a] < thi[a]) ? ov_o : 0, dov_lo = (wh[a] > 0 && lo[a] > tlo[a]) ? -ov_o : 0;
        const double dun_hi = ub < 1e-6 ? 0 : a1_o - dov_hi, dun_lo = ub < 1e-6 ? 0 : -a1_o - dov_lo;
        const double dl_hi = -(dov_hi * un - ov * dun_hi) / (un * un), dl_lo = -(dov_lo * un - ov * dun_lo) / (un * un);
        /* lo = c - s/2, hi = c + s/2; c = p + (d+ - d-)/2, s = d- + d+ */
        const double dl_c = dl_lo + dl_hi, dl_s = (dl_hi - dl_lo) / 2;
        r.dd[2 * a] = -dl_c / 2 + dl_s;
        r.dd[2 * a + 1] = dl_c / 2 + dl_s;
    }
    return r;
}
int32_t cg3d_pos_loss_nblocks(int64_t npos) { (void)npos; return 1; }
int cg3d_pos_loss_fwd(const float *cent, const float *bbox, const float *points, const float *ctr_t, const float *bbox_t,
                      int32_t tstride, const int64_t *scene, const float *n_pos, const float *ctr_den, const int64_t *pos,
                      int64_t npos, float wc, float wb, float eps, float *partial, cg3d_stream_t stream) {
    (void)stream;
    if (npos < 0 || tstride < 6) return CG3D_ERR_ARG;
    double s0 = 0, s1 = 0;
    for (int64_t i = 0; i < npos; i++) {
        const int64_t r = pos[i], sc = scene[r];
        const ol_pos T = ol_pos_terms(cent[r], ctr_t[r], points + r * 3, bbox + r * 6, bbox_t + r * tstride);
        s0 += T.bce * ((double)wc / ((double)n_pos[sc] + eps));
        s1 += T.loss_b * ((double)wb * ctr_t[r] / ctr_den[sc]);
    }
    partial[0] = (float)s0; partial[1] = (float)s1;
    return CG3D_OK;
}
int cg3d_pos_loss_bwd(const float *cent, const float *bbox, const float *points, const float *ctr_t, const float *bbox_t,
                      int32_t tstride, const int64_t *scene, const float *n_pos, const float *ctr_den, const int64_t *pos,
                      int64_t npos, float wc, float wb, float eps, const float *gscale, float *dcent, float *dbbox,
                      cg3d_stream_t stream) {
    (void)stream;
    if (npos < 0 || tstride < 6) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < npos; i++) {
        const int64_t r = pos[i], sc = scene[r];
        const ol_pos T = ol_pos_terms(cent[r], ctr_t[r], points + r * 3, bbox + r * 6, bbox_t + r * tstride);
        dcent[r] = (float)(gscale[0] * T.dbce * ((double)wc / ((double)n_pos[sc] + eps)));
        const double gb = (double)gscale[1] * ((double)wb * ctr_t[r] / ctr_den[sc]);
        for (int j = 0; j < 6; j++) dbbox[r * 6 + j] = (float)(gb * T.dd[j]);
    }
    return CG3D_OK;
}
/* ---- smooth-L1 with row weights, reduction 'sum' (SmoothL1Loss, loss_utils.py:1042-1123; the vote loss cagroup_head.py:512-519) */
int cg3d_smooth_l1_rows_fwd(const float *pred, const float *target, const float *w, int64_t n, int32_t d, float beta,
                            float *partial, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || d < 1 || !(beta > 0.f)) return CG3D_ERR_ARG;
    double s = 0;
    for (int64_t t = 0; t < n * d; t++) {
        const double e = fabs((double)pred[t] - (double)target[t]);
        s += (e < beta ? 0.5 * e * e / beta : e - 0.5 * beta) * w[t / d];
    }
    partial[0] = (float)s;
    return CG3D_OK;
}
int cg3d_smooth_l1_rows_bwd(const float *pred, const float *target, const float *w, const float *gscale, int64_t n, int32_t d,
                            float beta, float *dpred, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || d < 1 || !(beta > 0.f)) return CG3D_ERR_ARG;
    for (int64_t t = 0; t < n * d; t++) {
        const float df = pred[t] - target[t], e = fabsf(df);
        dpred[t] = gscale[0] * w[t / d] * (e < beta ? df / beta : (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)));
    }
    return CG3D_OK;
}
