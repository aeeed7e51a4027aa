/* oracle_stages.c -- CPU restatement of the stage-level operators of include/cagroup3d_stages.h.
 *
 * TEST INFRASTRUCTURE: the checker of tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never the product.
 * Plain loops in the reference's own operation order (citations into /root/reference); pinned against the torch mirrors of
 * the same reference code (tests/test_stages.py), which the reference's fixtures pin (tests/test_golden.py).
 * Built with -ffp-contract=off like oracle_geom.c, whose rotated overlap it calls.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "../include/cagroup3d_stages.h"

float og_box_overlap(const float *a, const float *b); /* oracle_geom.c (iou3d_nms_kernel.cu:99-225) */

static const float OS_2PI = 6.283185307179586f, OS_PI = 3.141592653589793f;

/* torch.remainder on floats (ATen BinaryOps: fmod moved into the sign of the divisor) */
static float os_remainder(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}

/* padded RoI i of scene b: reoder_rois_for_refining (cagroup_roi_head.py:328-362; heading sign :358) and the enlargement of
 * forward_train (:270-272) */
static void os_load_roi(const float *boxes, const float *scores, const int64_t *labels, const int32_t *roi_off, int b, int i,
                        float enlarge, float roi[7], int64_t *label, float *score) {
    const int o = roi_off[b], n = roi_off[b + 1] - o;
    if (i < n) {
        const float *p = boxes + (int64_t)(o + i) * 7;
        roi[0] = p[0]; roi[1] = p[1]; roi[2] = p[2];
        roi[3] = p[3] * enlarge; roi[4] = p[4] * enlarge; roi[5] = p[5] * enlarge;
        roi[6] = p[6] * -1.f;
        *label = labels[o + i];
        if (score) *score = scores[o + i];
    } else {
        for (int k = 0; k < 6; k++) roi[k] = 0.f;
        roi[6] = -0.f;
        *label = 0;
        if (score) *score = 0.f;
    }
}

/* boxes_iou3d_gpu, iou3d_nms_utils.py:59-79 */
static float os_iou3d(const float *a, const float *b) {
    const float a_hmax = a[2] + a[5] / 2, a_hmin = a[2] - a[5] / 2;
    const float b_hmax = b[2] + b[5] / 2, b_hmin = b[2] - b[5] / 2;
    const float ob = og_box_overlap(a, b);
    float oh = fminf(a_hmax, b_hmax) - fmaxf(a_hmin, b_hmin);
    oh = oh < 0.f ? 0.f : oh;
    const float o3 = ob * oh;
    const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
    float u = va + vb - o3;
    u = u < 1e-6f ? 1e-6f : u;
    return o3 / u;
}

/* get_max_iou_with_same_class, cagroup_proposal_target_layer.py:204-238 */
int cg3d_roi_match(const float *boxes, const int64_t *labels, const int32_t *roi_off, int32_t nb, int32_t rin, float enlarge,
                   const float *gt_boxes, int32_t gmax, int32_t gdim, const int32_t *n_gt, float *max_ov, int32_t *assign,
                   cg3d_stream_t stream) {
    (void)stream;
    if (nb < 0 || rin < 0 || gmax < 0 || gdim < 8) return CG3D_ERR_ARG;
    if ((int64_t)nb * rin == 0) return CG3D_OK;
    if (!roi_off || !n_gt || !max_ov || !assign) return CG3D_ERR_ARG;
    for (int b = 0; b < nb; b++)
        for (int i = 0; i < rin; i++) {
            float roi[7];
            int64_t label;
            os_load_roi(boxes, NULL, labels, roi_off, b, i, enlarge, roi, &label, NULL);
            float best = -1.f;
            int arg = 0;
            for (int g = 0; g < n_gt[b]; g++) {
                const float *q = gt_boxes + ((int64_t)b * gmax + g) * gdim;
                if ((int64_t)q[7] != label) continue;
                float gb[7] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6] * -1.f};
                const float v = os_iou3d(roi, gb);
                if (v > best) { best = v; arg = g; }
            }
            max_ov[(int64_t)b * rin + i] = best >= 0.f ? best : 0.f;
            assign[(int64_t)b * rin + i] = best >= 0.f ? arg : 0;
        }
    return CG3D_OK;
}

/* gathers of sample_rois_for_rcnn (:44-63), ProposalTargetLayer.forward (:22-33), assign_targets (cagroup_roi_head.py:291-326),
 * encode_torch (cagroup_utils.py:101-136) as called by get_box_reg_layer_loss (cagroup_roi_head.py:551-577) */
int cg3d_roi_targets(const float *boxes, const float *scores, const int64_t *labels, const int32_t *roi_off, int32_t nb,
                     int32_t rin, float enlarge, const float *gt_boxes, int32_t gmax, int32_t gdim, const float *max_ov,
                     const int32_t *assign, const int32_t *keep, int32_t rsel, int32_t code_size, float reg_fg, float cls_fg,
                     float cls_bg, float cls_span, float *o_rois, float *o_gt_src, float *o_gt, float *o_gt_label,
                     float *o_iou, float *o_score, int64_t *o_label, int64_t *o_reg_valid, float *o_cls_label,
                     float *o_reg_target, cg3d_stream_t stream) {
    (void)stream;
    if (nb < 0 || rin < 0 || rsel < 0 || gdim < 8 || (code_size < 6 || code_size > 8)) return CG3D_ERR_ARG;
    const int64_t m = (int64_t)nb * rsel;
    if (m == 0) return CG3D_OK;
    if (!roi_off || !gt_boxes || !max_ov || !assign || !keep || !o_rois || !o_gt_src || !o_gt || !o_gt_label || !o_iou ||
        !o_score || !o_label || !o_reg_valid || !o_cls_label || !o_reg_target)
        return CG3D_ERR_ARG;
    for (int64_t j = 0; j < m; j++) {
        const int b = (int)(j / rsel), src = keep[j];
        float roi[7], score;
        int64_t label;
        os_load_roi(boxes, scores, labels, roi_off, b, src, enlarge, roi, &label, &score);
        const int64_t pr = (int64_t)b * rin + src;
        const float iou = max_ov[pr];
        const float *q = gt_boxes + ((int64_t)b * gmax + assign[pr]) * gdim;
        float g[7] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6] * -1.f};
        for (int k = 0; k < 7; k++) { o_rois[j * 7 + k] = roi[k]; o_gt_src[j * 7 + k] = g[k]; }
        o_gt_label[j] = (float)(int64_t)q[7];
        o_iou[j] = iou;
        o_score[j] = score;
        o_label[j] = label;
        o_reg_valid[j] = iou > reg_fg ? 1 : 0;
        const int fg = iou > cls_fg, bg = iou < cls_bg;
        o_cls_label[j] = (!fg && !bg) ? (iou - cls_bg) / cls_span : (fg ? 1.f : 0.f);
        const float ry = os_remainder(roi[6], OS_2PI);
        float c[7];
        c[0] = g[0] - roi[0]; c[1] = g[1] - roi[1]; c[2] = g[2] - roi[2];
        c[3] = g[3]; c[4] = g[4]; c[5] = g[5];
        c[6] = os_remainder(g[6], OS_2PI) - ry;
        if (code_size > 6) {
            const float ang = -ry, cs = cosf(ang), sn = sinf(ang);
            const float x = c[0] * cs + c[1] * (-sn) + c[2] * 0.f, y = c[0] * sn + c[1] * cs + c[2] * 0.f;
            c[0] = x; c[1] = y;
            float h = os_remainder(c[6], OS_2PI);
            if (h > OS_PI * 0.5f && h < OS_PI * 1.5f) h = os_remainder(h + OS_PI, OS_2PI);
            if (h > OS_PI) h = h - OS_2PI;
            c[6] = fminf(fmaxf(h, -OS_PI / 2), OS_PI / 2);
        }
        const float a3 = fmaxf(roi[3], 1e-5f), a4 = fmaxf(roi[4], 1e-5f), a5 = fmaxf(roi[5], 1e-5f);
        c[3] = fmaxf(c[3], 1e-5f); c[4] = fmaxf(c[4], 1e-5f); c[5] = fmaxf(c[5], 1e-5f);
        for (int k = 0; k < 7; k++) o_gt[j * 7 + k] = c[k];
        const float diag = sqrtf(a3 * a3 + a4 * a4);
        float *t = o_reg_target + j * code_size;
        t[0] = c[0] / diag; t[1] = c[1] / diag; t[2] = c[2] / a5;
        t[3] = logf(c[3] / a3); t[4] = logf(c[4] / a4); t[5] = logf(c[5] / a5);
        if (code_size == 7) t[6] = c[6];
        if (code_size == 8) { t[6] = cosf(c[6]); t[7] = sinf(c[6]); } /* encode_angle_by_sincos (cagroup_utils.py:128-130) */
    }
    return CG3D_OK;
}

/* get_dense_grid_points / get_global_grid_points_of_roi (cagroup_roi_head.py:199-224), SimplePoolingLayer.forward :46-68 */
int cg3d_roi_grid_coords(const float *rois, int64_t n, int32_t rois_per_scene, int32_t grid, int32_t with_yaw, float voxel_size,
                         float clamp_lo, float clamp_hi, int32_t coord_key, int32_t *coords, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || rois_per_scene <= 0 || grid <= 0 || !(voxel_size > 0.f)) return CG3D_ERR_ARG;
    const int64_t g3 = (int64_t)grid * grid * grid;
    if (n * g3 == 0) return CG3D_OK;
    if (!rois || !coords) return CG3D_ERR_ARG;
    const float fg = (float)grid;
    for (int64_t r = 0; r < n; r++) {
        const float *p = rois + r * 7;
        const float cs = with_yaw ? cosf(p[6]) : 1.f, sn = with_yaw ? sinf(p[6]) : 0.f;
        for (int ix = 0; ix < grid; ix++)
            for (int iy = 0; iy < grid; iy++)
                for (int iz = 0; iz < grid; iz++) {
                    float lx = ((float)ix + 0.5f) / fg * p[3] - p[3] / 2;
                    float ly = ((float)iy + 0.5f) / fg * p[4] - p[4] / 2;
                    const float lz = ((float)iz + 0.5f) / fg * p[5] - p[5] / 2;
                    if (with_yaw) {
                        const float x = lx * cs + ly * (-sn) + lz * 0.f, y = lx * sn + ly * cs + lz * 0.f;
                        lx = x; ly = y;
                    }
                    const float q[3] = {lx + p[0], ly + p[1], lz + p[2]};
                    int32_t *o = coords + (r * g3 + ((int64_t)ix * grid + iy) * grid + iz) * 4;
                    o[0] = (int32_t)(r / rois_per_scene);
                    for (int k = 0; k < 3; k++) {
                        float f = floorf(q[k] / voxel_size);
                        f = f < clamp_lo ? clamp_lo : f;
                        f = f > clamp_hi ? clamp_hi : f;
                        o[1 + k] = (int32_t)f * coord_key;
                    }
                }
    }
    return CG3D_OK;
}

/* get_box_reg_layer_loss (cagroup_roi_head.py:551-590) + WeightedSmoothL1Loss (loss_utils.py:76-137); sums in double */
static float os_diff(const float *reg, const float *target, const float *code_w, int64_t e, int cs) {
    const float t = target[e], x = reg[e];
    float d = (t != t) ? 0.f : x - t;
    if (code_w) d = d * code_w[e % cs];
    return d;
}
int cg3d_roi_reg_loss_fwd(const float *reg, const float *target, const int64_t *valid, const float *code_w, int64_t m,
                          int32_t cs, float beta, float weight, float *out, cg3d_stream_t stream) {
    (void)stream;
    if (m < 0 || cs <= 0 || !out) return CG3D_ERR_ARG;
    if (m > 0 && (!reg || !target || !valid)) return CG3D_ERR_ARG;
    double s = 0.0;
    float cnt = 0.f;
    for (int64_t r = 0; r < m; r++) {
        if (valid[r] <= 0) continue;
        cnt += 1.f;
        for (int k = 0; k < cs; k++) {
            const float n = fabsf(os_diff(reg, target, code_w, r * cs + k, cs));
            s += (beta < 1e-5f) ? n : (n < beta ? 0.5f * n * n / beta : n - 0.5f * beta);
        }
    }
    out[0] = (float)s / (cnt < 1.f ? 1.f : cnt) * weight;
    out[1] = cnt;
    return CG3D_OK;
}
int cg3d_roi_reg_loss_bwd(const float *reg, const float *target, const int64_t *valid, const float *code_w, int64_t m,
                          int32_t cs, float beta, float weight, const float *fwd_out, const float *g, float *dreg,
                          cg3d_stream_t stream) {
    (void)stream;
    if (m < 0 || cs <= 0) return CG3D_ERR_ARG;
    if (m == 0) return CG3D_OK;
    if (!reg || !target || !valid || !fwd_out || !g || !dreg) return CG3D_ERR_ARG;
    const float cnt = fwd_out[1];
    for (int64_t e = 0; e < m * cs; e++) {
        float r = 0.f;
        if (valid[e / cs] > 0 && target[e] == target[e]) {
            const float d = os_diff(reg, target, code_w, e, cs), n = fabsf(d);
            float dl = (beta < 1e-5f || n >= beta) ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : d / beta;
            if (code_w) dl = dl * code_w[e % cs];
            r = g[0] * weight / (cnt < 1.f ? 1.f : cnt) * dl;
        }
        dreg[e] = r;
    }
    return CG3D_OK;
}

/* ================================================================================================ dense head: class rows
 * cagroup_head.py:209-258 (selection :227-233, [votes ; originals] :234-252, the two quantisations :254-271) */
#define OS_CR_BLK 1024
int32_t cg3d_class_nblk(int64_t n) { return (int32_t)(((n > 0 ? n : 1) + OS_CR_BLK - 1) / OS_CR_BLK); }

int cg3d_class_count(const uint8_t *hit, int64_t n, int32_t nc, const int32_t *coords, int32_t *block_off, int32_t *totals,
                     cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || nc <= 0 || nc > 65535 || !block_off || !totals) return CG3D_ERR_ARG;
    if (n > 0 && (!hit || !coords)) return CG3D_ERR_ARG;
    const int nblk = cg3d_class_nblk(n);
    for (int c = 0; c < nc; c++) {
        int run = 0;
        for (int blk = 0; blk < nblk; blk++) {
            block_off[(int64_t)c * nblk + blk] = run;
            const int64_t r1 = (int64_t)(blk + 1) * OS_CR_BLK < n ? (int64_t)(blk + 1) * OS_CR_BLK : n;
            for (int64_t r = (int64_t)blk * OS_CR_BLK; r < r1; r++) run += hit[r * nc + c] != 0;
        }
        totals[c] = run;
    }
    for (int k = 0; k < 3; k++) {
        int lo = 0x7fffffff, hi = (int)0x80000000;
        for (int64_t r = 0; r < n; r++) {
            const int v = coords[r * 4 + 1 + k];
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
        totals[nc + k] = lo;
        totals[nc + 3 + k] = hi;
    }
    return CG3D_OK;
}

int cg3d_class_rows(const uint8_t *hit, int64_t n, int32_t nc, int32_t nbatch, const int32_t *block_off, const int32_t *totals,
                    const int32_t *coords, const int32_t *pad_row, const float *offsets, int32_t nvote, float voxel_size,
                    int32_t ts, const float *vs_tab, int32_t expand, int32_t *src, int32_t *fine, int32_t *coarse,
                    cg3d_stream_t stream) {
    (void)stream; (void)block_off;
    if (n <= 0 || nc <= 0 || nc > 65535 || nbatch <= 0 || nbatch > OS_CR_BLK || nvote < 1 || expand < 1) return CG3D_ERR_ARG;
    if (!hit || !totals || !coords || !pad_row || !offsets || !vs_tab || !src || !fine || !coarse) return CG3D_ERR_ARG;
    if (n * (int64_t)(nvote + 1) >= 0x7fffffffLL) return CG3D_ERR_ARG;
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) {                       /* scene bounds, :209-212 */
        lo[k] = (float)(totals[nc + k] - ts) * voxel_size;
        hi[k] = (float)(totals[nc + 3 + k] + ts) * voxel_size;
    }
    const float fe = (float)expand;
    int64_t start = 0;
    for (int c = 0; c < nc; c++) {
        const int n_c = totals[c] + nbatch;
        const int64_t base = start * (nvote + 1);
        int j = 0;
        for (int64_t it = 0; it < n + nbatch; it++) {
            int64_t r;
            if (it < n) {
                if (!hit[it * nc + c]) continue;
                r = it;
            } else {
                r = pad_row[it - n];                    /* one pad voxel per scene, after the selection (:231-233) */
            }
            const int32_t *q = coords + r * 4;
            const float *vs = vs_tab + c * 3;
            for (int v = 0; v <= nvote; v++) {
                float p[3];
                int64_t dst;
                for (int k = 0; k < 3; k++) {
                    const float ori = (float)q[1 + k] * voxel_size;
                    if (v < nvote) {
                        float t = ori + offsets[r * (nvote * 3) + v * 3 + k];
                        t = t < hi[k] ? t : hi[k];
                        p[k] = t > lo[k] ? t : lo[k];
                    } else {
                        p[k] = ori;
                    }
                }
                if (v < nvote) { dst = base + (int64_t)j * nvote + v; src[dst] = (int32_t)(r * nvote + v); }
                else { dst = base + (int64_t)n_c * nvote + j; src[dst] = (int32_t)(n * nvote + r); }
                fine[dst * 4] = coarse[dst * 4] = c * nbatch + q[0];
                for (int k = 0; k < 3; k++) {
                    fine[dst * 4 + 1 + k] = (int32_t)floorf(p[k] / vs[k]);
                    coarse[dst * 4 + 1 + k] = (int32_t)(floorf(p[k] / (vs[k] * fe)) * fe);
                }
            }
            j++;
        }
        start += n_c;
    }
    return CG3D_OK;
}

int cg3d_gather_rows2(const float *Fa, const float *Fb, int64_t na, const int32_t *idx, float *out, int64_t n, int32_t c,
                      cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || c < 4 || c % 4 != 0 || na < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!Fa || !Fb || !idx || !out) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) {
        const int64_t r = idx[i];
        memcpy(out + i * c, r < na ? Fa + r * c : Fb + (r - na) * c, (size_t)c * sizeof(float));
    }
    return CG3D_OK;
}
int cg3d_scatter_add_rows2(const float *dout, const int32_t *idx, float *dFa, float *dFb, int64_t na, int64_t n, int32_t c,
                           cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || c < 1 || na < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!dout || !idx || !dFa || !dFb) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) {
        const int64_t r = idx[i];
        float *d = r < na ? dFa + r * c : dFb + (r - na) * c;
        for (int a = 0; a < c; a++) d[a] += dout[i * c + a];
    }
    return CG3D_OK;
}

static int os_count_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts, int check) {
    if (n < 0 || m <= 0 || m > 8192 || stride < 1 || !counts) return CG3D_ERR_ARG;
    memset(counts, 0, (size_t)(m + check) * 8);
    if (n == 0) return CG3D_OK;
    if (!ids) return CG3D_ERR_ARG;
    int64_t prev = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t v = is64 ? ((const int64_t *)ids)[i * stride] : (int64_t)((const int32_t *)ids)[i * stride];
        if (v >= 0 && v < m) counts[v]++;
        else if (check) counts[m]++;
        if (check && i > 0 && prev > v) counts[m]++;
        prev = v;
    }
    return CG3D_OK;
}
int cg3d_count_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts, cg3d_stream_t stream) {
    (void)stream;
    return os_count_ids(ids, n, stride, is64, m, counts, 0);
}
int cg3d_count_sorted_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts, cg3d_stream_t stream) {
    (void)stream;
    return os_count_ids(ids, n, stride, is64, m, counts, 1);
}

/* ================================================================================================ dense head: proposals
 * _get_bboxes_single (cagroup_head.py:579-624), _bbox_pred_to_bbox (:654-703), _nms entry selection (:747-770) */
static uint32_t os_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int cg3d_prop_keys(const int64_t *seg, const float *smax, int64_t n, int64_t *keys, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!seg || !smax || !keys) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) keys[i] = (int64_t)(((uint64_t)seg[i] << 32) | (uint64_t)(uint32_t)~os_bits(smax[i]));
    return CG3D_OK;
}
static int os_seg_of(const int32_t *cand_off, int nseg, int t) {
    int s = 0;
    while (s + 1 < nseg && cand_off[s + 1] <= t) s++;
    return s;
}
int cg3d_prop_entries(const int64_t *order, const int32_t *seg_start, const int32_t *cand_off, int32_t nseg, int32_t ncand,
                      int32_t nbatch, const float *scores, int32_t nc, float thr, int64_t *ekeys, int32_t *counts,
                      cg3d_stream_t stream) {
    (void)stream;
    if (nseg < 0 || ncand < 0 || nbatch <= 0 || nc <= 0 || (int64_t)nbatch * nc >= 512 || (int64_t)ncand * nc >= (1ll << 22) || !counts)
        return CG3D_ERR_ARG;
    memset(counts, 0, ((size_t)nbatch * nc + 1) * 4);
    if (ncand == 0 || nseg == 0) return CG3D_OK;
    if (!order || !seg_start || !cand_off || !scores || !ekeys) return CG3D_ERR_ARG;
    int64_t slot = 0;
    for (int t = 0; t < ncand; t++) {
        const int s = os_seg_of(cand_off, nseg, t);
        const int64_t row = order[seg_start[s] + t - cand_off[s]];
        for (int i = 0; i < nc; i++) {
            const float sc = scores[row * nc + i];
            if (!(sc > thr)) continue;                                  /* scores[:, i] > SCORE_THR (:763) */
            const int p = (s % nbatch) * nc + i;
            ekeys[slot++] = (int64_t)(((uint64_t)p << 54) | ((uint64_t)(uint32_t)~os_bits(sc) << 22) | (uint64_t)((int64_t)t * nc + i));
            counts[p]++;
        }
    }
    counts[nbatch * nc] = (int32_t)slot;
    return CG3D_OK;
}
int cg3d_prop_gather(const int64_t *ekeys, int64_t total, const int64_t *order, const int32_t *seg_start,
                     const int32_t *cand_off, int32_t nseg, int32_t nc, const float *points, const float *bbox_pred,
                     int32_t ndim, const float *scores, float *e_boxes, float *nms_boxes, float *e_score, cg3d_stream_t stream) {
    (void)stream;
    if (total < 0 || nseg < 0 || nc <= 0 || (ndim != 6 && ndim != 8)) return CG3D_ERR_ARG;
    if (total == 0) return CG3D_OK;
    if (!ekeys || !order || !seg_start || !cand_off || !points || !bbox_pred || !scores || !e_boxes || !nms_boxes || !e_score)
        return CG3D_ERR_ARG;
    for (int64_t k = 0; k < total; k++) {
        const int e = (int)((uint64_t)ekeys[k] & ((1ull << 22) - 1));
        const int t = e / nc, i = e - t * nc;
        const int s = os_seg_of(cand_off, nseg, t);
        const int64_t row = order[seg_start[s] + t - cand_off[s]];
        const float *p = points + row * 3, *b = bbox_pred + row * ndim;
        float o[7];
        o[0] = p[0] + (b[1] - b[0]) / 2; o[1] = p[1] + (b[3] - b[2]) / 2; o[2] = p[2] + (b[5] - b[4]) / 2;
        if (ndim == 6) {
            o[3] = b[0] + b[1]; o[4] = b[2] + b[3]; o[5] = b[4] + b[5]; o[6] = 0.f;
        } else {
            const float scale = b[0] + b[1] + b[2] + b[3];
            const float q = expf(sqrtf(b[6] * b[6] + b[7] * b[7]));
            o[3] = scale / (1 + q); o[4] = scale / (1 + q) * q; o[5] = b[5] + b[4];
            o[6] = 0.5f * atan2f(b[6], b[7]);
        }
        for (int c = 0; c < 7; c++) { e_boxes[k * 7 + c] = o[c]; nms_boxes[k * 7 + c] = (c == 6 && ndim == 8) ? o[c] * -1.f : o[c]; }
        e_score[k] = scores[row * nc + i];
    }
    return CG3D_OK;
}

/* ================================================================================================ rotated 3D IoU of box pairs
 * forward: the reference's expressions in fp32; backward: central differences of the same function evaluated in double
 * (an independent derivative: the product differentiates analytically). */
#define REAL float
#define RN(x) ri_##x##_f
#define RSIN sinf
#define RCOS cosf
#include "oracle_rotiou.inc"
#undef REAL
#undef RN
#undef RSIN
#undef RCOS
#define REAL double
#define RN(x) ri_##x##_d
#define RSIN sin
#define RCOS cos
#include "oracle_rotiou.inc"
#undef REAL
#undef RN
#undef RSIN
#undef RCOS

int cg3d_rotated_iou3d_fwd(const float *pred, const float *target, int64_t n, float *iou, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!pred || !target || !iou) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) iou[i] = ri_iou3d_f(pred + i * 7, target + i * 7);
    return CG3D_OK;
}
int cg3d_rotated_iou3d_bwd(const float *pred, const float *target, int64_t n, const float *g, float *dpred, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!pred || !target || !g || !dpred) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) {
        double p[7], q[7];
        for (int k = 0; k < 7; k++) { p[k] = pred[i * 7 + k]; q[k] = target[i * 7 + k]; }
        for (int k = 0; k < 7; k++) {
            const double h = 1e-6 * (fabs(p[k]) > 1.0 ? fabs(p[k]) : 1.0), keep = p[k];
            p[k] = keep + h;
            const double fp = ri_iou3d_d(p, q);
            p[k] = keep - h;
            const double fm = ri_iou3d_d(p, q);
            p[k] = keep;
            dpred[i * 7 + k] = (float)((double)g[i] * (fp - fm) / (2.0 * h));
        }
    }
    return CG3D_OK;
}

/* ================================================================================================ vote targets, ScanNet form
 * cagroup_head.py:454-498: per instance the bounding box of its points (:462-468), non-object instances voted far away
 * (:466), objects -> the centre of the nearest ground-truth box (:470-478); per voxel the instance of its nearest raw point
 * (:480-486) and the masked offset (:488-494). */
int cg3d_instance_centers(const float *xyz, const int64_t *ins, const int64_t *sem, int32_t nb, int32_t np, int32_t ni,
                          const float *gt_ctr, int32_t gmax, const int32_t *n_gt, int32_t n_classes, float *centers, int32_t *ws,
                          cg3d_stream_t stream) {
    (void)stream; (void)ws;
    if (nb <= 0 || np <= 0 || ni <= 0 || gmax < 0) return CG3D_ERR_ARG;
    if (!xyz || !ins || !sem || !gt_ctr || !n_gt || !centers) return CG3D_ERR_ARG;
    for (int b = 0; b < nb; b++)
        for (int id = 0; id < ni; id++) {
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            int first = np;
            for (int p = 0; p < np; p++) {
                if (ins[(int64_t)b * np + p] != id) continue;
                if (first == np) first = p;
                for (int k = 0; k < 3; k++) {
                    const float v = xyz[((int64_t)b * np + p) * 3 + k];
                    lo[k] = v < lo[k] ? v : lo[k];
                    hi[k] = v > hi[k] ? v : hi[k];
                }
            }
            float *o = centers + ((int64_t)b * ni + id) * 3;
            o[0] = o[1] = o[2] = 0.f;
            if (first == np) continue;
            if (!(sem[(int64_t)b * np + first] < n_classes)) { o[0] = o[1] = o[2] = -10000.f; continue; }
            float best = INFINITY;
            int arg = 0;
            for (int g = 0; g < n_gt[b]; g++) {
                const float *q = gt_ctr + ((int64_t)b * gmax + g) * 3;
                const float dx = 0.5f * (lo[0] + hi[0]) - q[0], dy = 0.5f * (lo[1] + hi[1]) - q[1], dz = 0.5f * (lo[2] + hi[2]) - q[2];
                const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                if (d < best) { best = d; arg = g; }
            }
            const float *q = gt_ctr + ((int64_t)b * gmax + arg) * 3;
            o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
        }
    return CG3D_OK;
}
int cg3d_vote_targets(const float *vox_xyz, const int64_t *vox_scene, const int64_t *nearest, int64_t n, const int64_t *ins,
                      int32_t np, const float *centers, int32_t ni, float *off_t, float *off_m, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || np <= 0 || ni <= 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!vox_xyz || !vox_scene || !nearest || !ins || !centers || !off_t || !off_m) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) {
        const int64_t b = vox_scene[i], major = ins[b * np + nearest[i]];
        const float *c = centers + (b * ni + major) * 3;
        int all = 1;
        for (int k = 0; k < 3; k++) {
            const float t = c[k] - vox_xyz[i * 3 + k];
            const int real = !(t < -100.f);
            all = all && real;
            off_t[i * 3 + k] = real ? t : 0.f;
        }
        off_m[i] = all ? 1.f : 0.f;
    }
    return CG3D_OK;
}

/* ================================================================================================ positives loss, yaw form
 * cagroup_head.py:532-546 with _bbox_pred_to_bbox 'fcaf3d' (:689-703): BCE in double like oracle_loss.c; the box term's
 * gradient by central differences (double) of b -> 1 - iou3d(decode(p, b), t): independent of the product's analytic chain. */
static void os_decode8_f(const float *p, const float *b, float *o) {
    o[0] = p[0] + (b[1] - b[0]) / 2; o[1] = p[1] + (b[3] - b[2]) / 2; o[2] = p[2] + (b[5] - b[4]) / 2;
    const float scale = b[0] + b[1] + b[2] + b[3];
    const float q = expf(sqrtf(b[6] * b[6] + b[7] * b[7]));
    o[3] = scale / (1 + q); o[4] = scale / (1 + q) * q; o[5] = b[5] + b[4];
    o[6] = 0.5f * atan2f(b[6], b[7]);
}
static double os_box_term_d(const double *p, const double *b, const double *t) {
    double o[7];
    o[0] = p[0] + (b[1] - b[0]) / 2; o[1] = p[1] + (b[3] - b[2]) / 2; o[2] = p[2] + (b[5] - b[4]) / 2;
    const double scale = b[0] + b[1] + b[2] + b[3], q = exp(sqrt(b[6] * b[6] + b[7] * b[7]));
    o[3] = scale / (1 + q); o[4] = scale / (1 + q) * q; o[5] = b[5] + b[4];
    o[6] = 0.5 * atan2(b[6], b[7]);
    return 1.0 - ri_iou3d_d(o, t);
}
int32_t cg3d_pos_loss_yaw_nblocks(int64_t npos) {
    const int64_t nb = (npos + 63) / 64;
    return (int32_t)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}
int cg3d_pos_loss_yaw_fwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                          const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                          const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                          float *partial, cg3d_stream_t stream) {
    (void)stream;
    if (npos < 0 || tstride < 7 || !partial) return CG3D_ERR_ARG;
    const int nb = cg3d_pos_loss_yaw_nblocks(npos);
    memset(partial, 0, (size_t)nb * 2 * sizeof(float));
    double s0 = 0.0, s1 = 0.0;
    for (int64_t i = 0; i < npos; i++) {
        const int64_t r = pos[i], sc = scene[r];
        const double x = centerness[r], ct = ctr_t[r];
        const double bce = fmax(x, 0.0) - x * ct + log1p(exp(-fabs(x)));
        float box[7];
        os_decode8_f(points + r * 3, bbox_pred + r * 8, box);
        const float iou = ri_iou3d_f(box, bbox_t + r * tstride);
        s0 += bce * ((double)wc / ((double)n_pos[sc] + eps));
        s1 += (1.0 - (double)iou) * ((double)wb * ct / (double)ctr_denorm[sc]);
    }
    partial[0] = (float)s0; partial[1] = (float)s1;
    return CG3D_OK;
}
int cg3d_pos_loss_yaw_bwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                          const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                          const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                          const float *gscale, float *dcenterness, float *dbbox_pred, cg3d_stream_t stream) {
    (void)stream;
    if (npos < 0 || tstride < 7) return CG3D_ERR_ARG;
    if (npos == 0) return CG3D_OK;
    for (int64_t i = 0; i < npos; i++) {
        const int64_t r = pos[i], sc = scene[r];
        const double x = centerness[r], ct = ctr_t[r];
        dcenterness[r] = (float)((double)gscale[0] * (1.0 / (1.0 + exp(-x)) - ct) * ((double)wc / ((double)n_pos[sc] + eps)));
        double p[3], b[8], t[7];
        for (int k = 0; k < 3; k++) p[k] = points[r * 3 + k];
        for (int k = 0; k < 8; k++) b[k] = bbox_pred[r * 8 + k];
        for (int k = 0; k < 7; k++) t[k] = bbox_t[r * tstride + k];
        const double g = (double)gscale[1] * ((double)wb * ct / (double)ctr_denorm[sc]);
        for (int k = 0; k < 8; k++) {
            const double h = 1e-6 * (fabs(b[k]) > 1.0 ? fabs(b[k]) : 1.0), keep = b[k];
            b[k] = keep + h;
            const double fp = os_box_term_d(p, b, t);
            b[k] = keep - h;
            const double fm = os_box_term_d(p, b, t);
            b[k] = keep;
            dbbox_pred[r * 8 + k] = (float)(g * (fp - fm) / (2.0 * h));
        }
    }
    return CG3D_OK;
}

/* ================================================================================================ class-branch outputs
 * forward_single, cagroup_head.py:627-652 (Scale + exp :640-645, points :647-650) for all class maps at once */
int cg3d_head_outputs_fwd(const float *reg, int32_t nd, const int32_t *coords, int64_t n, int32_t nbatch, const float *scale,
                          const float *vs_tab, int32_t nc, float boost, float *cls, float *bbox_pred, float *points,
                          cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || nd < 6 || nbatch <= 0 || nc <= 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!reg || !coords || !scale || !vs_tab || !bbox_pred || !points) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) {
        const int c = coords[i * 4] / nbatch;
        for (int k = 0; k < nd; k++) bbox_pred[i * nd + k] = k < 6 ? expf(reg[i * nd + k] * scale[c]) : reg[i * nd + k];
        for (int k = 0; k < 3; k++) points[i * 3 + k] = (float)coords[i * 4 + 1 + k] * vs_tab[c * 3 + k];
        if (boost != 0.f && cls) cls[i * nc + c] = cls[i * nc + c] + boost;
    }
    return CG3D_OK;
}
int cg3d_head_outputs_bwd(const float *dbbox, const float *bbox_pred, const float *reg, int32_t nd, const int32_t *coords, int64_t n,
                          int32_t nbatch, const float *scale, int32_t nc, float *dreg, float *dscale, cg3d_stream_t stream) {
    (void)stream;
    if (n < 0 || nd < 6 || nbatch <= 0 || nc <= 0 || nc > 4096 || !dscale) return CG3D_ERR_ARG;
    double acc[4096];
    for (int k = 0; k < nc; k++) acc[k] = 0.0;
    if (n > 0 && (!dbbox || !bbox_pred || !reg || !coords || !scale || !dreg)) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < n; i++) {
        const int c = coords[i * 4] / nbatch;
        for (int k = 0; k < nd; k++) {
            const float g = dbbox[i * nd + k];
            if (k < 6) {
                const float gb = g * bbox_pred[i * nd + k];
                dreg[i * nd + k] = gb * scale[c];
                acc[c] += (double)gb * reg[i * nd + k];
            } else {
                dreg[i * nd + k] = g;
            }
        }
    }
    for (int k = 0; k < nc; k++) dscale[k] = (float)acc[k];
    return CG3D_OK;
}
