// stages.hip -- stage-level fused operators of the two heads (include/cagroup3d_stages.h).
//
// Each entry point replaces a chain of tens of tensor launches of the reference (RoI <-> ground-truth matching, target
// construction, grid coordinates, the regression loss; class-row construction and proposal decoding of the dense head) by
// one pass with the same arithmetic in the same operation order.  Built with -ffp-contract=off: what is integer or a
// comparison of fp32 values comes out bit-identical to the CPU oracle (oracle/oracle_stages.c).
#include "dg_geom.h"
#include "../../include/cagroup3d_stages.h"

// ------------------------------------------------------------------------------------------------ helpers
// torch.remainder(a, b) for floats: fmod, moved into the sign of the divisor
__host__ __device__ static inline float st_remainder(float a, float b) {
    float m = fmodf(a, b);
    if (m != 0.f && ((b < 0.f) != (m < 0.f))) m += b;
    return m;
}
#define ST_2PI 6.283185307179586f
#define ST_PI 3.141592653589793f

// padded RoI `i` of scene `b` (reoder_rois_for_refining + the enlargement of forward_train): roi[7], label, score
__device__ static inline void st_load_roi(const float *__restrict__ boxes, const float *__restrict__ scores,
                                          const int64_t *__restrict__ labels, const int32_t *__restrict__ roi_off, int b, int i,
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
#pragma unroll
        for (int k = 0; k < 6; k++) roi[k] = 0.f;
        roi[6] = -0.f;
        *label = 0;
        if (score) *score = 0.f;
    }
}

// boxes_iou3d_gpu (iou3d_nms_utils.py:59-79), its operation order
__device__ static inline float st_iou3d(const float *a, const float *b) {
    const float a_hmax = a[2] + a[5] / 2, a_hmin = a[2] - a[5] / 2;
    const float b_hmax = b[2] + b[5] / 2, b_hmin = b[2] - b[5] / 2;
    const float ob = d_box_overlap(a, b);
    float oh = fminf(a_hmax, b_hmax) - fmaxf(a_hmin, b_hmin);
    oh = oh < 0.f ? 0.f : oh;
    const float o3 = ob * oh;
    const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
    float u = va + vb - o3;
    u = u < 1e-6f ? 1e-6f : u;
    return o3 / u;
}

// ------------------------------------------------------------------------------------------------ RoI <-> GT matching
// 16 lanes per RoI walk the scene's boxes; (value, index) max with the lowest index on ties.
__global__ __launch_bounds__(256) void k_roi_match(const float *__restrict__ boxes, const int64_t *__restrict__ labels,
                                                   const int32_t *__restrict__ roi_off, int nb, int rin, float enlarge,
                                                   const float *__restrict__ gt_boxes, int gmax, int gdim,
                                                   const int32_t *__restrict__ n_gt, float *__restrict__ max_ov,
                                                   int32_t *__restrict__ assign) {
    const int r = blockIdx.x * 16 + (threadIdx.x >> 4), lane = threadIdx.x & 15;
    const bool live = r < nb * rin;
    const int b = live ? r / rin : 0, i = live ? r % rin : 0;
    float roi[7];
    int64_t label;
    st_load_roi(boxes, nullptr, labels, roi_off, b, i, enlarge, roi, &label, nullptr);
    const int ng = live ? n_gt[b] : 0;
    float best = -1.f;
    int arg = 0x7fffffff;
    for (int g = lane; g < ng; g += 16) {
        const float *q = gt_boxes + ((int64_t)b * gmax + g) * gdim;
        if ((int64_t)q[7] != label) continue;
        float gb[7] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6] * -1.f};
        const float v = st_iou3d(roi, gb);
        if (v > best) { best = v; arg = g; }          // g ascends per lane: the first maximum stays
    }
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        const float ob = __shfl_xor(best, d, 16);
        const int oa = __shfl_xor(arg, d, 16);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (live && lane == 0) {
        const bool has = best >= 0.f;
        max_ov[r] = has ? best : 0.f;
        assign[r] = has ? arg : 0;
    }
}
extern "C" int cg3d_roi_match(const float *boxes, const int64_t *labels, const int32_t *roi_off, int32_t nb, int32_t rin,
                              float enlarge, const float *gt_boxes, int32_t gmax, int32_t gdim, const int32_t *n_gt,
                              float *max_ov, int32_t *assign, cg3d_stream_t stream) {
    if (nb < 0 || rin < 0 || gmax < 0 || gdim < 8) return CG3D_ERR_ARG;
    const int64_t n = (int64_t)nb * rin;
    if (n == 0) return CG3D_OK;
    if (!roi_off || !n_gt || !max_ov || !assign) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_roi_match, dim3((unsigned)cg3d_divup(n, 16)), dim3(256), 0, cg3d_hs(stream), boxes, labels, roi_off, nb,
                       rin, enlarge, gt_boxes, gmax, gdim, n_gt, max_ov, assign);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ sampled RoIs -> targets
__global__ __launch_bounds__(128) void k_roi_targets(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                     const int64_t *__restrict__ labels, const int32_t *__restrict__ roi_off,
                                                     int nb, int rin, float enlarge, const float *__restrict__ gt_boxes, int gmax,
                                                     int gdim, const float *__restrict__ max_ov,
                                                     const int32_t *__restrict__ assign, const int32_t *__restrict__ keep,
                                                     int rsel, int code_size, float reg_fg, float cls_fg, float cls_bg,
                                                     float cls_span, float *__restrict__ o_rois, float *__restrict__ o_gt_src,
                                                     float *__restrict__ o_gt, float *__restrict__ o_gt_label,
                                                     float *__restrict__ o_iou, float *__restrict__ o_score,
                                                     int64_t *__restrict__ o_label, int64_t *__restrict__ o_reg_valid,
                                                     float *__restrict__ o_cls_label, float *__restrict__ o_reg_target) {
    const int j = blockIdx.x * 128 + threadIdx.x;
    if (j >= nb * rsel) return;
    const int b = j / rsel, src = keep[j];
    float roi[7], score;
    int64_t label;
    st_load_roi(boxes, scores, labels, roi_off, b, src, enlarge, roi, &label, &score);
    const int64_t pr = (int64_t)b * rin + src;
    const float iou = max_ov[pr];
    const float *q = gt_boxes + ((int64_t)b * gmax + assign[pr]) * gdim;
    float g[7] = {q[0], q[1], q[2], q[3], q[4], q[5], q[6] * -1.f};
#pragma unroll
    for (int k = 0; k < 7; k++) { o_rois[(int64_t)j * 7 + k] = roi[k]; o_gt_src[(int64_t)j * 7 + k] = g[k]; }
    o_gt_label[j] = (float)(int64_t)q[7];
    o_iou[j] = iou;
    o_score[j] = score;
    o_label[j] = label;
    o_reg_valid[j] = iou > reg_fg ? 1 : 0;
    const bool fg = iou > cls_fg, bg = iou < cls_bg;
    o_cls_label[j] = (!fg && !bg) ? (iou - cls_bg) / cls_span : (fg ? 1.f : 0.f);
    // canonical frame of the RoI (assign_targets, cagroup_roi_head.py:300-324)
    const float ry = st_remainder(roi[6], ST_2PI);
    float c[7];
    c[0] = g[0] - roi[0]; c[1] = g[1] - roi[1]; c[2] = g[2] - roi[2];
    c[3] = g[3]; c[4] = g[4]; c[5] = g[5];
    c[6] = st_remainder(g[6], ST_2PI) - ry;
    if (code_size > 6) {
        const float ang = -ry, cs = cosf(ang), sn = sinf(ang);
        const float x = c[0] * cs + c[1] * (-sn) + c[2] * 0.f, y = c[0] * sn + c[1] * cs + c[2] * 0.f;
        c[0] = x; c[1] = y;
        float h = st_remainder(c[6], ST_2PI);
        if (h > ST_PI * 0.5f && h < ST_PI * 1.5f) h = st_remainder(h + ST_PI, ST_2PI);
        if (h > ST_PI) h = h - ST_2PI;
        c[6] = fminf(fmaxf(h, -ST_PI / 2), ST_PI / 2);
    }
    // regression targets: encode_torch against the RoI with centre (and heading) zeroed; sizes clamped in place like there
    const float a3 = fmaxf(roi[3], 1e-5f), a4 = fmaxf(roi[4], 1e-5f), a5 = fmaxf(roi[5], 1e-5f);
    c[3] = fmaxf(c[3], 1e-5f); c[4] = fmaxf(c[4], 1e-5f); c[5] = fmaxf(c[5], 1e-5f);
#pragma unroll
    for (int k = 0; k < 7; k++) o_gt[(int64_t)j * 7 + k] = c[k];
    const float diag = sqrtf(a3 * a3 + a4 * a4);
    float *t = o_reg_target + (int64_t)j * code_size;
    t[0] = c[0] / diag; t[1] = c[1] / diag; t[2] = c[2] / a5;
    t[3] = logf(c[3] / a3); t[4] = logf(c[4] / a4); t[5] = logf(c[5] / a5);
    if (code_size == 7) t[6] = c[6];
    if (code_size == 8) { t[6] = cosf(c[6]); t[7] = sinf(c[6]); }      /* encode_angle_by_sincos (cagroup_utils.py:128-130) */
}
extern "C" int cg3d_roi_targets(const float *boxes, const float *scores, const int64_t *labels, const int32_t *roi_off,
                                int32_t nb, int32_t rin, float enlarge, const float *gt_boxes, int32_t gmax, int32_t gdim,
                                const float *max_ov, const int32_t *assign, const int32_t *keep, int32_t rsel, int32_t code_size,
                                float reg_fg, float cls_fg, float cls_bg, float cls_span, float *o_rois, float *o_gt_src,
                                float *o_gt, float *o_gt_label, float *o_iou, float *o_score, int64_t *o_label,
                                int64_t *o_reg_valid, float *o_cls_label, float *o_reg_target, cg3d_stream_t stream) {
    if (nb < 0 || rin < 0 || rsel < 0 || gdim < 8 || (code_size < 6 || code_size > 8)) return CG3D_ERR_ARG;
    const int64_t m = (int64_t)nb * rsel;
    if (m == 0) return CG3D_OK;
    if (!roi_off || !gt_boxes || !max_ov || !assign || !keep || !o_rois || !o_gt_src || !o_gt || !o_gt_label || !o_iou ||
        !o_score || !o_label || !o_reg_valid || !o_cls_label || !o_reg_target)
        return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_roi_targets, dim3((unsigned)cg3d_divup(m, 128)), dim3(128), 0, cg3d_hs(stream), boxes, scores, labels,
                       roi_off, nb, rin, enlarge, gt_boxes, gmax, gdim, max_ov, assign, keep, rsel, code_size, reg_fg, cls_fg,
                       cls_bg, cls_span, o_rois, o_gt_src, o_gt, o_gt_label, o_iou, o_score, o_label, o_reg_valid, o_cls_label,
                       o_reg_target);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ RoI grid coordinates
__global__ __launch_bounds__(256) void k_roi_grid_coords(const float *__restrict__ rois, int64_t total, int rps, int grid,
                                                         int with_yaw, float vs, float lo, float hi, int ck,
                                                         int32_t *__restrict__ coords) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int g3 = grid * grid * grid;
    const int64_t r = t / g3;
    const int g = (int)(t - r * g3);
    const int ix = g / (grid * grid), iy = (g / grid) % grid, iz = g % grid;
    const float *p = rois + r * 7;
    const float fg = (float)grid;
    float lx = ((float)ix + 0.5f) / fg * p[3] - p[3] / 2;
    float ly = ((float)iy + 0.5f) / fg * p[4] - p[4] / 2;
    const float lz = ((float)iz + 0.5f) / fg * p[5] - p[5] / 2;
    if (with_yaw) {
        const float cs = cosf(p[6]), sn = sinf(p[6]);
        const float x = lx * cs + ly * (-sn) + lz * 0.f, y = lx * sn + ly * cs + lz * 0.f;
        lx = x; ly = y;
    }
    const float q[3] = {lx + p[0], ly + p[1], lz + p[2]};
    int4 o;
    o.x = (int)(r / rps);
    int v[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float f = floorf(q[k] / vs);
        f = f < lo ? lo : f;           // torch.clamp(min, max): max(min(x, hi), lo) with NaN propagating; finite inputs here
        f = f > hi ? hi : f;
        v[k] = (int)f * ck;
    }
    o.y = v[0]; o.z = v[1]; o.w = v[2];
    reinterpret_cast<int4 *>(coords)[t] = o;
}
extern "C" int cg3d_roi_grid_coords(const float *rois, int64_t n, int32_t rois_per_scene, int32_t grid, int32_t with_yaw,
                                    float voxel_size, float clamp_lo, float clamp_hi, int32_t coord_key, int32_t *coords,
                                    cg3d_stream_t stream) {
    if (n < 0 || rois_per_scene <= 0 || grid <= 0 || !(voxel_size > 0.f)) return CG3D_ERR_ARG;
    const int64_t total = n * grid * grid * grid;
    if (total == 0) return CG3D_OK;
    if (!rois || !coords) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_roi_grid_coords, dim3((unsigned)cg3d_divup(total, 256)), dim3(256), 0, cg3d_hs(stream), rois, total,
                       rois_per_scene, grid, with_yaw, voxel_size, clamp_lo, clamp_hi, coord_key, coords);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ RoI regression loss
__device__ static inline float st_sl1_diff(const float *__restrict__ reg, const float *__restrict__ target,
                                           const float *__restrict__ code_w, int64_t e, int cs) {
    const float t = target[e], x = reg[e];
    float d = (t != t) ? 0.f : x - t;
    if (code_w) d = d * code_w[e % cs];
    return d;
}
__global__ __launch_bounds__(256) void k_roi_reg_loss_fwd(const float *__restrict__ reg, const float *__restrict__ target,
                                                          const int64_t *__restrict__ valid, const float *__restrict__ code_w,
                                                          int64_t m, int cs, float beta, float weight, float *__restrict__ out) {
    __shared__ float ssum[256], scnt[256];
    float s = 0.f, c = 0.f;
    for (int64_t e = threadIdx.x; e < m * cs; e += 256) {
        const int64_t row = e / cs;
        if (valid[row] <= 0) continue;
        if (e % cs == 0) c += 1.f;
        const float n = fabsf(st_sl1_diff(reg, target, code_w, e, cs));
        s += (beta < 1e-5f) ? n : (n < beta ? 0.5f * n * n / beta : n - 0.5f * beta);
    }
    ssum[threadIdx.x] = s; scnt[threadIdx.x] = c;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) { ssum[threadIdx.x] += ssum[threadIdx.x + d]; scnt[threadIdx.x] += scnt[threadIdx.x + d]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float cnt = scnt[0];
        out[0] = ssum[0] / (cnt < 1.f ? 1.f : cnt) * weight;
        out[1] = cnt;
    }
}
__global__ __launch_bounds__(256) void k_roi_reg_loss_bwd(const float *__restrict__ reg, const float *__restrict__ target,
                                                          const int64_t *__restrict__ valid, const float *__restrict__ code_w,
                                                          int64_t m, int cs, float beta, float weight,
                                                          const float *__restrict__ fwd_out, const float *__restrict__ g,
                                                          float *__restrict__ dreg) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= m * cs) return;
    float r = 0.f;
    if (valid[e / cs] > 0 && target[e] == target[e]) {
        const float cnt = fwd_out[1];
        const float d = st_sl1_diff(reg, target, code_w, e, cs), n = fabsf(d);
        float dl = (beta < 1e-5f || n >= beta) ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : d / beta;
        if (code_w) dl = dl * code_w[e % cs];
        r = g[0] * weight / (cnt < 1.f ? 1.f : cnt) * dl;
    }
    dreg[e] = r;
}
extern "C" int cg3d_roi_reg_loss_fwd(const float *reg, const float *target, const int64_t *valid, const float *code_w, int64_t m,
                                     int32_t cs, float beta, float weight, float *out, cg3d_stream_t stream) {
    if (m < 0 || cs <= 0 || !out) return CG3D_ERR_ARG;
    if (m > 0 && (!reg || !target || !valid)) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_roi_reg_loss_fwd, dim3(1), dim3(256), 0, cg3d_hs(stream), reg, target, valid, code_w, m, cs, beta,
                       weight, out);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_roi_reg_loss_bwd(const float *reg, const float *target, const int64_t *valid, const float *code_w, int64_t m,
                                     int32_t cs, float beta, float weight, const float *fwd_out, const float *g, float *dreg,
                                     cg3d_stream_t stream) {
    if (m < 0 || cs <= 0) return CG3D_ERR_ARG;
    if (m == 0) return CG3D_OK;
    if (!reg || !target || !valid || !fwd_out || !g || !dreg) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_roi_reg_loss_bwd, dim3((unsigned)cg3d_divup(m * cs, 256)), dim3(256), 0, cg3d_hs(stream), reg, target,
                       valid, code_w, m, cs, beta, weight, fwd_out, g, dreg);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ================================================================================================ dense head: class rows
#define CR_BLK 1024          // voxels per block of the ordered compaction

extern "C" int32_t cg3d_class_nblk(int64_t n) { return (int32_t)cg3d_divup(n > 0 ? n : 1, CR_BLK); }

// selected voxels of (class blockIdx.y, block blockIdx.x); blocks of class 0 also reduce the coordinate columns
__global__ __launch_bounds__(CR_BLK) void k_class_count(const uint8_t *__restrict__ hit, int64_t n, int nc,
                                                        const int32_t *__restrict__ coords, int nblk,
                                                        int32_t *__restrict__ block_cnt, int32_t *__restrict__ block_mm) {
    __shared__ int s_cnt[CR_BLK / 64];
    __shared__ int s_mm[CR_BLK / 64][6];
    const int c = blockIdx.y, blk = blockIdx.x;
    const int64_t r = (int64_t)blk * CR_BLK + threadIdx.x;
    const bool h = r < n && hit[r * nc + c] != 0;
    const unsigned long long bal = __ballot(h);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    if (c == 0) {
        int v[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, (int)0x80000000, (int)0x80000000, (int)0x80000000};
        if (r < n) {
            const int4 q = reinterpret_cast<const int4 *>(coords)[r];
            v[0] = v[3] = q.y; v[1] = v[4] = q.z; v[2] = v[5] = q.w;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                v[k] = min(v[k], __shfl_xor(v[k], d));
                v[3 + k] = max(v[3 + k], __shfl_xor(v[3 + k], d));
            }
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 6; k++) s_mm[wave][k] = v[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < CR_BLK / 64; w++) t += s_cnt[w];
        block_cnt[c * nblk + blk] = t;
    }
    if (c == 0 && threadIdx.x < 6) {
        int m = s_mm[0][threadIdx.x];
        for (int w = 1; w < CR_BLK / 64; w++) m = threadIdx.x < 3 ? min(m, s_mm[w][threadIdx.x]) : max(m, s_mm[w][threadIdx.x]);
        block_mm[blk * 6 + threadIdx.x] = m;
    }
}
// one workgroup per class: exclusive prefix of its block counts (in place) and its total; workgroup nc reduces the bounds
__global__ __launch_bounds__(256) void k_class_scan(int32_t *__restrict__ block_cnt, const int32_t *__restrict__ block_mm, int nc,
                                                    int nblk, int32_t *__restrict__ totals) {
    __shared__ int s[256];
    const int c = blockIdx.x;
    if (c == nc) {
        if (threadIdx.x < 6) {
            const bool lo = threadIdx.x < 3;
            int m = lo ? 0x7fffffff : (int)0x80000000;
            for (int b = 0; b < nblk; b++) m = lo ? min(m, block_mm[b * 6 + threadIdx.x]) : max(m, block_mm[b * 6 + threadIdx.x]);
            totals[nc + threadIdx.x] = m;
        }
        return;
    }
    int32_t *row = block_cnt + (int64_t)c * nblk;
    int carry = 0;
    for (int base = 0; base < nblk; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < nblk ? row[i] : 0;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int t = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < nblk) row[i] = carry + s[threadIdx.x] - v;
        carry += s[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[c] = carry;
}
extern "C" int cg3d_class_count(const uint8_t *hit, int64_t n, int32_t nc, const int32_t *coords, int32_t *block_off,
                                int32_t *totals, cg3d_stream_t stream) {
    if (n < 0 || nc <= 0 || nc > 65535 || !block_off || !totals) return CG3D_ERR_ARG;
    if (n > 0 && (!hit || !coords)) return CG3D_ERR_ARG;
    const int nblk = cg3d_class_nblk(n);
    // the per-block bounds wait behind the counts in the same buffer: block_off must hold nc * nblk + 6 * nblk int32
    int32_t *block_mm = block_off + (int64_t)nc * nblk;
    hipLaunchKernelGGL(k_class_count, dim3(nblk, nc), dim3(CR_BLK), 0, cg3d_hs(stream), hit, n, nc, coords, nblk, block_off, block_mm);
    hipLaunchKernelGGL(k_class_scan, dim3(nc + 1), dim3(256), 0, cg3d_hs(stream), block_off, block_mm, nc, nblk, totals);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

__global__ __launch_bounds__(CR_BLK) void k_class_rows(const uint8_t *__restrict__ hit, int64_t n, int nc, int nbatch, int nblk,
                                                       const int32_t *__restrict__ block_off, const int32_t *__restrict__ totals,
                                                       const int32_t *__restrict__ coords, const int32_t *__restrict__ pad_row,
                                                       const float *__restrict__ offsets, int nvote, float voxel_size, int ts,
                                                       const float *__restrict__ vs_tab, int expand, int32_t *__restrict__ src,
                                                       int32_t *__restrict__ fine, int32_t *__restrict__ coarse) {
    __shared__ int s_cnt[CR_BLK / 64];
    const int c = blockIdx.y, blk = blockIdx.x;
    const bool pads = blk == nblk;                      // the last block of a class places its pad voxels
    int64_t r = (int64_t)blk * CR_BLK + threadIdx.x;
    bool h;
    if (pads) {
        h = (int)threadIdx.x < nbatch;
        r = h ? pad_row[threadIdx.x] : 0;
    } else {
        h = r < n && hit[r * nc + c] != 0;
    }
    const unsigned long long bal = __ballot(h);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) s_cnt[wave] = __popcll(bal);
    __syncthreads();
    if (!h) return;
    int before = __popcll(bal & ((1ull << lane) - 1));
    for (int w = 0; w < wave; w++) before += s_cnt[w];
    // class geometry of the list: start_c rows of earlier classes, n_c rows of this one (selected + pads)
    int start = 0;
    for (int k = 0; k < c; k++) start += totals[k] + nbatch;
    const int n_c = totals[c] + nbatch;
    const int j = pads ? totals[c] + before : block_off[c * nblk + blk] + before;
    const int64_t base = (int64_t)start * (nvote + 1);
    const int4 q = reinterpret_cast<const int4 *>(coords)[r];
    const int bp = c * nbatch + q.x;
    const float vs[3] = {vs_tab[c * 3 + 0], vs_tab[c * 3 + 1], vs_tab[c * 3 + 2]};
    const float ori[3] = {(float)q.y * voxel_size, (float)q.z * voxel_size, (float)q.w * voxel_size};
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        lo[k] = (float)(totals[nc + k] - ts) * voxel_size;
        hi[k] = (float)(totals[nc + 3 + k] + ts) * voxel_size;
    }
    for (int v = 0; v <= nvote; v++) {
        float p[3];
        int64_t dst;
        int s;
        if (v < nvote) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                float t = ori[k] + offsets[r * (nvote * 3) + v * 3 + k];
                t = t < hi[k] ? t : hi[k];            // torch.max(torch.min(voted, max_bound), min_bound)
                p[k] = t > lo[k] ? t : lo[k];
            }
            dst = base + (int64_t)j * nvote + v;
            s = (int)(r * nvote + v);
        } else {
            p[0] = ori[0]; p[1] = ori[1]; p[2] = ori[2];
            dst = base + (int64_t)n_c * nvote + j;
            s = (int)(n * nvote + r);
        }
        src[dst] = s;
        int4 f, g;
        f.x = g.x = bp;
        const float fe = (float)expand;
        f.y = (int)floorf(p[0] / vs[0]); f.z = (int)floorf(p[1] / vs[1]); f.w = (int)floorf(p[2] / vs[2]);
        g.y = (int)(floorf(p[0] / (vs[0] * fe)) * fe); g.z = (int)(floorf(p[1] / (vs[1] * fe)) * fe);
        g.w = (int)(floorf(p[2] / (vs[2] * fe)) * fe);
        reinterpret_cast<int4 *>(fine)[dst] = f;
        reinterpret_cast<int4 *>(coarse)[dst] = g;
    }
}
extern "C" int cg3d_class_rows(const uint8_t *hit, int64_t n, int32_t nc, int32_t nbatch, const int32_t *block_off,
                               const int32_t *totals, const int32_t *coords, const int32_t *pad_row, const float *offsets,
                               int32_t nvote, float voxel_size, int32_t ts, const float *vs_tab, int32_t expand, int32_t *src,
                               int32_t *fine, int32_t *coarse, cg3d_stream_t stream) {
    if (n <= 0 || nc <= 0 || nc > 65535 || nbatch <= 0 || nbatch > CR_BLK || nvote < 1 || expand < 1) return CG3D_ERR_ARG;
    if (!hit || !block_off || !totals || !coords || !pad_row || !offsets || !vs_tab || !src || !fine || !coarse) return CG3D_ERR_ARG;
    if (n * (int64_t)(nvote + 1) >= 0x7fffffffLL) return CG3D_ERR_ARG;
    const int nblk = cg3d_class_nblk(n);
    hipLaunchKernelGGL(k_class_rows, dim3(nblk + 1, nc), dim3(CR_BLK), 0, cg3d_hs(stream), hit, n, nc, nbatch, nblk, block_off,
                       totals, coords, pad_row, offsets, nvote, voxel_size, ts, vs_tab, expand, src, fine, coarse);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ two-piece gather
__global__ void k_gather_rows2(const float *__restrict__ Fa, const float *__restrict__ Fb, int64_t na,
                               const int32_t *__restrict__ idx, float *__restrict__ out, int64_t n, int cq) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i = t / cq;
    const int g = (int)(t % cq);
    if (i >= n) return;
    const int64_t r = idx[i];
    const float4 *s = r < na ? reinterpret_cast<const float4 *>(Fa) + r * cq : reinterpret_cast<const float4 *>(Fb) + (r - na) * cq;
    reinterpret_cast<float4 *>(out)[i * cq + g] = s[g];
}
extern "C" int cg3d_gather_rows2(const float *Fa, const float *Fb, int64_t na, const int32_t *idx, float *out, int64_t n,
                                 int32_t c, cg3d_stream_t stream) {
    if (n < 0 || c < 4 || c % 4 != 0 || na < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!Fa || !Fb || !idx || !out || (((uintptr_t)Fa | (uintptr_t)Fb | (uintptr_t)out) & 15)) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_gather_rows2, dim3((unsigned)cg3d_divup(n * (c / 4), 256)), dim3(256), 0, cg3d_hs(stream), Fa, Fb, na, idx,
                       out, n, c / 4);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
// one channel per lane, like k_scatter_add_rows (gather_scatter.hip): a wave's atomics land on consecutive floats
__global__ void k_scatter_add_rows2(const float *__restrict__ dout, const int32_t *__restrict__ idx, float *__restrict__ dFa,
                                    float *__restrict__ dFb, int64_t na, int64_t n, int c) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t i = t / c;
    const int a = (int)(t - i * c);
    if (i >= n) return;
    const int64_t r = idx[i];
    float *d = r < na ? dFa + r * c : dFb + (r - na) * c;
    unsafeAtomicAdd(d + a, dout[i * c + a]);
}
extern "C" int cg3d_scatter_add_rows2(const float *dout, const int32_t *idx, float *dFa, float *dFb, int64_t na, int64_t n,
                                      int32_t c, cg3d_stream_t stream) {
    if (n < 0 || c < 1 || na < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!dout || !idx || !dFa || !dFb) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_scatter_add_rows2, dim3((unsigned)cg3d_divup(n * c, 256)), dim3(256), 0, cg3d_hs(stream), dout, idx, dFa,
                       dFb, na, n, c);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ------------------------------------------------------------------------------------------------ id histogram
template <typename T>
__global__ __launch_bounds__(256) void k_count_ids(const T *__restrict__ ids, int64_t n, int stride, int m,
                                                   unsigned long long *__restrict__ counts, int check) {
    extern __shared__ int s_h[];
    for (int i = threadIdx.x; i < m + 1; i += 256) s_h[i] = 0;
    __syncthreads();
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t v = (int64_t)ids[i * stride];
        if (v >= 0 && v < m) atomicAdd(&s_h[v], 1);
        else bad++;
        if (check && i > 0 && (int64_t)ids[(i - 1) * stride] > v) bad++;
    }
    if (check && bad) atomicAdd(&s_h[m], bad);
    __syncthreads();
    for (int i = threadIdx.x; i < m + (check ? 1 : 0); i += 256)
        if (s_h[i]) atomicAdd(&counts[i], (unsigned long long)s_h[i]);
}
static int st_count_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts, int check,
                        cg3d_stream_t stream) {
    if (n < 0 || m <= 0 || m > 8192 || stride < 1 || !counts) return CG3D_ERR_ARG;
    if (hipMemsetAsync(counts, 0, (size_t)(m + check) * 8, cg3d_hs(stream)) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (n == 0) return CG3D_OK;
    if (!ids) return CG3D_ERR_ARG;
    const unsigned g = (unsigned)(cg3d_divup(n, 256 * 8) < 1024 ? cg3d_divup(n, 256 * 8) : 1024);
    if (is64) hipLaunchKernelGGL(k_count_ids<int64_t>, dim3(g), dim3(256), (size_t)(m + 1) * 4, cg3d_hs(stream), (const int64_t *)ids, n, stride, m, (unsigned long long *)counts, check);
    else hipLaunchKernelGGL(k_count_ids<int32_t>, dim3(g), dim3(256), (size_t)(m + 1) * 4, cg3d_hs(stream), (const int32_t *)ids, n, stride, m, (unsigned long long *)counts, check);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_count_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts,
                              cg3d_stream_t stream) {
    return st_count_ids(ids, n, stride, is64, m, counts, 0, stream);
}
extern "C" int cg3d_count_sorted_ids(const void *ids, int64_t n, int32_t stride, int32_t is64, int32_t m, int64_t *counts,
                                     cg3d_stream_t stream) {
    return st_count_ids(ids, n, stride, is64, m, counts, 1, stream);
}

// ================================================================================================ dense head: proposals
__global__ void k_prop_keys(const int64_t *__restrict__ seg, const float *__restrict__ smax, int64_t n,
                            int64_t *__restrict__ keys) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t b = ~__float_as_uint(smax[i]);
    keys[i] = (int64_t)(((uint64_t)seg[i] << 32) | (uint64_t)b);
}
extern "C" int cg3d_prop_keys(const int64_t *seg, const float *smax, int64_t n, int64_t *keys, cg3d_stream_t stream) {
    if (n < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!seg || !smax || !keys) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_prop_keys, dim3((unsigned)cg3d_divup(n, 256)), dim3(256), 0, cg3d_hs(stream), seg, smax, n, keys);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// segment of candidate t: the last s with cand_off[s] <= t
__device__ static inline int st_seg_of(const int32_t *__restrict__ cand_off, int nseg, int t) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (cand_off[mid] <= t) lo = mid; else hi = mid - 1;
    }
    return lo;
}
// A workgroup takes PE_PER entries per thread: one slot-range atomic and one histogram flush per workgroup (a slot atomic per
// wave and a counter atomic per entry were 20 k + 72 k atomics on 73 addresses: 129 us).
#define PE_PER 8
__global__ __launch_bounds__(256) void k_prop_entries(const int64_t *__restrict__ order, const int32_t *__restrict__ seg_start,
                                                      const int32_t *__restrict__ cand_off, int nseg, int nbatch,
                                                      const float *__restrict__ scores, int nc, float thr,
                                                      unsigned long long *__restrict__ ekeys, int32_t *__restrict__ counts) {
    __shared__ int s_hist[512];
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int ncand = cand_off[nseg], np = nbatch * nc;
    for (int i = threadIdx.x; i < np; i += 256) s_hist[i] = 0;
    __syncthreads();
    unsigned long long key[PE_PER];
    int mine = 0;
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * PE_PER;
    int s = -1, t_of_s = -1;
#pragma unroll
    for (int q = 0; q < PE_PER; q++) {
        const int64_t e = e0 + q;
        key[q] = 0;
        if (e < (int64_t)ncand * nc) {
            const int t = (int)(e / nc), i = (int)(e - (int64_t)t * nc);
            if (t != t_of_s) { s = st_seg_of(cand_off, nseg, t); t_of_s = t; }
            const int64_t row = order[seg_start[s] + t - cand_off[s]];
            const float sc = scores[row * nc + i];
            if (sc > thr) {
                const int p = (s % nbatch) * nc + i;
                key[q] = ((unsigned long long)p << 54) | ((unsigned long long)(~__float_as_uint(sc)) << 22) | (unsigned long long)e | (1ull << 63);
                atomicAdd(&s_hist[p], 1);
                mine++;
            }
        }
    }
    // exclusive prefix of `mine` over the workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int before = incl - mine;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    if (threadIdx.x == 0) {
        const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = tot ? atomicAdd(&counts[np], tot) : 0;
    }
    __syncthreads();
    int64_t dst = (int64_t)s_base + before;
#pragma unroll
    for (int q = 0; q < PE_PER; q++)
        if (key[q]) ekeys[dst++] = key[q] & ~(1ull << 63);
    for (int i = threadIdx.x; i < np; i += 256)
        if (s_hist[i]) atomicAdd(&counts[i], s_hist[i]);
}
extern "C" int cg3d_prop_entries(const int64_t *order, const int32_t *seg_start, const int32_t *cand_off, int32_t nseg,
                                 int32_t ncand, int32_t nbatch, const float *scores, int32_t nc, float thr, int64_t *ekeys,
                                 int32_t *counts, cg3d_stream_t stream) {
    if (nseg < 0 || ncand < 0 || nbatch <= 0 || nc <= 0 || (int64_t)nbatch * nc >= 512 || (int64_t)ncand * nc >= (1ll << 22) || !counts)
        return CG3D_ERR_ARG;
    if (hipMemsetAsync(counts, 0, ((size_t)nbatch * nc + 1) * 4, cg3d_hs(stream)) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (ncand == 0 || nseg == 0) return CG3D_OK;
    if (!order || !seg_start || !cand_off || !scores || !ekeys) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_prop_entries, dim3((unsigned)cg3d_divup((int64_t)ncand * nc, 256 * PE_PER)), dim3(256), 0, cg3d_hs(stream),
                       order, seg_start, cand_off, nseg, nbatch, scores, nc, thr, (unsigned long long *)ekeys, counts);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

__global__ __launch_bounds__(256) void k_prop_gather(const unsigned long long *__restrict__ ekeys, int64_t total,
                                                     const int64_t *__restrict__ order, const int32_t *__restrict__ seg_start,
                                                     const int32_t *__restrict__ cand_off, int nseg, int nc,
                                                     const float *__restrict__ points, const float *__restrict__ bbox_pred, int ndim,
                                                     const float *__restrict__ scores, float *__restrict__ e_boxes,
                                                     float *__restrict__ nms_boxes, float *__restrict__ e_score) {
    const int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (k >= total) return;
    const int e = (int)(ekeys[k] & ((1ull << 22) - 1));
    const int t = e / nc, i = e - t * nc;
    const int s = st_seg_of(cand_off, nseg, t);
    const int64_t row = order[seg_start[s] + t - cand_off[s]];
    const float *p = points + row * 3, *b = bbox_pred + row * ndim;
    float o[7];
    o[0] = p[0] + (b[1] - b[0]) / 2; o[1] = p[1] + (b[3] - b[2]) / 2; o[2] = p[2] + (b[5] - b[4]) / 2;
    if (ndim == 6) {
        o[3] = b[0] + b[1]; o[4] = b[2] + b[3]; o[5] = b[4] + b[5]; o[6] = 0.f;
    } else {                                            // 'fcaf3d': (sin(2a) ln q, cos(2a) ln q)  (cagroup_head.py:689-703)
        const float scale = b[0] + b[1] + b[2] + b[3];
        const float q = expf(sqrtf(b[6] * b[6] + b[7] * b[7]));
        o[3] = scale / (1 + q); o[4] = scale / (1 + q) * q; o[5] = b[5] + b[4];
        o[6] = 0.5f * atan2f(b[6], b[7]);
    }
#pragma unroll
    for (int c = 0; c < 7; c++) { e_boxes[k * 7 + c] = o[c]; nms_boxes[k * 7 + c] = (c == 6 && ndim == 8) ? o[c] * -1.f : o[c]; }
    e_score[k] = scores[row * nc + i];
}
extern "C" int cg3d_prop_gather(const int64_t *ekeys, int64_t total, const int64_t *order, const int32_t *seg_start,
                                const int32_t *cand_off, int32_t nseg, int32_t nc, const float *points, const float *bbox_pred,
                                int32_t ndim, const float *scores, float *e_boxes, float *nms_boxes, float *e_score,
                                cg3d_stream_t stream) {
    if (total < 0 || nseg < 0 || nc <= 0 || (ndim != 6 && ndim != 8)) return CG3D_ERR_ARG;
    if (total == 0) return CG3D_OK;
    if (!ekeys || !order || !seg_start || !cand_off || !points || !bbox_pred || !scores || !e_boxes || !nms_boxes || !e_score)
        return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_prop_gather, dim3((unsigned)cg3d_divup(total, 256)), dim3(256), 0, cg3d_hs(stream),
                       (const unsigned long long *)ekeys, total, order, seg_start, cand_off, nseg, nc, points, bbox_pred, ndim, scores,
                       e_boxes, nms_boxes, e_score);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ================================================================================================ rotated 3D IoU, fused
// One thread per box pair; the reference's tensor expressions in their operation order (fp32, no contraction).
#define RI_EPS 1e-8f
struct RiPair {
    float c1x[4], c1y[4], c2x[4], c2y[4];       // corners (box2corners_th)
    float vx[24], vy[24];                        // candidate vertices: 4 + 4 corners, 16 edge intersections (i major)
    float t2[16];                                // intersection parameter along the edge of box 1
    unsigned mask;
    int idx[9];
    float S;                                     // signed shoelace sum
};
__device__ static inline bool ri_cmp_vert(float x1, float y1, float x2, float y2) {      // sort_vert_kernel.cu:15-40
    const double E = 1e-8;
    if ((double)fabsf(x1 - x2) < E && (double)fabsf(y2 - y1) < E) return false;
    if (y1 > 0 && y2 < 0) return true;
    if (y1 < 0 && y2 > 0) return false;
    float n1 = (float)((double)(x1 * x1 + y1 * y1) + E);
    float n2 = (float)((double)(x2 * x2 + y2 * y2) + E);
    if (y1 > 0 && y2 > 0) return (double)(fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2) > E;
    if (y1 < 0 && y2 < 0) return (double)(fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2) < E;
    return false;
}
__device__ static inline void ri_corners(float x, float y, float w, float h, float a, float *cx, float *cy) {
    const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f}, sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
    const float sn = sinf(a), cs = cosf(a);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const float x4 = sx[c] * w, y4 = sy[c] * h;
        cx[c] = (x4 * cs + y4 * (-sn)) + x;
        cy[c] = (x4 * sn + y4 * cs) + y;
    }
}
__device__ static inline bool ri_in(float px, float py, const float *qx, const float *qy) {       // box1_in_box2, :57-82
    const float abx = qx[1] - qx[0], aby = qy[1] - qy[0], adx = qx[3] - qx[0], ady = qy[3] - qy[0];
    const float amx = px - qx[0], amy = py - qy[0];
    const float p_ab = abx * amx + aby * amy, n_ab = abx * abx + aby * aby;
    const float p_ad = adx * amx + ady * amy, n_ad = adx * adx + ady * ady;
    const float ra = p_ab / n_ab, rd = p_ad / n_ad;
    return ra > -1e-6f && ra < 1.000001f && rd > -1e-6f && rd < 1.000001f;
}
// 2D part: b = (x, y, w, h, alpha).  Returns the intersection area.
__device__ static float ri_area(const float *b1, const float *b2, RiPair &P) {
    ri_corners(b1[0], b1[1], b1[2], b1[3], b1[4], P.c1x, P.c1y);
    ri_corners(b2[0], b2[1], b2[2], b2[3], b2[4], P.c2x, P.c2y);
    unsigned mask = 0u;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        P.vx[k] = P.c1x[k]; P.vy[k] = P.c1y[k]; P.vx[4 + k] = P.c2x[k]; P.vy[4 + k] = P.c2y[k];
        if (ri_in(P.c1x[k], P.c1y[k], P.c2x, P.c2y)) mask |= 1u << k;
        if (ri_in(P.c2x[k], P.c2y[k], P.c1x, P.c1y)) mask |= 1u << (4 + k);
    }
    for (int i = 0; i < 4; i++) {
        const float x1 = P.c1x[i], y1 = P.c1y[i], x2 = P.c1x[(i + 1) & 3], y2 = P.c1y[(i + 1) & 3];
        for (int j = 0; j < 4; j++) {
            const float x3 = P.c2x[j], y3 = P.c2y[j], x4 = P.c2x[(j + 1) & 3], y4 = P.c2y[(j + 1) & 3];
            const float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
            const float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
            const float den_u = (x1 - x2) * (y1 - y3) - (y1 - y2) * (x1 - x3);
            float t = den_t / num, u = -den_u / num;
            if (num == 0.f) { t = -1.f; u = -1.f; }
            const bool m = (t > 0.f) && (t < 1.f) && (u > 0.f) && (u < 1.f);
            const float t2 = den_t / (num + RI_EPS);
            const int e = i * 4 + j;
            P.t2[e] = t2;
            const float mf = m ? 1.f : 0.f;
            P.vx[8 + e] = (x1 + t2 * (x2 - x1)) * mf;
            P.vy[8 + e] = (y1 + t2 * (y2 - y1)) * mf;
            if (m) mask |= 1u << (8 + e);
        }
    }
    P.mask = mask;
    const int nv = __popc(mask);
    // sort_indices (:127-147) + sort_vertices_forward on the vertices relative to their mean
    float mx = 0.f, my = 0.f;
    for (int k = 0; k < 24; k++)
        if ((mask >> k) & 1u) { mx += P.vx[k]; my += P.vy[k]; }
    mx = mx / (float)nv; my = my / (float)nv;
    int pad = 0;
    for (int j = 8; j < 24; ++j) if (!((mask >> j) & 1u)) { pad = j; break; }
    int *out = P.idx;
    if (nv < 3) {
        for (int j = 0; j < 9; ++j) out[j] = pad;
    } else {
        for (int j = 0; j < 9; ++j) out[j] = 0;
        for (int j = 0; j < nv && j < 9; ++j) {
            float x_min = 1.f, y_min = (float)(-1e-8);
            int i_take = 0;
            float x2 = 0.f, y2 = 0.f;
            if (j > 0) { x2 = P.vx[out[j - 1]] - mx; y2 = P.vy[out[j - 1]] - my; }
            for (int k = 0; k < 24; ++k) {
                if (!((mask >> k) & 1u)) continue;
                const float x = P.vx[k] - mx, y = P.vy[k] - my;
                bool take = ri_cmp_vert(x, y, x_min, y_min);
                if (j > 0) take = take && ri_cmp_vert(x2, y2, x, y);
                if (take) { x_min = x; y_min = y; i_take = k; }
            }
            out[j] = i_take;
        }
        if (nv < 9) out[nv] = out[0];
        for (int j = nv + 1; j < 9; ++j) out[j] = pad;
        if (nv == 8) {
            int counter = 0;
            for (int j = 0; j < 4; ++j)
                for (int k = 4; k < 8; ++k) if (out[k] == out[j]) counter++;
            if (counter == 4) { out[4] = out[0]; for (int j = 5; j < 9; ++j) out[j] = pad; }
        }
    }
    float S = 0.f;                                      // calculate_area, :150-166
    for (int m = 0; m < 8; m++) S += P.vx[out[m]] * P.vy[out[m + 1]] - P.vy[out[m]] * P.vx[out[m + 1]];
    P.S = S;
    return fabsf(S) / 2;
}
struct Ri3d { float area, zo, u3d, inter3d; bool zmax1_lt, zmin1_gt; };
__device__ static float ri_iou3d(const float *p, const float *q, RiPair &P, Ri3d &R) {             // cal_iou_3d, :86-109
    const float b1[5] = {p[0], p[1], p[3], p[4], p[6]}, b2[5] = {q[0], q[1], q[3], q[4], q[6]};
    const float zmax1 = p[2] + p[5] * 0.5f, zmin1 = p[2] - p[5] * 0.5f, zmax2 = q[2] + q[5] * 0.5f, zmin2 = q[2] - q[5] * 0.5f;
    float zo = fminf(zmax1, zmax2) - fmaxf(zmin1, zmin2);
    zo = zo < 0.f ? 0.f : zo;
    const float area = ri_area(b1, b2, P);
    const float u = b1[2] * b1[3] + b2[2] * b2[3] - area;
    const float iou2d = area / u;
    const float inter3d = iou2d * u * zo;
    const float v1 = p[3] * p[4] * p[5], v2 = q[3] * q[4] * q[5];
    const float u3d = v1 + v2 - inter3d;
    R.area = area; R.zo = zo; R.u3d = u3d; R.inter3d = inter3d; R.zmax1_lt = zmax1 < zmax2; R.zmin1_gt = zmin1 > zmin2;
    return inter3d / u3d;
}
__global__ __launch_bounds__(64) void k_rotated_iou3d_fwd(const float *__restrict__ pred, const float *__restrict__ target, int64_t n,
                                                          float *__restrict__ iou) {
    const int64_t i = blockIdx.x * (int64_t)64 + threadIdx.x;
    if (i >= n) return;
    float p[7], q[7];
#pragma unroll
    for (int k = 0; k < 7; k++) { p[k] = pred[i * 7 + k]; q[k] = target[i * 7 + k]; }
    RiPair P;
    Ri3d R;
    iou[i] = ri_iou3d(p, q, P, R);
}
// d[7] = go * d iou3d(p, q) / d p  (the gradient along the reference's autograd graph; see cg3d_rotated_iou3d_bwd).  Returns iou.
__device__ static float ri_backward(const float *p, const float *q, float go, float *d) {
    RiPair P;
    Ri3d R;
    const float iou_ = ri_iou3d(p, q, P, R);
    // iou = I / U, U = v1 + v2 - I, I = area * zo (the reference forms it as (area / u) * u * zo: the u's cancel in the gradient)
    const float I = R.inter3d, U = R.u3d;
    const float gI = go * (U + I) / (U * U), gv1 = -go * I / (U * U);
    const float gA = gI * R.zo, gzo = R.zo > 0.f ? gI * R.area : 0.f;
#pragma unroll
    for (int k = 0; k < 7; k++) d[k] = 0.f;
    // height overlap: min(zmax1, zmax2) - max(zmin1, zmin2), z +- l / 2
    d[2] += gzo * ((R.zmax1_lt ? 1.f : 0.f) - (R.zmin1_gt ? 1.f : 0.f));
    d[5] += gzo * 0.5f * ((R.zmax1_lt ? 1.f : 0.f) + (R.zmin1_gt ? 1.f : 0.f));
    // volume of the predicted box
    d[3] += gv1 * p[4] * p[5]; d[4] += gv1 * p[3] * p[5]; d[5] += gv1 * p[3] * p[4];
    // area = |S| / 2 -> the selected vertices
    const float gS = gA * (P.S > 0.f ? 0.5f : (P.S < 0.f ? -0.5f : 0.f));
    float gvx[24], gvy[24];
    for (int k = 0; k < 24; k++) { gvx[k] = 0.f; gvy[k] = 0.f; }
    for (int m = 0; m < 8; m++) {
        const int a = P.idx[m], b = P.idx[m + 1];
        gvx[a] += gS * P.vy[b]; gvy[a] -= gS * P.vx[b];
        gvy[b] += gS * P.vx[a]; gvx[b] -= gS * P.vy[a];
    }
    // vertices -> corners of box 1 (corners directly; intersections through the edge parameter)
    float gcx[4], gcy[4];
    for (int k = 0; k < 4; k++) { gcx[k] = gvx[k]; gcy[k] = gvy[k]; }
    for (int ii = 0; ii < 4; ii++) {
        const int i2 = (ii + 1) & 3;
        const float x1 = P.c1x[ii], y1 = P.c1y[ii], x2 = P.c1x[i2], y2 = P.c1y[i2];
        for (int j = 0; j < 4; j++) {
            const int e = ii * 4 + j;
            if (!((P.mask >> (8 + e)) & 1u)) continue;
            const float gx = gvx[8 + e], gy = gvy[8 + e];
            if (gx == 0.f && gy == 0.f) continue;
            const float x3 = P.c2x[j], y3 = P.c2y[j], x4 = P.c2x[(j + 1) & 3], y4 = P.c2y[(j + 1) & 3];
            const float num = (x1 - x2) * (y3 - y4) - (y1 - y2) * (x3 - x4);
            const float den_t = (x1 - x3) * (y3 - y4) - (y1 - y3) * (x3 - x4);
            const float D = num + RI_EPS, t2 = P.t2[e];
            const float a34y = y3 - y4, a34x = x3 - x4;
            const float gt = gx * (x2 - x1) + gy * (y2 - y1);
            // d t2 / d q = (d den_t / d q * D - den_t * d num / d q) / D^2
            const float iD2 = 1.f / (D * D);
            const float dt_x1 = (a34y * D - den_t * a34y) * iD2, dt_y1 = (-a34x * D + den_t * a34x) * iD2;
            const float dt_x2 = (den_t * a34y) * iD2, dt_y2 = (-den_t * a34x) * iD2;
            gcx[ii] += gx * (1.f - t2) + gt * dt_x1; gcy[ii] += gy * (1.f - t2) + gt * dt_y1;
            gcx[i2] += gx * t2 + gt * dt_x2;         gcy[i2] += gy * t2 + gt * dt_y2;
        }
    }
    // corners -> (x, y, w, h, alpha)
    {
        const float sx[4] = {0.5f, -0.5f, -0.5f, 0.5f}, sy[4] = {0.5f, 0.5f, -0.5f, -0.5f};
        const float sn = sinf(p[6]), cs = cosf(p[6]);
        for (int c = 0; c < 4; c++) {
            const float x4 = sx[c] * p[3], y4 = sy[c] * p[4];
            d[0] += gcx[c]; d[1] += gcy[c];
            d[3] += gcx[c] * sx[c] * cs + gcy[c] * sx[c] * sn;
            d[4] += -gcx[c] * sy[c] * sn + gcy[c] * sy[c] * cs;
            d[6] += gcx[c] * (-x4 * sn - y4 * cs) + gcy[c] * (x4 * cs - y4 * sn);
        }
    }
    return iou_;
}
__global__ __launch_bounds__(64) void k_rotated_iou3d_bwd(const float *__restrict__ pred, const float *__restrict__ target, int64_t n,
                                                          const float *__restrict__ g, float *__restrict__ dpred) {
    const int64_t i = blockIdx.x * (int64_t)64 + threadIdx.x;
    if (i >= n) return;
    float p[7], q[7], d[7];
#pragma unroll
    for (int k = 0; k < 7; k++) { p[k] = pred[i * 7 + k]; q[k] = target[i * 7 + k]; }
    ri_backward(p, q, g[i], d);
#pragma unroll
    for (int k = 0; k < 7; k++) dpred[i * 7 + k] = d[k];
}
extern "C" int cg3d_rotated_iou3d_fwd(const float *pred, const float *target, int64_t n, float *iou, cg3d_stream_t stream) {
    if (n < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!pred || !target || !iou) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_rotated_iou3d_fwd, dim3((unsigned)cg3d_divup(n, 64)), dim3(64), 0, cg3d_hs(stream), pred, target, n, iou);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_rotated_iou3d_bwd(const float *pred, const float *target, int64_t n, const float *g, float *dpred,
                                      cg3d_stream_t stream) {
    if (n < 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!pred || !target || !g || !dpred) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_rotated_iou3d_bwd, dim3((unsigned)cg3d_divup(n, 64)), dim3(64), 0, cg3d_hs(stream), pred, target, n, g, dpred);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ================================================================================================ vote targets (ScanNet form)
// order-preserving map float -> uint32 (atomicMin / atomicMax on the image order like the floats)
__device__ static inline uint32_t st_fkey(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
// (written with a shift: the `k & 0x80000000u ? k & 0x7fffffffu : ~k` form crashes instruction selection of this compiler)
__device__ static inline float st_funkey(uint32_t k) { return __uint_as_float((k >> 31) ? (k ^ 0x80000000u) : ~k); }

__global__ void k_inst_init(uint32_t *__restrict__ ws, int64_t cells, int np) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= cells * 8) return;
    const int f = (int)(i & 7);
    ws[i] = f < 3 ? 0xffffffffu : (f < 6 ? 0u : (f == 6 ? (uint32_t)np : 0u));       // min keys | max keys | first point | -
}
// A workgroup takes IS_PTS points of ONE scene into a private LDS table (instance ids repeat all over a scene: 200 k points onto
// ~80 cells x 7 words of global atomics serialised at the L2 for over a millisecond) and flushes the cells it touched.
#define IS_PTS 8192
#define IS_MAXI 512
__global__ __launch_bounds__(256) void k_inst_stats(const float *__restrict__ xyz, const int64_t *__restrict__ ins, int nb, int np, int ni,
                                                    uint32_t *__restrict__ ws) {
    __shared__ uint32_t tab[IS_MAXI * 7];
    const int chunks = (np + IS_PTS - 1) / IS_PTS;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const bool lds = ni <= IS_MAXI;
    if (lds) {
        for (int i = threadIdx.x; i < ni * 7; i += 256) {
            const int f = i % 7;
            tab[i] = f < 3 ? 0xffffffffu : (f < 6 ? 0u : (uint32_t)np);
        }
        __syncthreads();
    }
    const int p1 = (ch + 1) * IS_PTS < np ? (ch + 1) * IS_PTS : np;
    for (int pt = ch * IS_PTS + threadIdx.x; pt < p1; pt += 256) {
        const int64_t i = (int64_t)b * np + pt;
        const int64_t id = ins[i];
        if (id < 0 || id >= ni) continue;
        uint32_t *w = lds ? tab + id * 7 : ws + ((int64_t)b * ni + id) * 8;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t key = st_fkey(xyz[i * 3 + k]);
            atomicMin(&w[k], key);
            atomicMax(&w[3 + k], key);
        }
        atomicMin(&w[6], (uint32_t)pt);
    }
    if (!lds) return;
    __syncthreads();
    for (int c = threadIdx.x; c < ni; c += 256) {
        const uint32_t *t = tab + c * 7;
        if (t[6] >= (uint32_t)np) continue;                 // no point of this instance in the chunk
        uint32_t *w = ws + ((int64_t)b * ni + c) * 8;
#pragma unroll
        for (int k = 0; k < 3; k++) { atomicMin(&w[k], t[k]); atomicMax(&w[3 + k], t[3 + k]); }
        atomicMin(&w[6], t[6]);
    }
}
__global__ void k_inst_centers(const uint32_t *__restrict__ ws, const int64_t *__restrict__ sem, int nb, int np, int ni,
                               const float *__restrict__ gt_ctr, int gmax, const int32_t *__restrict__ n_gt, int n_classes,
                               float *__restrict__ centers) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nb * ni) return;
    const int b = c / ni;
    const uint32_t *w = ws + (int64_t)c * 8;
    const int first = (int)w[6];
    float ox = 0.f, oy = 0.f, oz = 0.f;
    if (first < np) {
        const int64_t sid = sem[(int64_t)b * np + first];
        if (sid >= (int64_t)n_classes) {
            ox = -10000.f; oy = -10000.f; oz = -10000.f;
        } else {
            const float cx = 0.5f * (st_funkey(w[0]) + st_funkey(w[3]));
            const float cy = 0.5f * (st_funkey(w[1]) + st_funkey(w[4]));
            const float cz = 0.5f * (st_funkey(w[2]) + st_funkey(w[5]));
            float best = 3.0e38f;
            int arg = 0;
            const int ng = n_gt[b];
            const float *q0 = gt_ctr + (int64_t)b * gmax * 3;
            for (int g = 0; g < ng; g++) {
                const float dx = cx - q0[g * 3], dy = cy - q0[g * 3 + 1], dz = cz - q0[g * 3 + 2];
                const float d = sqrtf(dx * dx + dy * dy + dz * dz);
                if (d < best) { best = d; arg = g; }
            }
            ox = q0[arg * 3]; oy = q0[arg * 3 + 1]; oz = q0[arg * 3 + 2];
        }
    }
    centers[(int64_t)c * 3] = ox; centers[(int64_t)c * 3 + 1] = oy; centers[(int64_t)c * 3 + 2] = oz;
}
extern "C" int cg3d_instance_centers(const float *xyz, const int64_t *ins, const int64_t *sem, int32_t nb, int32_t np, int32_t ni,
                                     const float *gt_ctr, int32_t gmax, const int32_t *n_gt, int32_t n_classes, float *centers,
                                     int32_t *ws, cg3d_stream_t stream) {
    if (nb <= 0 || np <= 0 || ni <= 0 || gmax < 0) return CG3D_ERR_ARG;
    if (!xyz || !ins || !sem || !gt_ctr || !n_gt || !centers || !ws) return CG3D_ERR_ARG;
    const int64_t cells = (int64_t)nb * ni;
    hipLaunchKernelGGL(k_inst_init, dim3((unsigned)cg3d_divup(cells * 8, 256)), dim3(256), 0, cg3d_hs(stream), (uint32_t *)ws, cells, np);
    hipLaunchKernelGGL(k_inst_stats, dim3((unsigned)(nb * cg3d_divup(np, IS_PTS))), dim3(256), 0, cg3d_hs(stream), xyz, ins, nb, np, ni,
                       (uint32_t *)ws);
    hipLaunchKernelGGL(k_inst_centers, dim3((unsigned)cg3d_divup(cells, 64)), dim3(64), 0, cg3d_hs(stream), (const uint32_t *)ws, sem, nb,
                       np, ni, gt_ctr, gmax, n_gt, n_classes, centers);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
__global__ void k_vote_targets(const float *__restrict__ vox_xyz, const int64_t *__restrict__ vox_scene,
                               const int64_t *__restrict__ nearest, int64_t n, const int64_t *__restrict__ ins, int np,
                               const float *__restrict__ centers, int ni, float *__restrict__ off_t, float *__restrict__ off_m) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = vox_scene[i];
    const int64_t major = ins[b * np + nearest[i]];
    const float *c = centers + (b * ni + major) * 3;
    bool all = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float t = c[k] - vox_xyz[i * 3 + k];
        const bool real = !(t < -100.f);
        all = all && real;
        off_t[i * 3 + k] = real ? t : 0.f;
    }
    off_m[i] = all ? 1.f : 0.f;
}
extern "C" int cg3d_vote_targets(const float *vox_xyz, const int64_t *vox_scene, const int64_t *nearest, int64_t n,
                                 const int64_t *ins, int32_t np, const float *centers, int32_t ni, float *off_t, float *off_m,
                                 cg3d_stream_t stream) {
    if (n < 0 || np <= 0 || ni <= 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!vox_xyz || !vox_scene || !nearest || !ins || !centers || !off_t || !off_m) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_vote_targets, dim3((unsigned)cg3d_divup(n, 256)), dim3(256), 0, cg3d_hs(stream), vox_xyz, vox_scene, nearest, n,
                       ins, np, centers, ni, off_t, off_m);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}


// ================================================================================================ positives loss, yaw form
// Centerness BCE + rotated IoU loss over the positive points of the class maps (SUN RGB-D: cagroup_head.py:532-546 with the
// 'fcaf3d' box decode :689-703 and IoU3DLoss / cal_iou_3d): the yaw counterpart of cg3d_pos_loss (loss.hip).
__device__ static inline void st_decode8(const float *p, const float *b, float *o) {
    o[0] = p[0] + (b[1] - b[0]) / 2; o[1] = p[1] + (b[3] - b[2]) / 2; o[2] = p[2] + (b[5] - b[4]) / 2;
    const float scale = b[0] + b[1] + b[2] + b[3];
    const float q = expf(sqrtf(b[6] * b[6] + b[7] * b[7]));
    o[3] = scale / (1 + q); o[4] = scale / (1 + q) * q; o[5] = b[5] + b[4];
    o[6] = 0.5f * atan2f(b[6], b[7]);
}
__global__ __launch_bounds__(64) void k_pos_loss_yaw_fwd(const float *__restrict__ cent, const float *__restrict__ bbox,
                                                         const float *__restrict__ points, const float *__restrict__ ctr_t,
                                                         const float *__restrict__ bbox_t, int tstride,
                                                         const int64_t *__restrict__ scene, const float *__restrict__ n_pos,
                                                         const float *__restrict__ ctr_den, const int64_t *__restrict__ pos,
                                                         int64_t npos, float wc, float wb, float eps, float *__restrict__ partial) {
    float s0 = 0.f, s1 = 0.f;
    for (int64_t i = blockIdx.x * (int64_t)64 + threadIdx.x; i < npos; i += (int64_t)gridDim.x * 64) {
        const int64_t r = pos[i];
        const float pc = cent[r], ct = ctr_t[r];
        const float bce = fmaxf(pc, 0.f) - pc * ct + log1pf(expf(-fabsf(pc)));
        float box[7], t[7];
        st_decode8(points + r * 3, bbox + r * 8, box);
#pragma unroll
        for (int k = 0; k < 7; k++) t[k] = bbox_t[r * tstride + k];
        RiPair P;
        Ri3d R;
        const float iou = ri_iou3d(box, t, P, R);
        const int64_t sc = scene[r];
        s0 += bce * (wc / (n_pos[sc] + eps));
        s1 += (1.f - iou) * (wb * ct / ctr_den[sc]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if (threadIdx.x == 0) { partial[blockIdx.x * 2] = s0; partial[blockIdx.x * 2 + 1] = s1; }
}
__global__ __launch_bounds__(64) void k_pos_loss_yaw_bwd(const float *__restrict__ cent, const float *__restrict__ bbox,
                                                         const float *__restrict__ points, const float *__restrict__ ctr_t,
                                                         const float *__restrict__ bbox_t, int tstride,
                                                         const int64_t *__restrict__ scene, const float *__restrict__ n_pos,
                                                         const float *__restrict__ ctr_den, const int64_t *__restrict__ pos,
                                                         int64_t npos, float wc, float wb, float eps,
                                                         const float *__restrict__ gscale, float *__restrict__ dcent,
                                                         float *__restrict__ dbbox) {
    const int64_t i = blockIdx.x * (int64_t)64 + threadIdx.x;
    if (i >= npos) return;
    const int64_t r = pos[i];
    const float pc = cent[r], ct = ctr_t[r];
    const int64_t sc = scene[r];
    dcent[r] = gscale[0] * (1.f / (1.f + expf(-pc)) - ct) * (wc / (n_pos[sc] + eps));
    const float *b = bbox + r * 8;
    float box[7], t[7], g[7];
    st_decode8(points + r * 3, b, box);
#pragma unroll
    for (int k = 0; k < 7; k++) t[k] = bbox_t[r * tstride + k];
    // d (1 - iou): upstream -gscale[1] * weight on the IoU
    ri_backward(box, t, -gscale[1] * (wb * ct / ctr_den[sc]), g);
    // box -> the eight predictions ('fcaf3d' decode)
    const float scale = b[0] + b[1] + b[2] + b[3];
    const float r2 = b[6] * b[6] + b[7] * b[7], rr = sqrtf(r2), q = expf(rr), iq = 1.f / (1.f + q);
    const float gs = g[3] * iq + g[4] * q * iq, gq = (g[4] - g[3]) * scale * iq * iq;
    float d[8];
    d[0] = -g[0] / 2 + gs; d[1] = g[0] / 2 + gs; d[2] = -g[1] / 2 + gs; d[3] = g[1] / 2 + gs;
    d[4] = -g[2] / 2 + g[5]; d[5] = g[2] / 2 + g[5];
    d[6] = 0.f; d[7] = 0.f;
    if (rr > 0.f) {
        d[6] = gq * q * b[6] / rr + g[6] * 0.5f * b[7] / r2;
        d[7] = gq * q * b[7] / rr - g[6] * 0.5f * b[6] / r2;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) dbbox[r * 8 + k] = d[k];
}
extern "C" int cg3d_pos_loss_yaw_fwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                                     const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                                     const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                                     float *partial, cg3d_stream_t stream) {
    if (npos < 0 || tstride < 7 || !partial) return CG3D_ERR_ARG;
    const int64_t nb = cg3d_divup(npos, 64);
    hipLaunchKernelGGL(k_pos_loss_yaw_fwd, dim3((unsigned)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb))), dim3(64), 0, cg3d_hs(stream),
                       centerness, bbox_pred, points, ctr_t, bbox_t, tstride, scene, n_pos, ctr_denorm, pos, npos, wc, wb, eps, partial);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int32_t cg3d_pos_loss_yaw_nblocks(int64_t npos) {
    const int64_t nb = cg3d_divup(npos, 64);
    return (int32_t)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}
extern "C" int cg3d_pos_loss_yaw_bwd(const float *centerness, const float *bbox_pred, const float *points, const float *ctr_t,
                                     const float *bbox_t, int32_t tstride, const int64_t *scene, const float *n_pos,
                                     const float *ctr_denorm, const int64_t *pos, int64_t npos, float wc, float wb, float eps,
                                     const float *gscale, float *dcenterness, float *dbbox_pred, cg3d_stream_t stream) {
    if (npos < 0 || tstride < 7) return CG3D_ERR_ARG;
    if (npos == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_pos_loss_yaw_bwd, dim3((unsigned)cg3d_divup(npos, 64)), dim3(64), 0, cg3d_hs(stream), centerness, bbox_pred,
                       points, ctr_t, bbox_t, tstride, scene, n_pos, ctr_denorm, pos, npos, wc, wb, eps, gscale, dcenterness,
                       dbbox_pred);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ================================================================================================ class-branch outputs
__global__ __launch_bounds__(256) void k_head_outputs_fwd(const float *__restrict__ reg, int nd, const int32_t *__restrict__ coords,
                                                          int64_t n, int nbatch, const float *__restrict__ scale,
                                                          const float *__restrict__ vs_tab, int nc, float boost,
                                                          float *__restrict__ cls, float *__restrict__ bbox_pred,
                                                          float *__restrict__ points) {
    const int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (i >= n) return;
    const int4 q = reinterpret_cast<const int4 *>(coords)[i];
    const int c = q.x / nbatch;
    const float sc = scale[c];
    for (int k = 0; k < nd; k++) bbox_pred[i * nd + k] = k < 6 ? expf(reg[i * nd + k] * sc) : reg[i * nd + k];
    points[i * 3] = (float)q.y * vs_tab[c * 3]; points[i * 3 + 1] = (float)q.z * vs_tab[c * 3 + 1];
    points[i * 3 + 2] = (float)q.w * vs_tab[c * 3 + 2];
    if (boost != 0.f && cls) cls[i * nc + c] = cls[i * nc + c] + boost;
}
__global__ __launch_bounds__(256) void k_head_outputs_bwd(const float *__restrict__ dbbox, const float *__restrict__ bbox_pred,
                                                          const float *__restrict__ reg, int nd, const int32_t *__restrict__ coords,
                                                          int64_t n, int nbatch, const float *__restrict__ scale, int nc,
                                                          float *__restrict__ dreg, float *__restrict__ dscale) {
    extern __shared__ float s_ds[];
    for (int k = threadIdx.x; k < nc; k += 256) s_ds[k] = 0.f;
    __syncthreads();
    for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int c = coords[i * 4] / nbatch;
        const float sc = scale[c];
        float acc = 0.f;
        for (int k = 0; k < nd; k++) {
            const float g = dbbox[i * nd + k];
            if (k < 6) {
                const float gb = g * bbox_pred[i * nd + k];
                dreg[i * nd + k] = gb * sc;
                acc += gb * reg[i * nd + k];
            } else {
                dreg[i * nd + k] = g;
            }
        }
        unsafeAtomicAdd(&s_ds[c], acc);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nc; k += 256)
        if (s_ds[k] != 0.f) unsafeAtomicAdd(&dscale[k], s_ds[k]);
}
extern "C" int cg3d_head_outputs_fwd(const float *reg, int32_t nd, const int32_t *coords, int64_t n, int32_t nbatch,
                                     const float *scale, const float *vs_tab, int32_t nc, float boost, float *cls, float *bbox_pred,
                                     float *points, cg3d_stream_t stream) {
    if (n < 0 || nd < 6 || nbatch <= 0 || nc <= 0) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    if (!reg || !coords || !scale || !vs_tab || !bbox_pred || !points || ((uintptr_t)coords & 15)) return CG3D_ERR_ARG;
    hipLaunchKernelGGL(k_head_outputs_fwd, dim3((unsigned)cg3d_divup(n, 256)), dim3(256), 0, cg3d_hs(stream), reg, nd, coords, n, nbatch,
                       scale, vs_tab, nc, boost, cls, bbox_pred, points);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_head_outputs_bwd(const float *dbbox, const float *bbox_pred, const float *reg, int32_t nd, const int32_t *coords,
                                     int64_t n, int32_t nbatch, const float *scale, int32_t nc, float *dreg, float *dscale,
                                     cg3d_stream_t stream) {
    if (n < 0 || nd < 6 || nbatch <= 0 || nc <= 0 || nc > 4096 || !dscale) return CG3D_ERR_ARG;
    if (hipMemsetAsync(dscale, 0, (size_t)nc * 4, cg3d_hs(stream)) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (n == 0) return CG3D_OK;
    if (!dbbox || !bbox_pred || !reg || !coords || !scale || !dreg) return CG3D_ERR_ARG;
    const unsigned g = (unsigned)(cg3d_divup(n, 256) < 256 ? cg3d_divup(n, 256) : 256);
    hipLaunchKernelGGL(k_head_outputs_bwd, dim3(g), dim3(256), (size_t)nc * 4, cg3d_hs(stream), dbbox, bbox_pred, reg, nd, coords, n,
                       nbatch, scale, nc, dreg, dscale);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
