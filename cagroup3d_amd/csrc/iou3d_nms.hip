// iou3d_nms.hip -- rotated / axis-aligned BEV overlap, IoU and NMS for gfx950.
//
// Replaces pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu (boxes_overlap_kernel :236-249,
// boxes_iou_bev_kernel :251-265, nms_kernel :267-311, nms_normal_kernel :328-372) and the HOST
// greedy scan of iou3d_nms.cpp:117-132 / :167-182, which here runs on the device.
//
// MI355X mapping: one wavefront IS the 64-box suppression tile -- lane j tests (row box, column
// box j) and the 64-bit mask word is a single wave ballot.  Only tiles on/above the diagonal are
// produced (the greedy scan never reads the others).  The scan processes the boxes 64 at a time:
// the in-tile dependency chain runs on the diagonal words held in registers (readlane), then the
// rows of the kept boxes are OR-ed into the running removal words with coalesced 8-byte loads.
// Built with -ffp-contract=off: IoU values are bit-identical to the CPU oracle.
#include "cg3d_common.h"

#include "dg_geom.h"

// ---------------------------------------------------------------- pairwise overlap / IoU
template <bool IOU>
__global__ __launch_bounds__(256) void k_pairwise(const float *__restrict__ A, int64_t na, const float *__restrict__ B,
                                                  int64_t nb, float *__restrict__ out) {
    // 16 x 16 pairs per workgroup; column boxes staged in LDS, consecutive lanes -> consecutive b
    __shared__ float sb[16 * 7], sa[16 * 7];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t a0 = (int64_t)blockIdx.y * 16, b0 = (int64_t)blockIdx.x * 16;
    if (threadIdx.x < 112) {
        int64_t bi = b0 + threadIdx.x / 7;
        sb[threadIdx.x] = bi < nb ? B[b0 * 7 + threadIdx.x] : 0.f;
    } else if (threadIdx.x >= 128 && threadIdx.x < 240) {
        int t = threadIdx.x - 128;
        int64_t ai = a0 + t / 7;
        sa[t] = ai < na ? A[a0 * 7 + t] : 0.f;
    }
    __syncthreads();
    const int64_t ai = a0 + ty, bi = b0 + tx;
    if (ai >= na || bi >= nb) return;
    float ba[7], bb[7];
#pragma unroll
    for (int i = 0; i < 7; i++) { ba[i] = sa[ty * 7 + i]; bb[i] = sb[tx * 7 + i]; }
    out[ai * nb + bi] = IOU ? d_iou_bev(ba, bb) : d_box_overlap(ba, bb);
}
extern "C" int cg3d_boxes_overlap_bev(const float *A, int64_t na, const float *B, int64_t nb, float *out,
                                      cg3d_stream_t stream) {
    if (na < 0 || nb < 0) return CG3D_ERR_ARG;
    if (na == 0 || nb == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_pairwise<false>, dim3((unsigned)cg3d_divup(nb, 16), (unsigned)cg3d_divup(na, 16)), dim3(256), 0,
                       cg3d_hs(stream), A, na, B, nb, out);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
// boxes_iou_bev_cpu (pcdet/ops/iou3d_nms/src/iou3d_nms_api.cpp:16, iou3d_cpu.cpp:232-252): the same pairwise BEV IoU on HOST
// pointers, computed by the host instantiation of the very functions the kernels run (bit-identical results; no device,
// no stream).  It is the reference's own CPU entry point of this module, not a fallback of the device path.
extern "C" int cg3d_boxes_iou_bev_cpu(const float *A, int64_t na, const float *B, int64_t nb, float *out) {
    if (na < 0 || nb < 0) return CG3D_ERR_ARG;
    for (int64_t i = 0; i < na; i++)
        for (int64_t j = 0; j < nb; j++) out[i * nb + j] = d_iou_bev(A + i * 7, B + j * 7);
    return CG3D_OK;
}
extern "C" int cg3d_boxes_iou_bev(const float *A, int64_t na, const float *B, int64_t nb, float *out,
                                  cg3d_stream_t stream) {
    if (na < 0 || nb < 0) return CG3D_ERR_ARG;
    if (na == 0 || nb == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_pairwise<true>, dim3((unsigned)cg3d_divup(nb, 16), (unsigned)cg3d_divup(na, 16)), dim3(256), 0,
                       cg3d_hs(stream), A, na, B, nb, out);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ---------------------------------------------------------------- NMS
// grid: x = column tile, y = row tile, z = segment.  One wave per (row tile, column tile): the 64
// row boxes are walked sequentially, lane j tests column box j, the mask word is the ballot.
template <bool ROTATED>
__global__ __launch_bounds__(64) void k_nms_mask(const float *__restrict__ boxes_all, const int64_t *__restrict__ seg_off,
                                                 const int64_t *__restrict__ mask_off, int64_t n_single, float thr,
                                                 unsigned long long *__restrict__ mask_all) {
    const int seg = blockIdx.z;
    const int64_t o = seg_off ? seg_off[seg] : 0;
    const int64_t n = seg_off ? seg_off[seg + 1] - o : n_single;
    const float *boxes = boxes_all + o * 7;
    unsigned long long *mask = mask_all + (mask_off ? mask_off[seg] : 0);
    const int64_t cb = (n + 63) / 64;
    const int64_t rt = blockIdx.y, ct = blockIdx.x;
    if (rt >= cb || ct >= cb || ct < rt) return;
    __shared__ float rows[64 * 7];
    const int lane = threadIdx.x;
    const int64_t rsz = (n - rt * 64 < 64) ? n - rt * 64 : 64;
    const int64_t csz = (n - ct * 64 < 64) ? n - ct * 64 : 64;
    for (int i = lane; i < rsz * 7; i += 64) rows[i] = boxes[rt * 64 * 7 + i];
    float cbx[7];
#pragma unroll
    for (int i = 0; i < 7; i++) cbx[i] = (lane < csz) ? boxes[(ct * 64 + lane) * 7 + i] : 0.f;
    __syncthreads();
    for (int i = 0; i < rsz; i++) {
        float rb[7];
#pragma unroll
        for (int q = 0; q < 7; q++) rb[q] = rows[i * 7 + q];
        bool hit = false;
        const int start = (rt == ct) ? i + 1 : 0;
        if (lane >= start && lane < csz) {
            float iou = ROTATED ? d_iou_bev(rb, cbx) : d_iou_normal(rb, cbx);
            hit = iou > thr;
        }
        unsigned long long word = __ballot(hit);
        if (lane == 0) mask[(rt * 64 + i) * cb + ct] = word;
    }
}

__global__ __launch_bounds__(256) void k_nms_scan(const int64_t *__restrict__ seg_off, const int64_t *__restrict__ mask_off,
                                                  int64_t n_single, const unsigned long long *__restrict__ mask_all,
                                                  int64_t *__restrict__ keep_all, int32_t *__restrict__ num_keep) {
    // Greedy scan, 64 boxes per step.  Wave 0 resolves the in-tile chain on the diagonal words
    // (registers + readlane); then all 4 waves OR the mask rows of the kept boxes into the later
    // removal words: thread (q, jj) folds the kept boxes b == q (mod 4) for column word j -- 16
    // independent 8-byte loads in flight per thread, one LDS atomic per (thread, word).
    extern __shared__ unsigned long long remv[];  // cb words + 1 (kept of the current tile)
    const int seg = blockIdx.x;
    const int64_t o = seg_off ? seg_off[seg] : 0;
    const int64_t n = seg_off ? seg_off[seg + 1] - o : n_single;
    const unsigned long long *mask = mask_all + (mask_off ? mask_off[seg] : 0);
    int64_t *keep = keep_all + o;
    const int64_t cb = (n + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    unsigned long long *kept_sh = remv + cb;
    for (int64_t j = tid; j < cb; j += 256) remv[j] = 0ULL;
    __syncthreads();
    int32_t nk = 0;
    for (int64_t blk = 0; blk < cb; blk++) {
        const int64_t bsz = (n - blk * 64 < 64) ? n - blk * 64 : 64;
        if (q == 0) {
            unsigned long long diag = (lane < bsz) ? mask[(blk * 64 + lane) * cb + blk] : 0ULL;
            unsigned long long cur = remv[blk];
            unsigned long long kept = 0ULL;
            for (int b = 0; b < bsz; b++) {
                unsigned long long d = __shfl(diag, b);
                if (!((cur >> b) & 1ULL)) { kept |= 1ULL << b; cur |= d; }
            }
            if ((kept >> lane) & 1ULL) {
                int pos = __popcll(kept & ((1ULL << lane) - 1ULL));
                keep[nk + pos] = blk * 64 + lane;
            }
            nk += __popcll(kept);
            if (lane == 0) *kept_sh = kept;
        }
        __syncthreads();
        const unsigned long long kept = *kept_sh;
        for (int64_t j = blk + 1 + lane; j < cb; j += 64) {
            unsigned long long acc = 0ULL;
#pragma unroll
            for (int b = 0; b < 16; b++) {
                const int bit = b * 4 + q;
                if ((kept >> bit) & 1ULL) acc |= mask[(blk * 64 + bit) * cb + j];
            }
            if (acc) atomicOr(&remv[j], acc);
        }
        __syncthreads();
    }
    if (tid == 0) num_keep[seg] = nk;
}

static int nms_launch(const float *boxes, const int64_t *seg_off, const int64_t *mask_off, int32_t nseg,
                      int64_t max_seg, float thr, int32_t rotated, uint64_t *mask_ws, int64_t *keep,
                      int32_t *num_keep, hipStream_t s) {
    if (max_seg < 0 || nseg < 0) return CG3D_ERR_ARG;
    if (nseg == 0) return CG3D_OK;
    if (hipMemsetAsync(num_keep, 0, sizeof(int32_t) * nseg, s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (max_seg == 0) return CG3D_OK;
    const int64_t cb = cg3d_divup(max_seg, 64);
    if (cb > 65535 || nseg > 65535) return CG3D_ERR_ARG;
    if ((cb + 1) * 8 > 64 * 1024) return CG3D_ERR_ARG;  // removal words live in LDS
    dim3 g((unsigned)cb, (unsigned)cb, (unsigned)nseg);
    if (rotated)
        hipLaunchKernelGGL(k_nms_mask<true>, g, dim3(64), 0, s, boxes, seg_off, mask_off, max_seg, thr,
                           (unsigned long long *)mask_ws);
    else
        hipLaunchKernelGGL(k_nms_mask<false>, g, dim3(64), 0, s, boxes, seg_off, mask_off, max_seg, thr,
                           (unsigned long long *)mask_ws);
    hipLaunchKernelGGL(k_nms_scan, dim3((unsigned)nseg), dim3(256), (size_t)(cb + 1) * 8, s, seg_off, mask_off, max_seg,
                       (const unsigned long long *)mask_ws, keep, num_keep);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? CG3D_OK : CG3D_ERR_LAUNCH;
}
extern "C" int cg3d_nms(const float *boxes, int64_t n, float thresh, int32_t rotated, uint64_t *mask_ws,
                        int64_t *keep, int32_t *num_keep, cg3d_stream_t stream) {
    if (n < 0) return CG3D_ERR_ARG;
    return nms_launch(boxes, nullptr, nullptr, 1, n, thresh, rotated, mask_ws, keep, num_keep, cg3d_hs(stream));
}
// nms_gpu / nms_normal_gpu in the reference's own shape (iou3d_nms.h:9-12): host keep list, count returned, host blocked.
extern "C" int64_t cg3d_nms_gpu_ws_bytes(int64_t n) {
    if (n < 0) n = 0;
    return n * ((n + 63) / 64) * 8 + n * 8 + 16;
}
static int nms_host_keep(const float *boxes, int64_t n, int64_t *keep_host, float thresh, int rotated, void *ws, hipStream_t s) {
    if (n < 0 || (n > 0 && (!boxes || !keep_host)) || !ws) return CG3D_ERR_ARG;
    if (n == 0) return 0;
    uint64_t *mask = (uint64_t *)ws;
    int64_t *keep = (int64_t *)(mask + n * ((n + 63) / 64));
    int32_t *num = (int32_t *)(keep + n);
    const int rc = nms_launch(boxes, nullptr, nullptr, 1, n, thresh, rotated, mask, keep, num, s);
    if (rc != CG3D_OK) return rc;
    int32_t nk = 0;
    if (hipMemcpyAsync(&nk, num, sizeof(int32_t), hipMemcpyDeviceToHost, s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (nk < 0 || nk > n) return CG3D_ERR_LAUNCH;
    if (nk > 0 && hipMemcpyAsync(keep_host, keep, (size_t)nk * sizeof(int64_t), hipMemcpyDeviceToHost, s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return CG3D_ERR_LAUNCH;
    return (int)nk;
}
extern "C" int cg3d_nms_gpu(const float *boxes, int64_t n, int64_t *keep_host, float thresh, void *ws, cg3d_stream_t stream) {
    return nms_host_keep(boxes, n, keep_host, thresh, 1, ws, cg3d_hs(stream));
}
extern "C" int cg3d_nms_normal_gpu(const float *boxes, int64_t n, int64_t *keep_host, float thresh, void *ws, cg3d_stream_t stream) {
    return nms_host_keep(boxes, n, keep_host, thresh, 0, ws, cg3d_hs(stream));
}
extern "C" int cg3d_nms_batched(const float *boxes, const int64_t *seg_off, const int64_t *mask_off, int32_t nseg,
                                int64_t max_seg, float thresh, int32_t rotated, uint64_t *mask_ws, int64_t *keep,
                                int32_t *num_keep, cg3d_stream_t stream) {
    return nms_launch(boxes, seg_off, mask_off, nseg, max_seg, thresh, rotated, mask_ws, keep, num_keep,
                      cg3d_hs(stream));
}


// ---------------------------------------------------------------- points strictly inside (rotated) boxes
// find_points_in_boxes (pcdet/models/dense_heads/target_assigner/cagroup3d_assigner.py:9-36): the point is moved into
// the box frame (shift, rotate by -heading about z, add the centre back), the six face distances are formed in the
// reference's operation order and the point is inside when the smallest is > 0.  One thread per (point, box); the
// reference materialises [n, g, 7] fp32 temporaries through ~40 tensor launches (5.5 ms per step at 155 k x 80).
// Optional segment ids restrict a point to the boxes of its own scene.
__global__ void k_points_in_boxes(const float *__restrict__ pts, int64_t n, const float *__restrict__ boxes, int32_t g,
                                  const int32_t *__restrict__ pseg, const int32_t *__restrict__ bseg,
                                  uint8_t *__restrict__ out) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * g) return;
    const int64_t i = t / g;
    const int j = (int)(t % g);
    const float *b = boxes + (int64_t)j * 7;
    const float px = pts[i * 3], py = pts[i * 3 + 1], pz = pts[i * 3 + 2];
    const float cx = b[0], cy = b[1], cz = b[2];
    const float sx = px - cx, sy = py - cy, sz = pz - cz;
    const float c = dg_cosf(-b[6]), s = dg_sinf(-b[6]);
    const float rx = sx * c + sy * s, ry = sy * c - sx * s;           // rows of [[c,-s,0],[s,c,0],[0,0,1]] applied to (sx,sy,sz)
    const float qx = cx + rx, qy = cy + ry, qz = cz + sz;
    const float hx = b[3] / 2, hy = b[4] / 2, hz = b[5] / 2;
    float m = qx - cx + hx;
    m = fminf(m, cx + hx - qx);
    m = fminf(m, qy - cy + hy);
    m = fminf(m, cy + hy - qy);
    m = fminf(m, qz - cz + hz);
    m = fminf(m, cz + hz - qz);
    bool in = m > 0.f;
    if (pseg && bseg) in = in && pseg[i] == bseg[j];
    out[t] = in ? 1 : 0;
}
extern "C" int cg3d_points_in_boxes(const float *points, int64_t n, const float *boxes, int32_t g, const int32_t *point_seg,
                                    const int32_t *box_seg, uint8_t *inside, cg3d_stream_t stream) {
    if (n < 0 || g < 0) return CG3D_ERR_ARG;
    if (n == 0 || g == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_points_in_boxes, dim3((unsigned)cg3d_divup(n * g, 256)), dim3(256), 0, cg3d_hs(stream), points, n,
                       boxes, g, point_seg, box_seg, inside);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ---------------------------------------------------------------- FCOS-style assignment of the class-map points
// CAGroup3DAssigner.assign (pcdet/models/dense_heads/target_assigner/cagroup3d_assigner.py:62-130; compute_centerness
// :39-46), all classes and scenes of the batch at once: a point of class map c competes for the GT boxes of class c of
// its own scene; it is positive for the smallest-volume such box it lies strictly inside AND among whose top-k most
// central points it is.  The reference (and the torch mirror) materialise [n, m, 7] face distances and five [n, m]
// companions through ~70 tensor launches; here
//   k_fcos_centerness  one thread per (point, box): the face distances in the reference's operation order (as
//                      k_points_in_boxes), centerness = sqrt(min/max * min/max * min/max) evaluated left to right like the
//                      tensor expression, -1 where the pair does not compete (outside / other class / other scene);
//   (the k-th largest centerness of every box column is taken by the caller: one top-k over the [n, m] table)
//   k_fcos_assign      one thread per point: smallest-volume box among those whose centerness beats the column's k-th
//                      value (ties -> the lower box index, as one min() over the row does), label, centerness and box targets.
__device__ static inline float fcos_centerness_of(const float *p, const float *b, bool *inside) {
    const float cx = b[0], cy = b[1], cz = b[2];
    const float sx = p[0] - cx, sy = p[1] - cy, sz = p[2] - cz;
    const float c = dg_cosf(-b[6]), s = dg_sinf(-b[6]);
    const float rx = sx * c + sy * s, ry = sy * c - sx * s;
    const float qx = cx + rx, qy = cy + ry, qz = cz + sz;
    const float hx = b[3] / 2, hy = b[4] / 2, hz = b[5] / 2;
    const float x0 = qx - cx + hx, x1 = cx + hx - qx, y0 = qy - cy + hy, y1 = cy + hy - qy, z0 = qz - cz + hz, z1 = cz + hz - qz;
    float m = x0;
    m = fminf(m, x1); m = fminf(m, y0); m = fminf(m, y1); m = fminf(m, z0); m = fminf(m, z1);
    *inside = m > 0.f;
    // x.min / x.max * y.min / y.max * z.min / z.max, left to right
    float v = fminf(x0, x1) / fmaxf(x0, x1);
    v = v * fminf(y0, y1);
    v = v / fmaxf(y0, y1);
    v = v * fminf(z0, z1);
    v = v / fmaxf(z0, z1);
    return sqrtf(v);
}
__global__ void k_fcos_centerness(const float *__restrict__ pts, const int64_t *__restrict__ pt_cls,
                                  const int64_t *__restrict__ pt_scene, int64_t n, const float *__restrict__ gt,
                                  const int64_t *__restrict__ gt_cls, const int64_t *__restrict__ gt_scene, int32_t m,
                                  float *__restrict__ cness) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= n * m) return;
    const int64_t i = t / m;
    const int j = (int)(t % m);
    bool inside;
    const float v = fcos_centerness_of(pts + i * 3, gt + (int64_t)j * 7, &inside);
    const bool compete = inside && pt_cls[i] == gt_cls[j] && (!pt_scene || pt_scene[i] == gt_scene[j]);
    cness[t] = compete ? v : -1.f;
}
__global__ void k_fcos_assign(const float *__restrict__ cness, const float *__restrict__ kth, int64_t n,
                              const float *__restrict__ gt, const int64_t *__restrict__ gt_cls, int32_t m,
                              float *__restrict__ ctr_t, float *__restrict__ box_t, int64_t *__restrict__ labels) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    float best = 1e8f;                       // FLOAT_MAX of the reference
    int bj = 0;
    for (int j = 0; j < m; j++) {
        const float c = cness[i * m + j];
        if (c > kth[j] && c >= 0.f) {        // competes (>= 0) and is among the top k of the column
            const float *b = gt + (int64_t)j * 7;
            const float vol = b[3] * b[4] * b[5];
            if (vol < best) { best = vol; bj = j; }
        }
    }
    const bool pos = best != 1e8f;
    labels[i] = pos ? gt_cls[bj] : -1;
    const float cb = cness[i * m + bj];
    ctr_t[i] = cb;                           // (rows with label -1: unspecified, never read -- as in assign_all_classes)
#pragma unroll
    for (int q = 0; q < 7; q++) box_t[i * 7 + q] = gt[(int64_t)bj * 7 + q];
}
extern "C" int cg3d_fcos_centerness(const float *points, const int64_t *pt_cls, const int64_t *pt_scene, int64_t n, const float *gt,
                                    const int64_t *gt_cls, const int64_t *gt_scene, int32_t m, float *cness,
                                    cg3d_stream_t stream) {
    if (n < 0 || m < 0 || (!pt_scene) != (!gt_scene)) return CG3D_ERR_ARG;
    if (n == 0 || m == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_fcos_centerness, dim3((unsigned)cg3d_divup(n * m, 256)), dim3(256), 0, cg3d_hs(stream), points, pt_cls,
                       pt_scene, n, gt, gt_cls, gt_scene, m, cness);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
extern "C" int cg3d_fcos_assign(const float *cness, const float *kth, int64_t n, const float *gt, const int64_t *gt_cls, int32_t m,
                                float *ctr_t, float *box_t, int64_t *labels, cg3d_stream_t stream) {
    if (n < 0 || m < 1) return CG3D_ERR_ARG;
    if (n == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_fcos_assign, dim3((unsigned)cg3d_divup(n, 256)), dim3(256), 0, cg3d_hs(stream), cness, kth, n, gt, gt_cls, m,
                       ctr_t, box_t, labels);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
