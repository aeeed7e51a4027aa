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
