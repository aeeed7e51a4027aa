// optim.hip -- the optimiser step of the training loop for gfx950: gradient clipping + AdamW for EVERY parameter of the
// model in one launch.
//
// Replaces, as one HBM pass, what `tools/train_utils/train_utils.py:40-47` does per iteration with
// `clip_grad_norm_(model.parameters(), GRAD_NORM_CLIP)` and `optimizer.step()` (AdamW, `optimization/__init__.py:24-26`):
// the multiply of every gradient by the clip coefficient (one multi-tensor pass over the gradients of its own) and the
// AdamW update (a second pass).  The gradient norm itself stays with the caller (a reduction that must finish before
// the first update); its result enters as a DEVICE scalar, so nothing is read back to the host.
//   table  int64 [nrows][5] = { param address, exp_avg address, exp_avg_sq address, first element, element count }, a row
//          per <= 32 K-element chunk of a parameter (all fp32, contiguous); row -> parameter id in `pid` (int32 [nrows]);
//   grads  int64 [nparams]: this step's gradient addresses (autograd hands out new gradient tensors every step -- the
//          only per-step upload);
//   clip   device float* or NULL: every gradient is multiplied by *clip before use (and is NOT written back).
// Arithmetic, per element, in fp32, in the order of torch's fused AdamW kernel:
//   p -= lr*wd*p;  m += (g - m)*(1 - beta1);  v = beta2*v + (1 - beta2)*g*g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - beta1^t, bc2 = 1 - beta2^t computed by the caller (t = step count after this step).
// HBM-bound: 16 B read + 12 B written per parameter.
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include "cg3d_common.h"

__global__ __launch_bounds__(256) void k_adamw_table(const int64_t *__restrict__ table, const int32_t *__restrict__ pid,
                                                     const int64_t *__restrict__ grads, const float *__restrict__ clip,
                                                     float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                     float bc2_sqrt) {
    const int64_t *row = table + (int64_t)blockIdx.x * 5;
    float *p = reinterpret_cast<float *>(row[0]) + row[3];
    float *m = reinterpret_cast<float *>(row[1]) + row[3];
    float *v = reinterpret_cast<float *>(row[2]) + row[3];
    const float *g = reinterpret_cast<const float *>(grads[pid[blockIdx.x]]) + row[3];
    const int n = (int)row[4];
    const float cs = clip ? *clip : 1.f;
    const float step_size = lr / bc1, omb1 = 1.f - beta1, omb2 = 1.f - beta2, lrwd = lr * wd;
    auto upd = [&](float &pp, float &mm, float &vv, float gg) {
        gg *= cs;
        pp -= lrwd * pp;
        mm += (gg - mm) * omb1;
        vv = beta2 * vv + omb2 * gg * gg;
        pp -= step_size * mm / (sqrtf(vv) / bc2_sqrt + eps);
    };
    const bool vec = !(((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15);
    if (vec) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            float4 pp = reinterpret_cast<float4 *>(p)[i], mm = reinterpret_cast<float4 *>(m)[i], vv = reinterpret_cast<float4 *>(v)[i];
            const float4 gg = reinterpret_cast<const float4 *>(g)[i];
            upd(pp.x, mm.x, vv.x, gg.x); upd(pp.y, mm.y, vv.y, gg.y); upd(pp.z, mm.z, vv.z, gg.z); upd(pp.w, mm.w, vv.w, gg.w);
            reinterpret_cast<float4 *>(p)[i] = pp; reinterpret_cast<float4 *>(m)[i] = mm; reinterpret_cast<float4 *>(v)[i] = vv;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) upd(p[i], m[i], v[i], g[i]);
    } else {
        for (int i = threadIdx.x; i < n; i += 256) upd(p[i], m[i], v[i], g[i]);
    }
}
extern "C" int cg3d_adamw_step(const int64_t *table, const int32_t *pid, int64_t nrows, const int64_t *grads, const float *clip,
                               float lr, float beta1, float beta2, float eps, float weight_decay, float bias_correction1,
                               float bias_correction2, cg3d_stream_t stream) {
    if (nrows < 0 || nrows > 0x7fffffffll || (nrows > 0 && (!table || !pid || !grads))) return CG3D_ERR_ARG;
    if (!(bias_correction1 > 0.f) || !(bias_correction2 > 0.f)) return CG3D_ERR_ARG;
    if (nrows == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_adamw_table, dim3((unsigned)nrows), dim3(256), 0, cg3d_hs(stream), table, pid, grads, clip, lr, beta1, beta2,
                       eps, weight_decay, bias_correction1, sqrtf(bias_correction2));
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}

// ---- gradient norm + clip coefficient (clip_grad_norm_, reference tools/train_utils/train_utils.py:40-47) over the SAME chunk
// table: one pass over every gradient (a chunk's sum of squares in double inside the workgroup, one fp64 atomic per chunk),
// then one thread turns the total into  norm = sqrt(sum)  and  coef = min(max_norm / (norm + 1e-6), 1)  -- torch's
// _foreach_norm + stack + vector_norm + clamp chain was 12 launches and three 432-element Python lists per step.
__global__ __launch_bounds__(256) void k_grad_sumsq(const int64_t *__restrict__ table, const int32_t *__restrict__ pid,
                                                    const int64_t *__restrict__ grads, double *__restrict__ sum) {
    __shared__ double red[4];
    const int64_t *row = table + (int64_t)blockIdx.x * 5;
    const float *g = reinterpret_cast<const float *>(grads[pid[blockIdx.x]]) + row[3];
    const int n = (int)row[4];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;              // <= 32 K elements per chunk, 128 per thread: fp32 partials
    if (!((uintptr_t)g & 15)) {
        const int n4 = n >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) {
            const float4 v = reinterpret_cast<const float4 *>(g)[i];
            s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w * v.w;
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) s0 += g[i] * g[i];
    } else {
        for (int i = threadIdx.x; i < n; i += 256) s0 += g[i] * g[i];
    }
    double s = (double)s0 + (double)s1 + (double)s2 + (double)s3;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, red[0] + red[1] + red[2] + red[3]);
}
__global__ void k_clip_coef(const double *__restrict__ sum, float max_norm, float *__restrict__ norm_out, float *__restrict__ coef_out) {
    const float norm = (float)sqrt(*sum);
    *norm_out = norm;
    const float c = max_norm / (norm + 1e-6f);
    // torch.clamp(max_norm / (norm + 1e-6), max=1) propagates NaN: a non-finite gradient norm poisons EVERY parameter, as
    // the reference's clip_grad_norm_ does (tools/train_utils/train_utils.py:40-47), instead of corrupting only some tensors
    *coef_out = (c < 1.f || c != c) ? c : 1.f;
}
extern "C" int cg3d_grad_norm_clip(const int64_t *table, const int32_t *pid, int64_t nrows, const int64_t *grads, float max_norm,
                                   double *scratch, float *norm, float *coef, cg3d_stream_t stream) {
    if (nrows < 0 || nrows > 0x7fffffffll || (nrows > 0 && (!table || !pid || !grads)) || !scratch || !norm || !coef) return CG3D_ERR_ARG;
    hipStream_t s = cg3d_hs(stream);
    if (hipMemsetAsync(scratch, 0, sizeof(double), s) != hipSuccess) return CG3D_ERR_LAUNCH;
    if (nrows > 0) hipLaunchKernelGGL(k_grad_sumsq, dim3((unsigned)nrows), dim3(256), 0, s, table, pid, grads, scratch);
    hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(1), 0, s, scratch, max_norm, norm, coef);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
