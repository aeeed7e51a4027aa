// sort_vertices.hip -- anticlockwise ordering of intersection-polygon vertices for gfx950.
//
// Replaces pcdet/ops/rotated_iou/cuda_op/sort_vert_kernel.cu:15-139.  The reference launches ONE
// block per batch element (and the batch is always 1, pcdet/utils/iou3d_loss.py:15), i.e. a single
// workgroup for all box pairs; here every box pair gets its own thread across as many
// workgroups as needed, the 24 candidate vertices are read once into registers.
// Built with -ffp-contract=off: the comparisons round exactly as in the CPU oracle.
#include "cg3d_common.h"

#define SV_MAXV 32
#define SV_EPS 1e-8  // double literal, as in the reference (:8): comparisons promote to double

__device__ static inline bool d_cmp_vert(float x1, float y1, float x2, float y2) {
    if ((double)fabsf(x1 - x2) < SV_EPS && (double)fabsf(y2 - y1) < SV_EPS) return false;
    if (y1 > 0 && y2 < 0) return true;
    if (y1 < 0 && y2 > 0) return false;
    float n1 = (float)((double)(x1 * x1 + y1 * y1) + SV_EPS);
    float n2 = (float)((double)(x2 * x2 + y2 * y2) + SV_EPS);
    if (y1 > 0 && y2 > 0) return (double)(fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2) > SV_EPS;
    if (y1 < 0 && y2 < 0) return (double)(fabsf(x1) * x1 / n1 - fabsf(x2) * x2 / n2) < SV_EPS;
    return false;
}

__global__ __launch_bounds__(256) void k_sort_vertices(int64_t total, int32_t m, const float *__restrict__ vertices,
                                                       const uint8_t *__restrict__ mask,
                                                       const int32_t *__restrict__ num_valid,
                                                       int32_t *__restrict__ idx) {
    const int64_t p = blockIdx.x * (int64_t)256 + threadIdx.x;
    if (p >= total) return;
    const float *v = vertices + p * m * 2;
    const uint8_t *mk = mask + p * m;
    float vx[SV_MAXV], vy[SV_MAXV];
    unsigned valid = 0u;
    for (int j = 0; j < m; j++) {
        vx[j] = v[j * 2]; vy[j] = v[j * 2 + 1];
        if (mk[j]) valid |= 1u << j;
    }
    const int nv = num_valid[p];
    int pad = 0;
    for (int j = 8; j < m; ++j) if (!((valid >> j) & 1u)) { pad = j; break; }
    int out[9];
    if (nv < 3) {
        for (int j = 0; j < 9; ++j) out[j] = pad;
    } else {
        for (int j = 0; j < 9; ++j) out[j] = 0;
        for (int j = 0; j < nv && j < 9; ++j) {
            float x_min = 1.f, y_min = (float)(-SV_EPS);
            int i_take = 0;
            float x2 = 0.f, y2 = 0.f;
            if (j > 0) { int i2 = out[j - 1]; x2 = vx[i2]; y2 = vy[i2]; }
            for (int k = 0; k < m; ++k) {
                if (!((valid >> k) & 1u)) continue;
                float x = vx[k], y = vy[k];
                bool take = d_cmp_vert(x, y, x_min, y_min);
                if (j > 0) take = take && d_cmp_vert(x2, y2, x, y);
                if (take) { x_min = x; y_min = y; i_take = k; }
            }
            out[j] = i_take;
        }
        if (nv < 9) out[nv] = out[0];
        for (int j = nv + 1; j < 9; ++j) out[j] = pad;
        if (nv == 8) {
            int counter = 0;
            for (int j = 0; j < 4; ++j) {
                int check = out[j];
                for (int k = 4; k < 8; ++k) if (out[k] == check) counter++;
            }
            if (counter == 4) { out[4] = out[0]; for (int j = 5; j < 9; ++j) out[j] = pad; }
        }
    }
    for (int j = 0; j < 9; ++j) idx[p * 9 + j] = out[j];
}

extern "C" int cg3d_sort_vertices(int32_t b, int32_t n, int32_t m, const float *vertices, const uint8_t *mask,
                                  const int32_t *num_valid, int32_t *idx, cg3d_stream_t stream) {
    if (b < 0 || n < 0 || m < 9 || m > SV_MAXV) return CG3D_ERR_ARG;
    const int64_t total = (int64_t)b * n;
    if (total == 0) return CG3D_OK;
    hipLaunchKernelGGL(k_sort_vertices, dim3((unsigned)cg3d_divup(total, 256)), dim3(256), 0, cg3d_hs(stream), total, m,
                       vertices, mask, num_valid, idx);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
