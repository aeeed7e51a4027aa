// knn.hip -- brute-force k nearest neighbours for gfx950.
//
// Replaces pcdet/ops/knn/src/knn_cuda.cu:58-115 (knn_kernel + launcher).  The reference streams
// all n reference points from global memory in every thread; here a workgroup stages tiles of
// 1024 points in LDS (12 KB, conflict-free broadcast reads) and each thread owns one query.
// k == 1 (the only value on the CAGroup3D path, cagroup_head.py:479-481) keeps the running best
// in two registers; general k restates the reference's max-heap exactly so neighbour order and
// ties (strict `<`: the lowest index wins) are bit-identical to the oracle.
// Built with -ffp-contract=off so d2 = (dx*dx + dy*dy) + dz*dz rounds as on the host.
#include "cg3d_common.h"

#define KNN_TILE 1024

__device__ static inline void d_reheap(float *d, int *ix, int k) {
    int root = 0, child = 1;
    while (child < k) {
        if (child + 1 < k && d[child + 1] > d[child]) child++;
        if (d[root] > d[child]) return;
        float td = d[root]; d[root] = d[child]; d[child] = td;
        int ti = ix[root]; ix[root] = ix[child]; ix[child] = ti;
        root = child; child = root * 2 + 1;
    }
}

template <bool K1>
__global__ __launch_bounds__(256) void k_knn(int32_t n, int32_t m, int32_t k, const float *__restrict__ xyz,
                                             const float *__restrict__ new_xyz, int32_t *__restrict__ idx,
                                             float *__restrict__ dist2) {
    __shared__ float tile[KNN_TILE * 3];
    const int bi = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const float *X = xyz + (int64_t)bi * n * 3;
    const bool active = q < m;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (active) {
        const float *p = new_xyz + ((int64_t)bi * m + q) * 3;
        px = p[0]; py = p[1]; pz = p[2];
    }
    float bd[K1 ? 1 : 100];
    int bx[K1 ? 1 : 100];
    for (int i = 0; i < (K1 ? 1 : k); i++) { bd[i] = 1e10f; bx[i] = 0; }

    for (int base = 0; base < n; base += KNN_TILE) {
        const int cntp = (n - base < KNN_TILE) ? n - base : KNN_TILE;
        __syncthreads();
        for (int i = threadIdx.x; i < cntp * 3; i += 256) tile[i] = X[(int64_t)base * 3 + i];
        __syncthreads();
        if (active) {
            for (int i = 0; i < cntp; i++) {
                float x = tile[i * 3], y = tile[i * 3 + 1], z = tile[i * 3 + 2];
                float d2 = (px - x) * (px - x) + (py - y) * (py - y) + (pz - z) * (pz - z);
                if (d2 < bd[0]) {
                    bd[0] = d2; bx[0] = base + i;
                    if (!K1) d_reheap(bd, bx, k);
                }
            }
        }
    }
    if (!active) return;
    if (!K1) {
        for (int i = k - 1; i > 0; i--) {
            float td = bd[0]; bd[0] = bd[i]; bd[i] = td;
            int ti = bx[0]; bx[0] = bx[i]; bx[i] = ti;
            d_reheap(bd, bx, i);
        }
    }
    const int64_t o = ((int64_t)bi * m + q) * k;
    for (int i = 0; i < (K1 ? 1 : k); i++) { idx[o + i] = bx[i]; dist2[o + i] = bd[i]; }
}

extern "C" int cg3d_knn(int32_t b, int32_t n, int32_t m, int32_t k, const float *xyz, const float *new_xyz,
                        int32_t *idx, float *dist2, cg3d_stream_t stream) {
    if (k < 1 || k > 100 || b < 0 || n < 0 || m < 0 || b > 65535) return CG3D_ERR_ARG;
    if (b == 0 || m == 0) return CG3D_OK;
    dim3 g((unsigned)cg3d_divup(m, 256), (unsigned)b);
    if (k == 1) hipLaunchKernelGGL(k_knn<true>, g, dim3(256), 0, cg3d_hs(stream), n, m, k, xyz, new_xyz, idx, dist2);
    else hipLaunchKernelGGL(k_knn<false>, g, dim3(256), 0, cg3d_hs(stream), n, m, k, xyz, new_xyz, idx, dist2);
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
