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

// k == 1 fast path (the only value on the CAGroup3D path).  A ScanNet scene has ~43 k queries = 168
// workgroups of 256 -- fewer than the 256 CUs -- so the REFERENCE points are split across workgroups
// too (grid.y) and the partial winners are merged with one 64-bit atomicMin per query on the packed
// key (bits(d2) << 32 | index): d2 >= 0, so unsigned order == float order, and on equal distances the
// lower index wins -- exactly the strict `<` of the sequential scan (knn_cuda.cu:83).
// Reference points are staged as float4: one ds_read_b128 broadcast per point.
__global__ __launch_bounds__(256) void k_knn1(int32_t n, int32_t m, int32_t splits, const float *__restrict__ xyz,
                                              const float *__restrict__ new_xyz, unsigned long long *__restrict__ best) {
    __shared__ float4 tile[KNN_TILE];
    const int bi = blockIdx.z;
    const float *X = xyz + (int64_t)bi * n * 3;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int qq = q < m ? q : (m - 1);
    const float *p = new_xyz + ((int64_t)bi * m + qq) * 3;
    const float px = p[0], py = p[1], pz = p[2];
    float bd = 1e10f;
    int bx = 0;
    const int per = (n + splits - 1) / splits;
    const int lo = blockIdx.y * per, hi = (lo + per < n) ? lo + per : n;
    for (int base = lo; base < hi; base += KNN_TILE) {
        const int cntp = (hi - base < KNN_TILE) ? hi - base : KNN_TILE;
        __syncthreads();
        for (int i = threadIdx.x; i < cntp; i += 256) {
            const float *s = X + (int64_t)(base + i) * 3;
            tile[i] = make_float4(s[0], s[1], s[2], 0.f);
        }
        __syncthreads();
        for (int i = 0; i < cntp; i++) {
            const float4 t = tile[i];
            const float d2 = (px - t.x) * (px - t.x) + (py - t.y) * (py - t.y) + (pz - t.z) * (pz - t.z);
            if (d2 < bd) { bd = d2; bx = base + i; }
        }
    }
    if (q < m) {
        const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bx;
        atomicMin(&best[(int64_t)bi * m + q], key);
    }
}
__global__ void k_knn1_init(unsigned long long *best, int64_t total) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < total) best[t] = ((unsigned long long)__float_as_uint(1e10f) << 32);
}
__global__ void k_knn1_unpack(const unsigned long long *__restrict__ best, int64_t total, int32_t *__restrict__ idx,
                              float *__restrict__ dist2) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    const unsigned long long key = best[t];
    idx[t] = (int32_t)(key & 0xffffffffULL);
    dist2[t] = __uint_as_float((unsigned)(key >> 32));
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

extern "C" int64_t cg3d_knn_ws_bytes(int32_t b, int32_t m, int32_t k) { return k == 1 ? (int64_t)b * m * 8 : 0; }

extern "C" int cg3d_knn(int32_t b, int32_t n, int32_t m, int32_t k, const float *xyz, const float *new_xyz,
                        int32_t *idx, float *dist2, void *ws, cg3d_stream_t stream) {
    if (k < 1 || k > 100 || b < 0 || n < 0 || m < 0 || b > 65535) return CG3D_ERR_ARG;
    if (b == 0 || m == 0) return CG3D_OK;
    hipStream_t s = cg3d_hs(stream);
    dim3 g((unsigned)cg3d_divup(m, 256), (unsigned)b);
    if (k == 1 && n > 0 && ws != nullptr) {
        const int64_t total = (int64_t)b * m;
        const int64_t qblocks = cg3d_divup(m, 256);
        int64_t splits = cg3d_divup(2048, qblocks * b);          // >= 2048 workgroups in flight
        const int64_t max_splits = cg3d_divup(n, 2 * KNN_TILE);   // but at least two LDS tiles of work each
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        if (splits > 65535) splits = 65535;
        unsigned long long *best = (unsigned long long *)ws;
        hipLaunchKernelGGL(k_knn1_init, dim3((unsigned)cg3d_divup(total, 256)), dim3(256), 0, s, best, total);
        hipLaunchKernelGGL(k_knn1, dim3((unsigned)qblocks, (unsigned)splits, (unsigned)b), dim3(256), 0, s, n, m,
                           (int32_t)splits, xyz, new_xyz, best);
        hipLaunchKernelGGL(k_knn1_unpack, dim3((unsigned)cg3d_divup(total, 256)), dim3(256), 0, s, best, total, idx, dist2);
    } else if (k == 1) {
        hipLaunchKernelGGL(k_knn<true>, g, dim3(256), 0, s, n, m, k, xyz, new_xyz, idx, dist2);
    } else {
        hipLaunchKernelGGL(k_knn<false>, g, dim3(256), 0, s, n, m, k, xyz, new_xyz, idx, dist2);
    }
    CG3D_CHECK_LAUNCH();
    return CG3D_OK;
}
