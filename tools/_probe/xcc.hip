#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x + blockIdx.y * gridDim.x] = (int)(x & 0xf);
}
int main() {
    int *d; hipMalloc(&d, 4096 * 4);
    int h[4096];
    for (int cfg = 0; cfg < 3; cfg++) {
        dim3 g = cfg == 0 ? dim3(64) : cfg == 1 ? dim3(497) : dim3(61, 2);
        int n = g.x * g.y;
        hipLaunchKernelGGL(k, g, dim3(256), cfg == 1 ? 60000 : 0, 0, d);
        hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        printf("grid %d x %d:", g.x, g.y);
        int bad = 0;
        for (int i = 0; i < n; i++) { if (i < 40) printf(" %d", h[i]); if (h[i] != h[i % 8]) bad++; }
        printf("  ... mismatches vs i%%8 pattern: %d of %d\n", bad, n);
    }
    return 0;
}
