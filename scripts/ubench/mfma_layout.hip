// Probe of the lane layout of v_mfma_f64_4x4x4_4b_f64 (gfx950): one-hot A lane p x one-hot B lane q
// -> which D lane is hit.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int *hit) {
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p)
        for (int q = 0; q < 64; ++q) {
            const double a = lane == p ? 1.0 : 0.0, b = lane == q ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) hit[p * 64 + q] = lane;
        }
}
int main() {
    int *dh; (void)hipMalloc(&dh, 4096 * 4); (void)hipMemset(dh, 0xff, 4096 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dh);
    int h[4096]; (void)hipMemcpy(h, dh, sizeof(h), hipMemcpyDeviceToHost);
    for (int p = 0; p < 64; ++p) {
        printf("A lane %2d pairs with B lanes:", p);
        for (int q = 0; q < 64; ++q) if (h[p * 64 + q] >= 0) printf(" %d->D%d", q, h[p * 64 + q]);
        printf("\n");
    }
    return 0;
}
