// Does v_mfma_f64_4x4x4_4b_f64 honour CBSZ / ABID (broadcast of one block's A operand to all four
// blocks) on gfx950?  A[lane] = 100 + lane, B one-hot at lane p: the D lanes of p's block then show
// WHICH A element was used.  Expected with cbsz = 2, abid = x: A element of block x.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CBSZ, int ABID>
__global__ void probe(double *out) {
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        const double a = 100.0 + lane, b = lane == p ? 1.0 : 0.0;
        out[p * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    }
}
template <int CBSZ, int ABID>
void run() {
    double *dh; (void)hipMalloc(&dh, 4096 * 8);
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dh);
    static double h[4096]; (void)hipMemcpy(h, dh, sizeof(h), hipMemcpyDeviceToHost);
    printf("cbsz %d abid %d:", CBSZ, ABID);
    for (int p : {0, 5, 22, 63}) {          // B one-hot lanes: (k, q, j) = (p >> 4, (p >> 2) & 3, p & 3)
        printf("  p=%d ->", p);
        for (int l = 0; l < 64; ++l) if (h[p * 64 + l] != 0.0) printf(" D%d=%.0f", l, h[p * 64 + l]);
    }
    printf("\n");
    (void)hipFree(dh);
}
int main() {
    run<0, 0>(); run<2, 0>(); run<2, 1>(); run<2, 3>(); run<1, 1>();
    return 0;
}
