// What keeps v_mfma_f64_4x4x4_4b_f64 below its 72 TF loop rate inside the syrk kernel?
// 36 in-place accumulators per wave (as the kernel), 2 waves per SIMD, variants:
//   V0 operands fixed                      V1 operands cycle through 16 registers
//   V2 V1 + 16 ds_read_b64 per 36 MFMAs feeding the operands (software pipelined one step ahead)
//   V3 V2 + 8 v_mul_f64 per step            V4 V3 with the loads NOT pipelined (load -> wait -> use)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int V>
__global__ __launch_bounds__(256) void k(double *out, int iters, double s) {
    __shared__ double lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    double acc[36];
    for (int i = 0; i < 36; ++i) acc[i] = 0;
    double a[8], b[8], na[8], nb[8];
    for (int i = 0; i < 8; ++i) { a[i] = 1.0 + i; b[i] = 2.0 + i; na[i] = a[i]; nb[i] = b[i]; }
    const double *p = lds + lane;
    for (int it = 0; it < iters; ++it) {
        if (V >= 2 && V != 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { na[i] = p[((it + i) & 31) * 64]; nb[i] = p[((it + i + 8) & 31) * 64 + 2048 - 2048 * (i & 1)]; }
        }
        if (V == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = p[((it + i) & 31) * 64]; b[i] = p[((it + i + 8) & 31) * 64 + 2048 - 2048 * (i & 1)]; }
        }
        double xa[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xa[i] = V >= 3 ? a[i] * s : a[i];
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const int ia = V == 0 ? 0 : i % 8, ib = V == 0 ? 0 : (i * 3) % 8;
            acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(xa[ia], b[ib], acc[i], 0, 0, 0);
        }
        if (V >= 2 && V != 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = na[i]; b[i] = nb[i]; }
        }
    }
    double r = 0;
    for (int i = 0; i < 36; ++i) r += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int V>
void run(double *out) {
    const int iters = 4000, grid = 512;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, 10, 1.0000001);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    printf("V%d: %7.3f ms  %6.2f TFLOP/s\n", V, ms, (double)grid * 4 * iters * 36 * 512.0 / ms / 1e9);
}
int main() {
    double *out; (void)hipMalloc(&out, 8 * 256 * 4096);
    run<0>(out); run<1>(out); run<2>(out); run<3>(out); run<4>(out);
    return 0;
}
