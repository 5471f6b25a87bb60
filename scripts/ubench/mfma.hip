// Microbenchmark: peak rate of v_mfma_f64_16x16x4_f64 and v_mfma_f32_16x16x4_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k64(double *out, int iters, double a, double b) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, float a, float b) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    void *out; hipMalloc(&out, 8 * 256 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    auto time = [&](auto launch) { launch(10); hipEventRecord(a); launch(iters); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); return ms; };
    for (int bpc = 1; bpc <= 2; ++bpc) {
        int grid = 256 * bpc;
        float ms = time([&](int it) { hipLaunchKernelGGL(k64<8>, dim3(grid), dim3(256), 0, 0, (double *)out, it, 1.0, 2.0); });
        double fl = (double)grid * 4 * iters * 8 * 2.0 * 16 * 16 * 4;
        printf("f64 16x16x4, 8 acc, %d waves/SIMD: %7.2f ms  %7.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", bpc, ms, fl / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)iters * 8 * bpc));
        ms = time([&](int it) { hipLaunchKernelGGL(k32<8>, dim3(grid), dim3(256), 0, 0, (float *)out, it, 1.0f, 2.0f); });
        printf("f32 16x16x4, 8 acc, %d waves/SIMD: %7.2f ms  %7.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", bpc, ms, fl / ms / 1e9,
               ms * 1e-3 * 2.4e9 / ((double)iters * 8 * bpc));
    }
    float ms = time([&](int it) { hipLaunchKernelGGL(k64<2>, dim3(256), dim3(256), 0, 0, (double *)out, it, 1.0, 2.0); });
    printf("f64 16x16x4, 2 acc, 1 wave/SIMD: %.1f cyc/MFMA/SIMD\n", ms * 1e-3 * 2.4e9 / ((double)iters * 2));
    return 0;
}
