// f64 MFMA rate on gfx950: 16x16x4 vs 4x4x4 (4 blocks), 1..4 waves per SIMD, NACC independent
// accumulators back to back.  Prints TFLOP/s and cycles per MFMA and SIMD from clock64.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int SHAPE>
__global__ __launch_bounds__(256) void k64(double *out, int iters, double a, double b, long long *cyc) {
    d4 acc[NACC];
    double acc1[NACC];
    for (int i = 0; i < NACC; ++i) { acc[i] = d4{0, 0, 0, 0}; acc1[i] = 0; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (SHAPE == 16) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            else acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC, int SHAPE>
void run(int bpc, double *out, long long *cyc) {
    const int iters = 20000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int grid = 256 * bpc;
    hipLaunchKernelGGL((k64<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, 10, 1.0, 2.0, cyc);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k64<NACC, SHAPE>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 2.0, cyc);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double fl_per = SHAPE == 16 ? 2.0 * 16 * 16 * 4 : 2.0 * 4 * 4 * 4 * 4;
    const double n_mfma_simd = (double)iters * NACC * bpc;
    printf("f64 %2dx%2dx4  %d acc  %d waves/SIMD: %7.3f ms  %6.2f TFLOP/s   %6.1f clk64/MFMA/SIMD  (clock %.2f GHz)\n", SHAPE, SHAPE, NACC,
           bpc, ms, (double)grid * 4 * iters * NACC * fl_per / ms / 1e9, (double)c / ((double)iters * NACC * bpc) ,
           (double)c / (ms * 1e6));
}
int main() {
    double *out; long long *cyc;
    (void)hipMalloc(&out, 8 * 256 * 4096); (void)hipMalloc(&cyc, 8 * 4096);
    for (int bpc : {1, 2, 4}) { run<8, 16>(bpc, out, cyc); run<8, 4>(bpc, out, cyc); }
    run<4, 16>(4, out, cyc); run<2, 16>(4, out, cyc); run<16, 4>(2, out, cyc);
    return 0;
}
