// Issue rate of the float64 vector FMA forms on gfx950: VOP3 v_fma_f64, VOP2 v_fmac_f64, v_fmac_f64_dpp (row_newbcast),
// 8 independent accumulators per wave, 16 waves per CU (4 per SIMD).  Spec: 78.6 TFLOP/s = one FMA per lane and clock
// = 4 cycles per wave instruction and SIMD.
//   hipcc --offload-arch=gfx950 -O3 fma64_rate.hip -o fma64_rate && ./fma64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(1024) void k(double *out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    double x = 1.0000001, y = 0.9999999;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                         "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (MODE == 1) {
            asm volatile("v_fmac_f64 %0, %8, %9\n\tv_fmac_f64 %1, %8, %9\n\tv_fmac_f64 %2, %8, %9\n\tv_fmac_f64 %3, %8, %9\n\t"
                         "v_fmac_f64 %4, %8, %9\n\tv_fmac_f64 %5, %8, %9\n\tv_fmac_f64 %6, %8, %9\n\tv_fmac_f64 %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (MODE == 2) {
            asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else if (MODE == 3) {   // v_mul_f64
            asm volatile("v_mul_f64 %0, %8, %0\n\tv_mul_f64 %1, %8, %1\n\tv_mul_f64 %2, %8, %2\n\tv_mul_f64 %3, %8, %3\n\t"
                         "v_mul_f64 %4, %8, %4\n\tv_mul_f64 %5, %8, %5\n\tv_mul_f64 %6, %8, %6\n\tv_mul_f64 %7, %8, %7"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y));
        } else {                  // v_fma_f32 for reference
            float f0 = (float)a0, f1 = (float)a1;
            asm volatile("v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1\n\t"
                         "v_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1\n\tv_fma_f32 %0, %2, %3, %0\n\tv_fma_f32 %1, %2, %3, %1"
                         : "+v"(f0), "+v"(f1) : "v"((float)x), "v"((float)y));
            a0 = f0; a1 = f1;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

int main() {
    double *out; hipMalloc(&out, 8 * 256 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    const char *names[] = {"v_fma_f64 (VOP3)", "v_fmac_f64 (VOP2)", "v_fmac_f64_dpp row_newbcast", "v_mul_f64", "v_fma_f32 (2 chains)"};
    auto run = [&](auto kern, int mode, int threads) {
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, 10);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double instr_per_simd = (double)iters * 8 * (threads / 64) / 4.0;
        printf("%-30s %2d waves/CU: %8.3f ms   %5.2f cycles per wave instruction and SIMD (2.4 GHz)\n", names[mode], threads / 64, ms,
               ms * 1e-3 * 2.4e9 / instr_per_simd);
    };
    for (int threads : {1024, 512, 256}) {
        run(k<0>, 0, threads); run(k<1>, 1, threads); run(k<2>, 2, threads); run(k<3>, 3, threads); run(k<4>, 4, threads);
    }
    return 0;
}
