// Microbenchmark: throughput of scatter-accumulate primitives on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomics.hip -o atomics && ./atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned rng(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(double *out, int iters, int spread) {
    extern __shared__ double tile[];   // 16384 doubles = 128 KB
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        int idx;
        if (spread == 0) idx = (rng(s) & 16383);                       // random per lane
        else idx = ((rng(s) & 255) * 64 + lane) & 16383;               // row-contiguous per wave (conflict-free)
        if (spread == 1) { unsigned r = __shfl(idx, 0, 64); idx = ((r & ~63) + lane) & 16383; }
        double v = 1.0 + lane;
        if (MODE == 0) atomicAdd(&tile[idx], v);                        // ds_add_f64
        else if (MODE == 1) tile[idx] += v;                             // plain RMW (racy, rate only)
        else if (MODE == 2) atomicAdd((float *)&tile[idx], (float)v);   // ds_add_f32
        else if (MODE == 3) atomicAdd((unsigned long long *)&tile[idx], (unsigned long long)lane);  // ds_add_u64
        else if (MODE == 4) v += tile[idx], out[0] = v > 1e300 ? v : out[0];   // read only
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x + 1] = tile[5];
}

__global__ __launch_bounds__(256) void k_glob(double *table, int n_table_mask, int iters) {
    unsigned s = (blockIdx.x * 256 + threadIdx.x) * 7919u + 1;
    for (int it = 0; it < iters; ++it) atomicAdd(&table[rng(s) & n_table_mask], 1.0);
}

int main() {
    double *out; CHECK(hipMalloc(&out, 8 * 4096));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256, threads = 1024, iters = 2000;
    const char *names[] = {"ds_add_f64", "plain RMW f64", "ds_add_f32", "ds_add_u64", "ds_read_b64 only"};
    auto run = [&](auto kern, int mode, int spread) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 131072, 0, out, 10, spread);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 131072, 0, out, iters, spread);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double lanes = (double)blocks * threads * iters;
        printf("LDS %-18s %-14s: %8.3f ms  %7.2f lane-ops/clk/CU (2.4GHz)  %6.1f cyc per wave-instr/CU\n", names[mode],
               spread ? "row-contiguous" : "random", ms, lanes / 256 / (ms * 1e-3 * 2.4e9), 64.0 / (lanes / 256 / (ms * 1e-3 * 2.4e9)));
    };
    for (int spread = 0; spread < 2; ++spread) {
        run(k_lds<0>, 0, spread); run(k_lds<1>, 1, spread); run(k_lds<2>, 2, spread); run(k_lds<3>, 3, spread); run(k_lds<4>, 4, spread);
    }
    for (int logn = 10; logn <= 24; logn += 7) {
        double *table; CHECK(hipMalloc(&table, 8ull << logn)); hipMemset(table, 0, 8ull << logn);
        const int gb = 2048, gi = 500;
        hipLaunchKernelGGL(k_glob, dim3(gb), dim3(256), 0, 0, table, (1 << logn) - 1, 5);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_glob, dim3(gb), dim3(256), 0, 0, table, (1 << logn) - 1, gi);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        double ops = (double)gb * 256 * gi;
        printf("global_atomic_add_f64 random over %8d doubles: %8.3f ms  %8.2f G atomics/s\n", 1 << logn, ms, ops / ms / 1e6);
        hipFree(table);
    }
    return 0;
}
