// (atomics_fill.hip with the element type as a macro: -DTT=float / double)
// Microbenchmark: cost of ds_add_f64 as a function of the number of active lanes and of the
// address pattern (K2's question: does a half-empty LDS atomic cost half?).
//   hipcc --offload-arch=gfx950 -O3 atomics_fill.hip -o atomics_fill && ./atomics_fill
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ unsigned rng(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// PAT 0: random addresses; 1: conflict-free (consecutive per lane); 2: random row, lane column
#ifndef TT
#define TT float
#endif
template <int PAT>
__global__ __launch_bounds__(1024) void k_fill(double *out, int iters, unsigned long long mask) {
    extern __shared__ unsigned char smem_[];
    TT *tile = reinterpret_cast<TT *>(smem_);   // 16384 elements
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1;
    const int lane = threadIdx.x & 63;
    const bool on = (mask >> lane) & 1;
    int idx[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned r = rng(s);
        idx[j] = PAT == 0 ? (r & 16383) : PAT == 1 ? ((j * 64 + lane) & 16383) : (((r & 127) * 128 + lane) & 16383);
    }
    const TT v = (TT)(1.0 + lane);
    for (int it = 0; it < iters; ++it) {
        if (on) {
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(&tile[(idx[j] + it * 129) & 16383], v);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x + 1] = tile[5];
}

int main() {
    double *out; hipMalloc(&out, 8 * 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256, iters = 4000;
    struct M { const char *name; unsigned long long m; } masks[] = {
        {"64 lanes", ~0ull}, {"32 even lanes", 0x5555555555555555ull}, {"32 low lanes", 0xffffffffull},
        {"16 lanes (every 4th)", 0x1111111111111111ull}, {"16 low lanes", 0xffffull},
        {"8 lanes (every 8th)", 0x0101010101010101ull}, {"8 low lanes", 0xffull}, {"1 lane", 1ull}};
    auto run = [&](auto kern, const char *pat, int threads) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        for (auto &m : masks) {
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 131072, 0, out, 10, m.m);
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 131072, 0, out, iters, m.m);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr = (double)(threads / 64) * iters * 8;   // wave instructions per CU
            printf("%-14s %2d waves  %-22s: %7.3f ms  %6.1f cyc per wave-instr per CU (2.1 GHz)\n", pat, threads / 64,
                   m.name, ms, ms * 1e-3 * 2.1e9 / instr);
        }
    };
    for (int threads : {1024}) {
        run(k_fill<0>, "random", threads);
        run(k_fill<1>, "conflict-free", threads);
        run(k_fill<2>, "random row", threads);
    }
    return 0;
}
