// Are LDS atomics to an address beyond the workgroup's allocation dropped (no fault, no effect), and what does a lane
// that is "switched off" that way cost against one masked out through EXEC?  (K2b: a pair slot without a pair.)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics lds_oob.hip -o lds_oob && ./lds_oob
// MODE 0: all 64 lanes add (random columns of a 128 KB tile)          -- reference rate
// MODE 1: half of the lanes (random half per instruction) masked by EXEC (if)
// MODE 2: the same half sent to byte address 0x40000000 | x instead   -- no branch, no v_cmp / s_and_saveexec
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned rng(unsigned &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ __launch_bounds__(1024) void k_lds(double *out, int iters) {
    extern __shared__ double tile[];   // 16384 doubles = 128 KB
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0;
    __syncthreads();
    unsigned s = threadIdx.x * 7919u + blockIdx.x * 104729u + 1;
    char *base = reinterpret_cast<char *>(tile);
    for (int it = 0; it < iters; ++it) {
        const unsigned r = rng(s);
        unsigned off = (r & 16383) << 3;
        const bool on = (r >> 14) & 1;
        if (MODE == 0) {
            atomicAdd(reinterpret_cast<double *>(base + off), 1.0);
        } else if (MODE == 1) {
            if (on) atomicAdd(reinterpret_cast<double *>(base + off), 1.0);
        } else {
            off |= on ? 0u : 0x40000000u;
            asm volatile("ds_add_f64 %0, %1" ::"v"(off), "v"(1.0) : "memory");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // sum of the tile = number of adds that landed
    double acc = 0;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) acc += tile[i];
    atomicAdd(&out[blockIdx.x], acc);
}

int main() {
    double *out; CHECK(hipMalloc(&out, 8 * 256));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256, threads = 1024, iters = 2000;
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 131072, 0, out, 10);
        hipMemset(out, 0, 8 * 256);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 131072, 0, out, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        hipError_t e = hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        double h[256]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        double tot = 0; for (double v : h) tot += v;
        printf("%-44s %8.3f ms   %5.1f cyc per wave-instr/CU   landed %.4f of the lane-ops   (%s)\n", name, ms,
               ms * 1e-3 * 2.4e9 / ((double)threads / 64 * iters), tot / ((double)blocks * threads * iters), hipGetErrorString(e));
    };
    run(k_lds<0>, "all lanes");
    run(k_lds<1>, "half of the lanes masked by EXEC");
    run(k_lds<2>, "half of the lanes sent out of range");
    return 0;
}
