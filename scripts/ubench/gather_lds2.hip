// Microbenchmark: static ELL iteration with BOTH the values and the row offsets broadcast from
// per-wave LDS rings (uniform-address ds_read_b128: 2 values resp. 4 offsets per read), software
// pipelined over groups of 8 slots (the ring reads of group g + 1 are issued before the slab reads
// of group g are waited for).  Per slot: 0.5 + 0.25 ring reads, v_add, ds_read_b64, v_fma_f64.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int PIPE>
__global__ __launch_bounds__(1024) void k2ring(double *out, const unsigned *kin, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 64; i += 1024) reinterpret_cast<double *>(smem)[i] = 1.0 + (i & 7);
    double *ring = reinterpret_cast<double *>(smem + 65536) + wave * 64;
    unsigned *kring = reinterpret_cast<unsigned *>(smem + 65536 + 16 * 512) + wave * 64;
    ring[lane] = 0.5 + lane;
    kring[lane] = kin[lane + wave * 64] % 128 * 512;
    __syncthreads();
    const unsigned lane_off = lane * 8;
    double acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0;
    for (int it = 0; it < iters; ++it) {
        u4 kv[2][2]; d2 av[2][4];
        auto load_ring = [&](int g, int slot) {
#pragma unroll
            for (int q = 0; q < 2; ++q) kv[slot][q] = *reinterpret_cast<const u4 *>(kring + g * 8 + q * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) av[slot][q] = *reinterpret_cast<const d2 *>(ring + g * 8 + q * 2);
        };
        load_ring(0, 0);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const int sl = PIPE ? (g & 1) : 0;
            if (!PIPE && g > 0) load_ring(g, 0);
            double x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = *reinterpret_cast<const double *>(smem + kv[sl][e / 4][e % 4] + lane_off);
            if (PIPE && g + 1 < 8) load_ring(g + 1, (g + 1) & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[(g * 8 + e) / 2] = fma(av[sl][e / 2][e % 2], x[e], acc[(g * 8 + e) / 2]);
#pragma unroll
            for (int c = g * 4; c < g * 4 + 4; ++c) asm volatile("" : "+v"(acc[c]));
            __builtin_amdgcn_sched_barrier(0);
        }
        kring[lane] = (kring[lane] + 512) & 0xFFFF;
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s += acc[c];
    out[blockIdx.x * 1024 + tid] = s;
}
int main() {
    double *out; unsigned *kin;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&kin, 4 * 1024);
    unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i * 2654435761u) >> 7;
    hipMemcpy(kin, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000;
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 100 * 1024, 0, out, kin, 10);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 100 * 1024, 0, out, kin, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %8.3f ms   %6.2f cyc / slot / CU\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 16));
    };
    run(k2ring<0>, "values + offsets from LDS rings");
    run(k2ring<1>, "same, ring reads one group ahead");
    return 0;
}
