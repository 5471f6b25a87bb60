// Microbenchmark: cost of the instruction mix of the K3 gather iteration on gfx950 (per 64-entry
// iteration and wave: 64 x {v_readlane, v_add, ds_read_b64 row, v_fma_f64} + 32 uniform
// ds_read_b128), 16 waves per CU, no global memory traffic.  MODE bits switch parts off.
//   bit0: FMAs   bit1: ring (uniform 16-byte) reads   bit2: slab row reads   bit3: readlane + add
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(1024) void kmix(double *out, const unsigned *kin, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 64; i += 1024) reinterpret_cast<double *>(smem)[i] = 1.0 + (i & 7);
    double *ring = reinterpret_cast<double *>(smem + 65536) + wave * 64;
    ring[lane] = 0.5 + lane;
    __syncthreads();
    unsigned vk = kin[lane + wave * 64] % 128 * 512;
    const int lane_off = lane * 8;
    double acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g0 = 0; g0 < 64; g0 += 8) {
            double x[8], a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                unsigned k = (MODE & 8) ? (unsigned)__builtin_amdgcn_readlane(vk, g0 + e) : (unsigned)((g0 + e) * 512);
                x[e] = (MODE & 4) ? *reinterpret_cast<const double *>(smem + k + lane_off) : (double)k;
            }
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                d2 av = (MODE & 2) ? *reinterpret_cast<const d2 *>(ring + g0 + e) : d2{1.0, 2.0};
                a[e] = av[0]; a[e + 1] = av[1];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (MODE & 1) acc[(g0 + e) / 2] = fma(a[e], x[e], acc[(g0 + e) / 2]);
                else acc[(g0 + e) / 2] += (e == 0 ? a[e] + x[e] : 0.0);
            }
#pragma unroll
            for (int c = g0 / 2; c < (g0 + 8) / 2; ++c) asm volatile("" : "+v"(acc[c]));
            __builtin_amdgcn_sched_barrier(0);
        }
        vk = (vk + 512) & 0xFFFF;
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s += acc[c];
    out[blockIdx.x * 1024 + tid] = s;
}

template <int MODE>
void run(const char *name, double *out, unsigned *kin) {
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void *>(&kmix<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kmix<MODE>, dim3(256), dim3(1024), 140 * 1024, 0, out, kin, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(kmix<MODE>, dim3(256), dim3(1024), 140 * 1024, 0, out, kin, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // cycles per entry per CU: 16 waves x 64 entries per iteration
    printf("%-44s %8.3f ms   %6.2f cyc / wave-entry / CU   (%5.1f cyc per wave per entry)\n", name, ms,
           ms * 1e-3 * 2.4e9 / ((double)iters * 64 * 16), ms * 1e-3 * 2.4e9 / ((double)iters * 64));
}

int main() {
    double *out; unsigned *kin;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&kin, 4 * 1024);
    unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (i * 2654435761u) >> 7;
    hipMemcpy(kin, h, sizeof(h), hipMemcpyHostToDevice);
    run<15>("full mix (readlane+add, row, ring, fma)", out, kin);
    run<14>("no fma", out, kin);
    run<13>("no ring reads", out, kin);
    run<11>("no row reads", out, kin);
    run<7>("no readlane/add (static rows)", out, kin);
    run<5>("rows + fma only", out, kin);
    run<4>("rows only", out, kin);
    run<2>("ring only", out, kin);
    run<1>("fma only", out, kin);
    run<8>("readlane+add only", out, kin);
    return 0;
}
