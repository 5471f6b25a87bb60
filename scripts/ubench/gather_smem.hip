// Microbenchmark: static (ELL) gather iteration with the stream broadcast through SCALAR loads:
// per batch of 8 slots two s_load (16 dwords of values, 8 dwords of row offsets) feed
// 8 x { v_add (LDS address = lane offset + SGPR), ds_read_b64 slab row, v_fma_f64 with the value as
// SGPR operand }.  16 waves per CU; the stream of a wave is re-read every iteration (scalar cache /
// L2 resident, like a real stream it is never in LDS).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
struct __attribute__((aligned(128))) Batch {   // 8 slots
    double a[8];
    unsigned k[8];
    unsigned pad[8];
};

template <int PF>
__global__ __launch_bounds__(1024) void ksmem(double *__restrict__ out, const Batch *__restrict__ stream, int batches_per_wave, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 128 * 64; i += 1024) reinterpret_cast<double *>(smem)[i] = 1.0 + (i & 7);
    __syncthreads();
    const Batch *sw = stream + (size_t)(blockIdx.x * 16 + wave) * batches_per_wave;
    double acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0;
    const unsigned base = lane * 8;
    for (int it = 0; it < iters; ++it) {
        for (int b0 = 0; b0 < batches_per_wave; b0 += 8) {     // one "iteration" = 8 batches = 64 slots
            double ca[8]; unsigned ck[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { ca[e] = sw[b0].a[e]; ck[e] = sw[b0].k[e]; }
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                double na[8]; unsigned nk[8];
                const Batch *pn = sw + b0 + (PF ? (b + 1 < 8 ? b + 1 : 7) : b);
                if (PF) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { na[e] = pn->a[e]; nk[e] = pn->k[e]; }
                }
                double x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = *reinterpret_cast<const double *>(smem + ck[e] + base);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[(b * 8 + e) / 2] = fma(ca[e], x[e], acc[(b * 8 + e) / 2]);
#pragma unroll
                for (int c = b * 4; c < b * 4 + 4; ++c) asm volatile("" : "+v"(acc[c]));
                __builtin_amdgcn_sched_barrier(0);
                if (PF) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ca[e] = na[e]; ck[e] = nk[e]; }
                } else if (b + 1 < 8) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ca[e] = sw[b0 + b + 1].a[e]; ck[e] = sw[b0 + b + 1].k[e]; }
                }
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 32; ++c) s += acc[c];
    out[blockIdx.x * 1024 + tid] = s;
}

int main() {
    const int bpw = 48, iters = 300;      // 48 batches = 6 iterations of 64 slots per wave pass
    const size_t nb = (size_t)256 * 16 * bpw;
    std::vector<Batch> h(nb);
    unsigned s = 12345;
    for (size_t i = 0; i < nb; ++i)
        for (int e = 0; e < 8; ++e) { s = s * 1664525u + 1013904223u; h[i].a[e] = 0.5 + (s >> 28); h[i].k[e] = ((s >> 8) % 128) * 512; }
    Batch *d; double *out;
    hipMalloc(&d, nb * sizeof(Batch)); hipMalloc(&out, 8 * 1024 * 256);
    hipMemcpy(d, h.data(), nb * sizeof(Batch), hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 80 * 1024, 0, out, d, bpw, 3);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 80 * 1024, 0, out, d, bpw, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-40s %8.3f ms   %6.2f cyc / slot / CU\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)bpw * 8 * 16 * iters));
    };
    run(ksmem<0>, "scalar-load stream, no prefetch");
    run(ksmem<1>, "scalar-load stream, next batch prefetched");
    return 0;
}
