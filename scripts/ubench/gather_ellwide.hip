// Microbenchmark: WIDE static ELL iterations: 16 columns x 4 slots per iteration, slab of 64 rows x
// 128 dense columns in LDS, lane <-> 2 dense columns (one ds_read_b128 + 2 FMAs per nonzero).
// Skip mask over batches of SK slots.  Synthetic (slab, group) blocks with Binomial(64, 0.05) runs;
// reports cycles per REAL nonzero (= 128 dense columns) per CU; the 64-column kernel needs 2 x 6.9.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr unsigned PAD = 0xFFFFFFFFu;

template <int SK>   // slots per skip batch: 8 (4 columns), 4 (2 columns), 64 (no skipping)
__global__ __launch_bounds__(1024) void kell(double *out, const double *vals_all, const unsigned *koff_all,
                                             const int *iters_all, int cap, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65 * 128; i += 1024) reinterpret_cast<double *>(smem)[i] = i < 64 * 128 ? 1.0 + (i & 7) : 0.0;
    double *ring = reinterpret_cast<double *>(smem + 65 * 1024) + wave * 64;
    __syncthreads();
    const double *vals = vals_all + (size_t)wave * cap;
    const unsigned *koff = koff_all + (size_t)wave * cap;
    const int iters = iters_all[wave];
    const unsigned lane_off = lane * 16;
    d2 acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = d2{0, 0};
    for (int rep = 0; rep < reps; ++rep) {
        double na = vals[lane];
        unsigned nk = koff[lane];
        for (int it = 0; it < iters; ++it) {
            // chunk enters: value -> ring, row offset -> register (padding -> the zero row)
            const unsigned long long real = __builtin_amdgcn_ballot_w64(nk != PAD);
            const unsigned vk = nk != PAD ? nk : 64u * 1024u;
            __builtin_amdgcn_wave_barrier();
            ring[lane] = na;
            __builtin_amdgcn_wave_barrier();
            const int nx = min(it + 1, iters - 1) * 64 + lane;
            na = vals[nx];
            nk = koff[nx];
#pragma unroll
            for (int g0 = 0; g0 < 64; g0 += 8) {
                if (SK == 8 && ((real >> g0) & 0xFFull) == 0) continue;
#pragma unroll
                for (int h = 0; h < 8; h += 4) {
                    if (SK == 4 && ((real >> (g0 + h)) & 0xFull) == 0) continue;
#pragma unroll
                    for (int q = 0; q < 4; q += 2) {
                        if (SK == 2 && ((real >> (g0 + h + q)) & 0x3ull) == 0) continue;
                        d2 x[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            x[e] = *reinterpret_cast<const d2 *>(smem + (unsigned)__builtin_amdgcn_readlane((int)vk, g0 + h + q + e) + lane_off);
                        const d2 av = *reinterpret_cast<const d2 *>(ring + g0 + h + q);
                        d2 &A = acc[(g0 + h + q) / 4];
                        A[0] = fma(av[0], x[0][0], A[0]);
                        A[1] = fma(av[0], x[0][1], A[1]);
                        A[0] = fma(av[1], x[1][0], A[0]);
                        A[1] = fma(av[1], x[1][1], A[1]);
                        if (SK == 2) asm volatile("" : "+v"(A));
                    }
                    if (SK != 2) asm volatile("" : "+v"(acc[(g0 + h) / 4]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) s += acc[c][0] + acc[c][1];
    out[blockIdx.x * 1024 + tid] = s;
}

int main() {
    const int cap = 64 * 40, reps = 600;
    std::vector<double> hv(16 * cap, 0.0); std::vector<unsigned> hk(16 * cap, PAD); std::vector<int> hi(16);
    srand(3);
    double real = 0, slots = 0;
    for (int w = 0; w < 16; ++w) {
        int cnt[16], mx = 0;
        for (int c = 0; c < 16; ++c) { int n = 0; for (int r = 0; r < 64; ++r) n += (rand() % 100) < 5; cnt[c] = n; if (n > mx) mx = n; real += n; }
        const int iters = mx ? (mx + 3) / 4 : 1; hi[w] = iters; slots += iters * 64;
        for (int c = 0; c < 16; ++c)
            for (int e = 0; e < cnt[c]; ++e) { const int p = (e / 4) * 64 + c * 4 + (e & 3); hv[w * cap + p] = 0.5 + (p & 3); hk[w * cap + p] = (rand() % 64) * 1024; }
    }
    double *out, *dv; unsigned *dk; int *di;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&dv, 8 * hv.size()); hipMalloc(&dk, 4 * hk.size()); hipMalloc(&di, 64);
    hipMemcpy(dv, hv.data(), 8 * hv.size(), hipMemcpyHostToDevice); hipMemcpy(dk, hk.data(), 4 * hk.size(), hipMemcpyHostToDevice);
    hipMemcpy(di, hi.data(), 64, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    printf("real entries per block %.1f, ELL slots per block %.1f (x%.2f)\n", real / 16, slots / 16, slots / real);
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 80 * 1024, 0, out, dv, dk, di, cap, 5);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 80 * 1024, 0, out, dv, dk, di, cap, reps);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-44s %8.3f ms   %6.2f cyc / REAL entry / CU\n", name, ms, ms * 1e-3 * 2.4e9 / (real * reps));
    };
    run(kell<64>, "ELL, no skipping");
    run(kell<8>, "ELL, skip empty batches of 8 slots");
    run(kell<4>, "ELL, skip empty batches of 4 slots");
    run(kell<2>, "ELL, skip empty columns (2 slots)");
    return 0;
}
