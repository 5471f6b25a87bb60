// Microbenchmark: the DYNAMIC per-column run loop of the K3 gather kernel in isolation (no slab
// staging, no workgroup barrier): 16 waves per CU, each wave walks a synthetic (slab, group) stream
// with Binomial(128, 0.05) run lengths for 32 static columns, again and again.
//   MODE 0: as in sparse.hip (batches of 4 + one-entry tail loop, value ring + v_readlane offsets)
//   MODE 1: batches of 2 only
//   MODE 2: value from a constant (no ring reads)
//   MODE 3: row offset from a constant (no v_readlane)
//   MODE 4: neither (slab read + fma + loop control only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
constexpr int CPW = 32;

template <int MODE, int C>
struct Col {
    static __device__ __forceinline__ void run(double (&acc)[CPW], const unsigned char *slab, const double *ring,
                                               unsigned &kcur, int cntv, int &pos, double &na, unsigned &nk,
                                               const double *vals, const unsigned *koff, int total, int lane, int lane_off) {
        int nc = __builtin_amdgcn_readlane(cntv, C);
        if (MODE == 7 || MODE == 8) {
            // branch-free: NB masked batches of 4 per column (entries beyond are dropped: timing only),
            // products of slots >= nc replaced by 0 with selects
            constexpr int NB = MODE == 7 ? 2 : 3;
            const int l0 = pos & 63;
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                double av[4], xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int L = (l0 + 4 * b + i) & 63;
                    av[i] = ring[L];
                    const unsigned K = (unsigned)__builtin_amdgcn_readlane((int)kcur, L);
                    xv[i] = *reinterpret_cast<const double *>(slab + K + lane_off);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ok = 4 * b + i < nc;
                    acc[C] = fma(ok ? av[i] : 0.0, ok ? xv[i] : 0.0, acc[C]);
                }
            }
            pos += min(nc, 4 * NB);
            if constexpr (C + 1 < CPW) Col<MODE, C + 1>::run(acc, slab, ring, kcur, cntv, pos, na, nk, vals, koff, total, lane, lane_off);
            return;
        }
        while (nc > 0) {
            const int l0 = pos & 63;
            const int m = MODE >= 5 ? nc : min(nc, 64 - l0);
            int t = l0;
            const int tend = l0 + m;
#define STEP(A, X, L)                                                                              \
    const double A = (MODE == 2 || MODE == 4 || MODE == 5) ? 1.5 : ring[(L) & 63];                 \
    const unsigned K##A = (MODE == 3 || MODE == 4 || MODE == 5) ? (unsigned)(((L) & 127) * 512)     \
                                                   : (unsigned)__builtin_amdgcn_readlane((int)kcur, (L) & 63); \
    const double X = *reinterpret_cast<const double *>(slab + K##A + lane_off);
            constexpr int B = MODE == 1 ? 2 : 4;
            for (; t + B <= tend; t += B) {
                STEP(a0, x0, t) STEP(a1, x1, t + 1)
                acc[C] = fma(a0, x0, acc[C]);
                acc[C] = fma(a1, x1, acc[C]);
                if (B == 4) {
                    STEP(a2, x2, t + 2) STEP(a3, x3, t + 3)
                    acc[C] = fma(a2, x2, acc[C]);
                    acc[C] = fma(a3, x3, acc[C]);
                }
            }
            if (MODE == 9) {
                const int rem = tend - t;
                if (rem == 3) {
                    STEP(a0, x0, t) STEP(a1, x1, t + 1) STEP(a2, x2, t + 2)
                    acc[C] = fma(a0, x0, acc[C]); acc[C] = fma(a1, x1, acc[C]); acc[C] = fma(a2, x2, acc[C]);
                } else if (rem == 2) {
                    STEP(a0, x0, t) STEP(a1, x1, t + 1)
                    acc[C] = fma(a0, x0, acc[C]); acc[C] = fma(a1, x1, acc[C]);
                } else if (rem == 1) {
                    STEP(a0, x0, t)
                    acc[C] = fma(a0, x0, acc[C]);
                }
            } else {
            for (; t < tend; ++t) {
                STEP(a0, x0, t)
                acc[C] = fma(a0, x0, acc[C]);
            }
            }
            pos += m;
            nc -= m;
            if (MODE < 5 && (pos & 63) == 0) {
                __builtin_amdgcn_wave_barrier();
                const_cast<double *>(ring)[lane] = na;
                kcur = nk;
                __builtin_amdgcn_wave_barrier();
                const int nxt = pos + 64 + lane;
                if (nxt < total) { na = vals[nxt]; nk = koff[nxt]; }
            }
        }
        if constexpr (C + 1 < CPW) Col<MODE, C + 1>::run(acc, slab, ring, kcur, cntv, pos, na, nk, vals, koff, total, lane, lane_off);
    }
};

template <int MODE>
__global__ __launch_bounds__(1024) void kdyn(double *out, const double *vals_all, const unsigned *koff_all,
                                             const int *cnt_all, const int *total_all, int stream_cap, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 128 * 64; i += 1024) reinterpret_cast<double *>(smem)[i] = 1.0 + (i & 7);
    double *ring = reinterpret_cast<double *>(smem + 65536) + wave * 64;
    __syncthreads();
    const double *vals = vals_all + (size_t)wave * stream_cap;
    const unsigned *koff = koff_all + (size_t)wave * stream_cap;
    const int total = total_all[wave];
    const int cntv = lane < CPW ? cnt_all[wave * CPW + lane] : 0;
    double acc[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) acc[c] = 0;
    const int lane_off = lane * 8;
    for (int it = 0; it < iters; ++it) {
        int pos = 0;
        unsigned kcur = lane < total ? koff[lane] : 0;
        ring[lane] = lane < total ? vals[lane] : 0.0;
        double na = 64 + lane < total ? vals[64 + lane] : 0.0;
        unsigned nk = 64 + lane < total ? koff[64 + lane] : 0;
        __builtin_amdgcn_wave_barrier();
        Col<MODE, 0>::run(acc, smem, ring, kcur, cntv, pos, na, nk, vals, koff, total, lane, lane_off);
    }
    double s = 0;
#pragma unroll
    for (int c = 0; c < CPW; ++c) s += acc[c];
    out[blockIdx.x * 1024 + tid] = s;
}

int main() {
    const int cap = 512, iters = 600;
    std::vector<double> hv(16 * cap); std::vector<unsigned> hk(16 * cap); std::vector<int> hc(16 * CPW), ht(16);
    srand(3);
    double entries = 0;
    for (int w = 0; w < 16; ++w) {
        int p = 0;
        for (int c = 0; c < CPW; ++c) {
            int n = 0; for (int r = 0; r < 128; ++r) n += (rand() % 100) < 5;
            hc[w * CPW + c] = n;
            for (int e = 0; e < n; ++e, ++p) { hv[w * cap + p] = 0.5 + (p & 3); hk[w * cap + p] = (rand() % 128) * 512; }
        }
        ht[w] = p; entries += p;
    }
    double *out, *dv; unsigned *dk; int *dc, *dt;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&dv, 8 * hv.size()); hipMalloc(&dk, 4 * hk.size()); hipMalloc(&dc, 4 * hc.size()); hipMalloc(&dt, 64);
    hipMemcpy(dv, hv.data(), 8 * hv.size(), hipMemcpyHostToDevice); hipMemcpy(dk, hk.data(), 4 * hk.size(), hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), 4 * hc.size(), hipMemcpyHostToDevice); hipMemcpy(dt, ht.data(), 64, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char *name) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 80 * 1024, 0, out, dv, dk, dc, dt, cap, 5);
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), 80 * 1024, 0, out, dv, dk, dc, dt, cap, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-52s %8.3f ms   %6.2f cyc / wave-entry / CU\n", name, ms, ms * 1e-3 * 2.4e9 / (entries * iters));
    };
    printf("avg entries per (slab, group) stream: %.1f\n", entries / 16);
    run(kdyn<0>, "as in sparse.hip (batch 4 + tail, ring + readlane)");
    run(kdyn<1>, "batches of 2");
    run(kdyn<2>, "no value ring");
    run(kdyn<3>, "no v_readlane");
    run(kdyn<4>, "neither (slab read + fma + loop control)");
    run(kdyn<9>, "as MODE 0 with the tail as straight-line code (switch on 1..3)");
    run(kdyn<7>, "branch-free: 2 masked batches of 4 per column (8 slots)");
    run(kdyn<8>, "branch-free: 3 masked batches of 4 per column (12 slots)");
    run(kdyn<5>, "neither, no chunk rotation / boundary split");
    run(kdyn<6>, "ring + readlane, no chunk rotation / boundary split");
    return 0;
}
