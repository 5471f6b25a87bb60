// Microbenchmark: "lane-private" gather for K3 on gfx950.  Lane <-> sparse column with its own
// stream entry (value a, row k); the dense slab is column-major in LDS (slab[j][k]); per entry
// the lane does NJ x { ds_read_b64 slab[j][k_lane] ; v_fma_f64 acc[j] += a * x }.
// No broadcast (no v_readlane, no ring).  Row patterns: random rows (bank conflicts) versus
// rows distinct mod 32 within each 32-lane group (conflict-free schedule).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
constexpr int R = 128, NJ = 32;

template <bool B128>
__global__ __launch_bounds__(1024) void klane(double *out, const unsigned *rows, int iters, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *slab = reinterpret_cast<double *>(smem);      // [64 cols][R rows] column-major
    for (int i = tid; i < 64 * R; i += 1024) slab[i] = 1.0 + (i & 7);
    __syncthreads();
    const int j0 = (wave & 1) * NJ;                        // wave's dense-column half
    double acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = 0;
    for (int it = 0; it < iters; ++it) {
        for (int t = 0; t < T; ++t) {
            const unsigned k = rows[(t * 16 + wave) * 64 + lane];   // pre-generated row per (t, wave, lane)
            const double a = 0.5 + k;
            if (B128) {
                // row-major variant: slab2[k][j], 16-byte reads of two adjacent dense columns
                const unsigned char *p = smem + (size_t)k * (64 * 8 + 16) + j0 * 8;
                typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int j = 0; j < NJ; j += 2) {
                    const d2 x = *reinterpret_cast<const d2 *>(p + j * 8);
                    acc[j] = fma(a, x[0], acc[j]);
                    acc[j + 1] = fma(a, x[1], acc[j + 1]);
                }
            } else {
                const double *p = slab + j0 * R + k;
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = fma(a, p[j * R], acc[j]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) s += acc[j];
    out[blockIdx.x * 1024 + tid] = s;
}

int main() {
    const int T = 12, iters = 400;
    double *out; unsigned *rows;
    hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&rows, 4 * T * 16 * 64);
    unsigned *h = (unsigned *)malloc(4 * T * 16 * 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int mode = 0; mode < 4; ++mode) {
        const bool b128 = mode >= 2, free_ = mode & 1;
        srand(1);
        for (int i = 0; i < T * 16 * 64; ++i) {
            int lane = i & 63;
            // conflict-free: residue (mod 32 for b64 column-major / mod 16-lane groups for b128) tied to the lane
            h[i] = free_ ? (b128 ? ((lane & 15) * 8 + rand() % 8) % R : ((lane & 31) + 32 * (rand() % 4)))
                         : rand() % R;
        }
        hipMemcpy(rows, h, 4 * T * 16 * 64, hipMemcpyHostToDevice);
        auto launch = [&](int it) {
            if (b128) hipLaunchKernelGGL(klane<true>, dim3(256), dim3(1024), 96 * 1024, 0, out, rows, it, T);
            else hipLaunchKernelGGL(klane<false>, dim3(256), dim3(1024), 96 * 1024, 0, out, rows, it, T);
        };
        hipFuncSetAttribute(reinterpret_cast<const void *>(&klane<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&klane<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        launch(2);
        hipEventRecord(a); launch(iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // lane-entries x 32 cols per CU: 16 waves x 64 lanes x T x iters;  report cycles per (entry x 64 dense cols) per CU
        const double ent = 16.0 * 64 * T * iters;
        printf("%s, %-13s: %8.3f ms  %6.3f cyc per (entry x 64 dense cols) per CU\n", b128 ? "row-major b128" : "col-major b64 ",
               free_ ? "conflict-free" : "random rows", ms, ms * 1e-3 * 2.4e9 / ent * 2.0);
    }
    return 0;
}
