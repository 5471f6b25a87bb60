// LDS read-rate microbenchmark (gfx950): cycles per ds_read wave-instruction per CU for the access
// shapes of the K3 gather kernels.  256 workgroups (one per CU) x NW waves; every wave issues
// K independent reads per s_waitcnt, rows picked pseudo-randomly in a 64-row slab of 1 KiB rows.
//   pattern 0: all 64 lanes read one row (lane*W bytes apart)           -- ellw (W=16: 1 KiB)
//   pattern 1: lanes 0-31 one row, lanes 32-63 another (lane&31)*W       -- lane-group kernel
//   pattern 2: the four rows of 16 lanes read four different rows        -- 4 lane groups
// build: hipcc --offload-arch=gfx950 -O3 -o lds_rate lds_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int WB, int PAT, int K, int HALF = 0>
__global__ __launch_bounds__(1024) void krate(unsigned *out, int reps, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 66 * 256; i += blockDim.x) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned base = (unsigned)(uintptr_t)(lds_byte *)smem;
    unsigned grp = PAT == 0 ? 0 : PAT == 1 ? (lane >> 5) : (lane >> 4);
    unsigned lo = PAT == 0 ? lane * WB : PAT == 1 ? (lane & 31) * WB : (lane & 15) * WB;
    unsigned rnd = wave * 7 + grp * 13;
    unsigned sink = 0;
    long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        unsigned a[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            rnd = rnd * 1664525u + 1013904223u;
            a[k] = base + ((rnd >> 16) & 63) * 1024 + lo;
        }
        if constexpr (WB == 16) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 x[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                // HALF 1: only lanes 0-31 execute the read; HALF 2: halves alternate
                if constexpr (HALF == 0) {
                    asm volatile("ds_read_b128 %0, %1" : "=v"(x[k]) : "v"(a[k]));
                } else {
                    const unsigned long long m = (HALF == 2 && (k & 1)) ? 0xFFFFFFFF00000000ull : 0xFFFFFFFFull;
                    asm volatile("s_mov_b64 exec, %2\n\tds_read_b128 %0, %1\n\ts_mov_b64 exec, -1"
                                 : "=v"(x[k]) : "v"(a[k]), "s"(m));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" ::"v"(x[k]));
            sink ^= x[0][0];
        } else if constexpr (WB == 8) {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 x[K];
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(x[k]) : "v"(a[k]));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" ::"v"(x[k]));
            sink ^= x[0][0];
        } else {
            unsigned x[K];
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("ds_read_b32 %0, %1" : "=v"(x[k]) : "v"(a[k]));
            asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("" ::"v"(x[k]));
            sink ^= x[0];
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + tid] = sink;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WB, int PAT, int K, int HALF = 0>
void run(const char *name, int nw, unsigned *out, long long *cyc) {
    const int reps = 20000 / K;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto kern = krate<WB, PAT, K, HALF>;
    hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024);
    hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 66 * 1024, 0, out, 10, cyc);
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 66 * 1024, 0, out, reps, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
    const double instr = (double)reps * K * nw;      // wave instructions per CU
    printf("%-34s waves %2d  K %2d : %6.2f cyc/instr/CU (2.4 GHz wall)  %6.2f (clock64)  %7.1f B/clk/CU\n",
           name, nw, K, ms * 1e-3 * 2.4e9 / instr, (double)c0 / instr,
           64.0 * WB / (ms * 1e-3 * 2.4e9 / instr));
}

int main() {
    unsigned *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    for (int nw : {4, 8, 16}) {
        run<16, 0, 8>("b128 one row / wave", nw, out, cyc);
        run<16, 1, 8>("b128 two rows (halves)", nw, out, cyc);
        run<16, 2, 8>("b128 four rows (16-lane rows)", nw, out, cyc);
        run<8, 0, 8>("b64 one row / wave", nw, out, cyc);
        run<8, 1, 8>("b64 two rows (halves)", nw, out, cyc);
        run<4, 0, 8>("b32 one row / wave", nw, out, cyc);
    }
    run<16, 0, 2>("b128 one row / wave", 16, out, cyc);
    run<16, 0, 4>("b128 one row / wave", 16, out, cyc);
    run<16, 0, 16>("b128 one row / wave", 16, out, cyc);
    run<16, 1, 4>("b128 two rows (halves)", 16, out, cyc);
    run<16, 1, 8, 1>("b128 halves, exec = lanes 0-31", 16, out, cyc);
    run<16, 1, 8, 2>("b128 halves, exec alternates", 16, out, cyc);
    run<16, 1, 8, 1>("b128 halves, exec = lanes 0-31", 8, out, cyc);
    return 0;
}
