// Microbenchmark (round 4): the K3 gather with DYNAMIC accumulator registers.
//
// The lane-group kernel keeps a sparse column's accumulators in static registers and therefore walks its
// stream column by column, padded to the longer column of a pair (1.5 LDS reads per nonzero) behind scalar
// bit tests.  Here an ENTRY {value, row, column} picks its accumulator at run time with the VGPR index
// mode of gfx9 (s_set_gpr_idx_on: M0[7:0] is added to the register number of the enabled operands), so
// the stream is a plain list of entries in batches of 16 -- no padding inside a batch, no branches:
//     v_add_u32_dpp   (row offset of entry i, broadcast in every row of 16 lanes)      1 VALU
//     ds_read_b128    (lane <-> 2 of the 128 dense columns: the whole wave reads the 1 KiB row)
//     v_readlane_b32  (4 * column -> SGPR)                                             1 VALU
//     s_set_gpr_idx_on / 2 x v_fmac_f64_dpp (value by row_newbcast) / s_set_gpr_idx_off
// Reports cycles per entry and CU (the shipped kernel: 13.2 at cfg4; LDS floor 4.4).
// MODE 0: index mode + fmac_dpp; 1: the same stream, static accumulator (no index instructions) -- the
// price of the mode switches; 2: index mode + VOP3 v_fma_f64 with the value in SGPRs (2 more readlanes).
// build: hipcc --offload-arch=gfx950 -O3 -o gather_idx gather_idx.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>

typedef double d16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x32 __attribute__((ext_vector_type(32)));

template <int N, typename Fn>
__device__ __forceinline__ void sfor(Fn &&f) {
    if constexpr (N > 0) {
        sfor<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

#if defined(__HIP_DEVICE_COMPILE__)
template <int I, bool HALF = false>
__device__ __forceinline__ void issue_read(u32x32 &X, unsigned kq, unsigned lane_off) {
    unsigned tmp;
    if constexpr (HALF)
        asm volatile("v_add_u32_dpp %1, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                     "ds_read_b64 v[32+4*%5:32+4*%5+1], %1"
                     : "+{v[32:63]}"(X), "=&v"(tmp)
                     : "v"(kq), "v"(lane_off >> 1), "n"(I), "n"(I % 8));
    else
    asm volatile("v_add_u32_dpp %1, %2, %3 row_newbcast:%4 row_mask:0xf bank_mask:0xf\n\t"
                 "ds_read_b128 v[32+4*%5:32+4*%5+3], %1"
                 : "+{v[32:63]}"(X), "=&v"(tmp)
                 : "v"(kq), "v"(lane_off), "n"(I), "n"(I % 8));
}
// MODE 6: the batch's 16 LDS addresses are formed first; the index mode then stays ON over the whole batch
// (s_set_gpr_idx_idx changes the index only; DS instructions are not indexed)
template <int I>
__device__ __forceinline__ unsigned make_addr(unsigned kq, unsigned lane_off) {
    unsigned r;
    asm volatile("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(kq), "v"(lane_off), "n"(I));
    return r;
}
template <int I>
__device__ __forceinline__ void read_at(u32x32 &X, unsigned addr) {
    asm volatile("ds_read_b128 v[32+4*%2:32+4*%2+3], %1" : "+{v[32:63]}"(X) : "v"(addr), "n"(I % 8));
}
template <int I, int WAIT>
__device__ __forceinline__ void fma_idx(d16 &T0, d16 &T1, u32x32 &X, int sj, double a) {
    asm volatile("s_waitcnt lgkmcnt(%6)\n\t"
                 "s_set_gpr_idx_idx %3\n\t"
                 "v_fmac_f64_dpp v[64:65], %4, v[32+4*%5:32+4*%5+1] row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp v[66:67], %4, v[32+4*%5+2:32+4*%5+3] row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                 : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                 : "s"(sj), "v"(a), "n"(I % 8), "n"(WAIT), "n"(I));
}
template <int I>
__device__ __forceinline__ int lane_to_s(unsigned v) {
    int s;
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(v), "n"(I));
    return s;
}
template <int MODE, int I, int WAIT>
__device__ __forceinline__ void fma_entry(d16 &T0, d16 &T1, u32x32 &X, int sj, double a, int slo, int shi) {
    if constexpr (MODE == 0) {
        asm volatile("s_waitcnt lgkmcnt(%6)\n\t"
                     "s_set_gpr_idx_on %3, 0xc\n\t"
                     "v_fmac_f64_dpp v[64:65], %4, v[32+4*%5:32+4*%5+1] row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp v[66:67], %4, v[32+4*%5+2:32+4*%5+3] row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "s_set_gpr_idx_off"
                     : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                     : "s"(sj), "v"(a), "n"(I % 8), "n"(WAIT), "n"(I));
    } else if constexpr (MODE == 1) {
        asm volatile("s_waitcnt lgkmcnt(%6)\n\t"
                     "v_fmac_f64_dpp v[64:65], %4, v[32+4*%5:32+4*%5+1] row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp v[66:67], %4, v[32+4*%5+2:32+4*%5+3] row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                     : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                     : "s"(sj), "v"(a), "n"(I % 8), "n"(WAIT), "n"(I));
    } else if constexpr (MODE == 3) {      // static accumulator, plain fmac (no DPP)
        asm volatile("s_waitcnt lgkmcnt(%6)\n\t"
                     "v_fmac_f64 v[64:65], %4, v[32+4*%5:32+4*%5+1]\n\t"
                     "v_fmac_f64 v[66:67], %4, v[32+4*%5+2:32+4*%5+3]"
                     : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                     : "s"(sj), "v"(a), "n"(I % 8), "n"(WAIT), "n"(I));
    } else if constexpr (MODE == 7) {      // half rows: one fmac per entry (reads are b64, see issue_read)
        asm volatile("s_waitcnt lgkmcnt(%6)\n\t"
                     "v_fmac_f64_dpp v[64:65], %4, v[32+4*%5:32+4*%5+1] row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                     : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                     : "s"(sj), "v"(a), "n"(I % 8), "n"(WAIT), "n"(I));
    } else if constexpr (MODE == 4 || MODE == 8) {      // LDS reads only
        asm volatile("s_waitcnt lgkmcnt(%6)"
                     : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                     : "s"(sj), "v"(a), "n"(I % 8), "n"(WAIT), "n"(I));
    } else if constexpr (MODE == 5) {      // index mode + fmac_dpp, no waiting for LDS (issue only)
        asm volatile("s_set_gpr_idx_on %3, 0xc\n\t"
                     "v_fmac_f64_dpp v[64:65], %4, v[32+4*%5:32+4*%5+1] row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp v[66:67], %4, v[32+4*%5+2:32+4*%5+3] row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                     "s_set_gpr_idx_off"
                     : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                     : "s"(sj), "v"(a), "n"(I % 8), "n"(WAIT), "n"(I));
    } else {
        long long sv = ((long long)(unsigned)shi << 32) | (unsigned)slo;
        asm volatile("s_waitcnt lgkmcnt(%6)\n\t"
                     "s_set_gpr_idx_on %3, 0xc\n\t"
                     "v_fma_f64 v[64:65], %4, v[32+4*%5:32+4*%5+1], v[64:65]\n\t"
                     "v_fma_f64 v[66:67], %4, v[32+4*%5+2:32+4*%5+3], v[66:67]\n\t"
                     "s_set_gpr_idx_off"
                     : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                     : "s"(sj), "s"(sv), "n"(I % 8), "n"(WAIT), "n"(I));
    }
}
#else
template <int I, bool HALF = false> void issue_read(u32x32 &, unsigned, unsigned) {}
template <int I> unsigned make_addr(unsigned, unsigned) { return 0; }
template <int I> void read_at(u32x32 &, unsigned) {}
template <int I, int WAIT> void fma_idx(d16 &, d16 &, u32x32 &, int, double) {}
template <int I> int lane_to_s(unsigned) { return 0; }
template <int MODE, int I, int WAIT> void fma_entry(d16 &, d16 &, u32x32 &, int, double, int, int) {}
#endif

// meta = (1 + row) << 10 | 4 * column
template <int MODE>
__global__ __launch_bounds__(1024) void kidx(double *out, const double *vals_all, const unsigned *meta_all, int nb,
                                             int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nth = blockDim.x;
    for (int i = tid; i < 65 * 128; i += nth)
        reinterpret_cast<double *>(smem)[i] = i < 128 ? 0.0 : 1.0 + ((i * 7) & 15) * 0.125;
    __syncthreads();
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_byte *)smem;
    const double *vals = vals_all + (size_t)wave * nb * 16;
    const unsigned *meta = meta_all + (size_t)wave * nb * 16;
    const unsigned lane_off = lane * 16;
    d16 T0, T1;
    u32x32 X;
#pragma unroll
    for (int c = 0; c < 16; ++c) { T0[c] = 0.0; T1[c] = 0.0; }
#pragma unroll
    for (int c = 0; c < 32; ++c) X[c] = 0u;
    for (int rep = 0; rep < reps; ++rep) {
        double na = vals[lane & 15];
        unsigned nm = meta[lane & 15];
        for (int b = 0; b < nb; ++b) {
            const double a = na;
            const unsigned m = nm;
            const int nx = min(b + 1, nb - 1) * 16 + (lane & 15);
            na = vals[nx];
            nm = meta[nx];
            const unsigned kq = lds_base + (m >> 10 << 10);
            const unsigned jv = m & 0x3ffu;
            int sj[16], slo[16], shi[16];
            sfor<16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                sj[i] = lane_to_s<i>(jv);
                if constexpr (MODE == 2) {
                    slo[i] = lane_to_s<i>((unsigned)__double2loint(a));
                    shi[i] = lane_to_s<i>((unsigned)__double2hiint(a));
                } else {
                    slo[i] = shi[i] = 0;
                }
            });
            if constexpr (MODE == 6) {
                unsigned ad[16];
                sfor<16>([&](auto ic) { ad[decltype(ic)::value] = make_addr<decltype(ic)::value>(kq, lane_off); });
                sfor<8>([&](auto ic) { read_at<decltype(ic)::value>(X, ad[decltype(ic)::value]); });
                asm volatile("s_set_gpr_idx_on %0, 0xc" :: "s"(sj[0]) : "m0");
                sfor<16>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    constexpr int wait = i <= 8 ? 7 : 15 - i;
                    fma_idx<i, wait>(T0, T1, X, sj[i], a);
                    if constexpr (i + 8 < 16) read_at<i + 8>(X, ad[i + 8]);
                });
                asm volatile("s_set_gpr_idx_off");
            } else {
            constexpr bool HB = MODE == 7 || MODE == 8;
            if constexpr (MODE != 5) sfor<8>([&](auto ic) { issue_read<decltype(ic)::value, HB>(X, kq, lane_off); });
            sfor<16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int wait = i <= 8 ? 7 : 15 - i;
                fma_entry<MODE, i, wait>(T0, T1, X, sj[i], a, slo[i], shi[i]);
                if constexpr (i + 8 < 16 && MODE != 5) issue_read<i + 8, HB>(X, kq, lane_off);
            });
            }
        }
    }
    double *o = out + ((size_t)blockIdx.x * (nth / 64) + wave) * 16 * 128;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        // element c of T0 = VGPR pair 64 + 2c: column c / 2, part c & 1
        o[(c / 2) * 128 + lane * 2 + (c & 1)] = T0[c];
        o[(8 + c / 2) * 128 + lane * 2 + (c & 1)] = T1[c];
    }
}

int main() {
    const int nb = 64, reps = 400, NW = 16;
    std::vector<double> hv((size_t)NW * nb * 16);
    std::vector<unsigned> hm((size_t)NW * nb * 16);
    srand(5);
    for (size_t e = 0; e < hv.size(); ++e) {
        const int row = rand() % 64, j = rand() % 16;
        hv[e] = 0.25 * (1 + rand() % 7);
        hm[e] = (unsigned)((1 + row) << 10) | (unsigned)(4 * j);
    }
    // expected result of ONE pass (reps = 1) per wave
    auto bval = [](int row, int c) { const int i = (1 + row) * 128 + c; return 1.0 + ((i * 7) & 15) * 0.125; };
    std::vector<double> ref((size_t)NW * 16 * 128, 0.0);
    for (int w = 0; w < NW; ++w)
        for (int e = 0; e < nb * 16; ++e) {
            const size_t q = (size_t)w * nb * 16 + e;
            const int row = (int)(hm[q] >> 10) - 1, j = (int)(hm[q] & 0x3ff) / 4;
            for (int c = 0; c < 128; ++c) ref[((size_t)w * 16 + j) * 128 + c] += hv[q] * bval(row, c);
        }
    double *out, *dv;
    unsigned *dm;
    hipMalloc(&out, 8 * (size_t)256 * NW * 16 * 128);
    hipMalloc(&dv, 8 * hv.size());
    hipMalloc(&dm, 4 * hm.size());
    hipMemcpy(dv, hv.data(), 8 * hv.size(), hipMemcpyHostToDevice);
    hipMemcpy(dm, hm.data(), 4 * hm.size(), hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<double> ho((size_t)NW * 16 * 128);
    auto run = [&](auto kern, const char *name, int nw, bool check) {
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024);
        hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 66 * 1024, 0, out, dv, dm, nb, 1);
        hipDeviceSynchronize();
        if (check) {
            hipMemcpy(ho.data(), out, 8 * (size_t)nw * 16 * 128, hipMemcpyDeviceToHost);
            double err = 0;
            for (size_t i = 0; i < (size_t)nw * 16 * 128; ++i) err = fmax(err, fabs(ho[i] - ref[i]));
            printf("%-40s max abs error vs host %.3e %s\n", name, err, err < 1e-9 ? "OK" : "WRONG");
        }
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 66 * 1024, 0, out, dv, dm, nb, reps);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        const double entries = (double)nw * nb * 16 * reps;
        printf("%-40s waves %2d: %8.3f ms  %6.2f cyc / entry / CU\n", name, nw, ms, ms * 1e-3 * 2.4e9 / entries);
    };
    for (int nw : {16, 8, 4}) {
        run(kidx<0>, "index mode, fmac_dpp", nw, true);
        run(kidx<1>, "static accumulator (wrong sums)", nw, false);
        run(kidx<2>, "index mode, VOP3 fma, SGPR value", nw, true);
        run(kidx<6>, "index mode on over the batch, idx_idx", nw, true);
        run(kidx<3>, "static accumulator, fmac without DPP", nw, false);
        run(kidx<4>, "LDS reads only (no FMA)", nw, false);
        run(kidx<7>, "HALF rows: ds_read_b64 + 1 fmac (static)", nw, false);
        run(kidx<8>, "HALF rows: ds_read_b64 only", nw, false);
        run(kidx<5>, "index mode + fmac_dpp only (no LDS)", nw, false);
    }
    return 0;
}
