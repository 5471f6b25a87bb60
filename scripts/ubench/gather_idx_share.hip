// Microbenchmark (round 6): K3 batches in which TWO entries of the same row share one LDS read of the row of B
// (derived from gather_idx.hip, round 4: the K3 gather with DYNAMIC accumulator registers).
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


// FMA pair of entry I reading landing slot S (= I for single batches, I / 2 for pair batches)
template <int I, int S, int WAIT>
__device__ __forceinline__ void fma_idx_s(d16 &T0, d16 &T1, u32x32 &X, int sj, double a) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(%6)\n\t"
                 "s_set_gpr_idx_idx %3\n\t"
                 "v_fmac_f64_dpp v[64:65], %4, v[32+4*%5:32+4*%5+1] row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f64_dpp v[66:67], %4, v[32+4*%5+2:32+4*%5+3] row_newbcast:%7 row_mask:0xf bank_mask:0xf"
                 : "+{v[64:95]}"(T0), "+{v[96:127]}"(T1), "+{v[32:63]}"(X)
                 : "s"(sj), "v"(a), "n"(S % 8), "n"(WAIT), "n"(I));
#endif
}
template <int I, int S>
__device__ __forceinline__ void read_at_s(u32x32 &X, unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("ds_read_b128 v[32+4*%2:32+4*%2+3], %1" : "+{v[32:63]}"(X) : "v"(addr), "n"(S % 8));
#endif
}

// meta = (1 + row) << 10 | 4 * column.  SHARE = 1: every entry its own read (the shipped batch); 2: entries 2k and
// 2k + 1 of a batch are in the same row and share the read of entry 2k; 4: quads share
template <int SHARE>
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
    constexpr int NR = 16 / SHARE;            // reads per batch
    constexpr int FL = NR < 8 ? NR : 8;       // reads in flight
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
            int sj[16];
            sfor<16>([&](auto ic) { sj[decltype(ic)::value] = lane_to_s<decltype(ic)::value>(jv); });
            unsigned ad[NR];
            sfor<NR>([&](auto ic) { ad[decltype(ic)::value] = make_addr<decltype(ic)::value * SHARE>(kq, lane_off); });
            sfor<FL>([&](auto ic) { read_at_s<0, decltype(ic)::value>(X, ad[decltype(ic)::value]); });
            asm volatile("s_set_gpr_idx_on %0, 0xc" :: "s"(sj[0]) : "m0");
            sfor<16>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                constexpr int r = i / SHARE;                       // the read this entry uses
                constexpr int issued = (r + FL < NR ? r + FL : NR) - 1;      // index of the newest read issued so far
                constexpr int wait = issued - r;                   // reads allowed to be outstanding
                fma_idx_s<i, r, wait>(T0, T1, X, sj[i], a);
                if constexpr ((i % SHARE) == SHARE - 1 && r + FL < NR) read_at_s<0, r + FL>(X, ad[r + FL]);
            });
            asm volatile("s_set_gpr_idx_off");
        }
    }
    double *o = out + ((size_t)blockIdx.x * (nth / 64) + wave) * 16 * 128;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        o[(c / 2) * 128 + lane * 2 + (c & 1)] = T0[c];
        o[(8 + c / 2) * 128 + lane * 2 + (c & 1)] = T1[c];
    }
}

int main() {
    const int nb = 64, reps = 400, NW = 16;
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    double *out, *dv;
    unsigned *dm;
    hipMalloc(&out, 8 * (size_t)256 * NW * 16 * 128);
    hipMalloc(&dv, 8 * (size_t)NW * nb * 16);
    hipMalloc(&dm, 4 * (size_t)NW * nb * 16);
    auto bval = [](int row, int c) { const int i = (1 + row) * 128 + c; return 1.0 + ((i * 7) & 15) * 0.125; };
    auto run = [&](auto kern, const char *name, int share) {
        std::vector<double> hv((size_t)NW * nb * 16);
        std::vector<unsigned> hm((size_t)NW * nb * 16);
        srand(5);
        int row = 0;
        for (size_t e = 0; e < hv.size(); ++e) {
            if (e % share == 0) row = rand() % 64;
            const int j = rand() % 16;
            hv[e] = 0.25 * (1 + rand() % 7);
            hm[e] = (unsigned)((1 + row) << 10) | (unsigned)(4 * j);
        }
        std::vector<double> ref((size_t)NW * 16 * 128, 0.0);
        for (int w = 0; w < NW; ++w)
            for (int e = 0; e < nb * 16; ++e) {
                const size_t q = (size_t)w * nb * 16 + e;
                const int r = (int)(hm[q] >> 10) - 1, j = (int)(hm[q] & 0x3ff) / 4;
                for (int c = 0; c < 128; ++c) ref[((size_t)w * 16 + j) * 128 + c] += hv[q] * bval(r, c);
            }
        hipMemcpy(dv, hv.data(), 8 * hv.size(), hipMemcpyHostToDevice);
        hipMemcpy(dm, hm.data(), 4 * hm.size(), hipMemcpyHostToDevice);
        for (int nw : {16, 8}) {
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 66 * 1024);
            hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 66 * 1024, 0, out, dv, dm, nb, 1);
            hipDeviceSynchronize();
            std::vector<double> ho((size_t)nw * 16 * 128);
            hipMemcpy(ho.data(), out, 8 * ho.size(), hipMemcpyDeviceToHost);
            double err = 0;
            for (size_t i = 0; i < ho.size(); ++i) err = fmax(err, fabs(ho[i] - ref[i]));
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(256), dim3(nw * 64), 66 * 1024, 0, out, dv, dm, nb, reps);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double entries = (double)nw * nb * 16 * reps;
            printf("%-44s waves %2d: %8.3f ms  %6.2f cyc / entry / CU   err %.1e %s\n", name, nw, ms,
                   ms * 1e-3 * 2.4e9 / entries, err, err < 1e-9 ? "OK" : "WRONG");
        }
    };
    run(kidx<1>, "one LDS read per entry (shipped batch)", 1);
    run(kidx<2>, "pairs of a row share a read (8 per batch)", 2);
    run(kidx<4>, "quads of a row share a read (4 per batch)", 4);
    return 0;
}
