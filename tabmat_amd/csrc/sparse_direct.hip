// K2 for WIDE, VERY SPARSE blocks: out = A^T diag(d) A with one global atomic per pair
// (reference: ext/sparse.pyx:17-77 sparse_sandwich).
//
// The tiled kernels (sparse.hip) pay per (row, 128 x 128 tile): fine while a row has nonzeros in
// most chunks, quadratic in the width when it has fewer than one per chunk (4096 columns at 0.2 %:
// 528 tiles, 11.6 ms for 2M rows and 75M pairs).  Here the cost follows the PAIRS: a 16-lane group
// takes one row, holds 16 of its entries one per lane and meets every other entry by rotating the
// B side through the group with DPP (row_ror), one global_atomic_add_f64 per pair into a double
// m x m accumulator (lower triangle, L2-resident: ~20 G atomics/s); a second kernel mirrors and
// converts.  Chosen by the host when the pair count says it is cheaper than the tile passes.
#include "common.hpp"

namespace tmh {

template <int S>
__device__ __forceinline__ int dpp_ror16_i32(int v) {
    if constexpr (S == 0) return v;
    else return __builtin_amdgcn_mov_dpp(v, 0x120 + S, 0xF, 0xF, true);       // row_ror:S
}
template <int S>
__device__ __forceinline__ double dpp_ror16(double v) {
    return __hiloint2double(dpp_ror16_i32<S>(__double2hiint(v)), dpp_ror16_i32<S>(__double2loint(v)));
}

template <typename F>
__global__ __launch_bounds__(256) void sparse_sandwich_direct_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ d, int64_t n, int64_t m, double *__restrict__ acc) {
    const int lane = threadIdx.x & 63;
    const int t = lane & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    // all four 16-lane groups of a wave run the same number of iterations (uniform DPP code)
    const int64_t nstep = (n + ngrp - 1) / ngrp;
    for (int64_t it = 0; it < nstep; ++it) {
        const int64_t k = it * ngrp + grp;
        int64_t p0 = 0, p1 = 0;
        double dk = 0.0;
        if (k < n) {
            dk = (double)d[k];
            if (dk != 0.0) {
                p0 = ptr[k];
                p1 = ptr[k + 1];
            }
        }
        const int len = (int)(p1 - p0);
        int maxlen = len;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));
        for (int a0 = 0; a0 < maxlen; a0 += 16) {
            const bool aok = a0 + t < len;
            const int ca = aok ? ind[p0 + a0 + t] : -1;
            const double va = aok ? (double)data[p0 + a0 + t] * dk : 0.0;
            for (int b0 = 0; b0 <= a0; b0 += 16) {
                const bool bok = b0 + t < len;
                const int cb = bok ? ind[p0 + b0 + t] : 0x7fffffff;
                const double vb = bok ? (double)data[p0 + b0 + t] : 0.0;
                const int live = min(maxlen - b0, 16);         // rotations that can meet an entry
                static_for<16>([&](auto sc) {
                    constexpr int S = decltype(sc)::value;
                    // rotation S pairs lane t with the B entry of lane (t + S) or (t - S) mod 16:
                    // all 16 rotations are needed unless the block is short on BOTH sides
                    if (S < live || S > 16 - live || a0 != b0) {
                        const int cbs = dpp_ror16_i32<S>(cb);
                        const double vbs = dpp_ror16<S>(vb);
                        if (ca >= 0 && cbs <= ca)              // lower triangle; padding has cb = INT_MAX
                            atomicAdd(acc + (int64_t)ca * m + cbs, va * vbs);
                    }
                });
            }
        }
    }
}

// out[i][j] = out[j][i] = acc[max][min]
template <typename F>
__global__ void sparse_direct_finish_kernel(const double *__restrict__ acc, int64_t m,
                                            F *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * m) return;
    const int64_t i = e / m, j = e % m;
    out[e] = (F)acc[max(i, j) * m + min(i, j)];
}

template <typename F>
static int run_sparse_sandwich_direct(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n,
                                      int64_t m, const F *d, F *out, hipStream_t st) {
    if (m == 0) return TM_OK;
    TM_REQUIRE(m < (1 << 20), "too many columns");
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(double) * (size_t)(m * m) + 256, &wsv, st);
    if (rc) return rc;
    double *acc = reinterpret_cast<double *>(wsv);
    TM_HIP(hipMemsetAsync(acc, 0, sizeof(double) * (size_t)(m * m), st));
    if (n > 0) {
        const int64_t nblk = std::min<int64_t>(ceil_div(n, 16), (int64_t)NUM_CU * 8);
        prof_begin(st);
        hipLaunchKernelGGL((sparse_sandwich_direct_kernel<F>), dim3((unsigned)nblk), dim3(256), 0, st,
                           data, ind, ptr, d, n, m, acc);
        prof_end(st);
        TM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((sparse_direct_finish_kernel<F>), dim3((unsigned)ceil_div(m * m, 256)), dim3(256),
                       0, st, acc, m, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {
int tm_sparse_sandwich_direct_f32(const float *csr_data, const int32_t *csr_indices,
                                  const int64_t *csr_indptr, int64_t n, int64_t m, const float *d,
                                  float *out, void *stream) {
    return tmh::run_sparse_sandwich_direct<float>(csr_data, csr_indices, csr_indptr, n, m, d, out,
                                                  tmh::as_stream(stream));
}
int tm_sparse_sandwich_direct_f64(const double *csr_data, const int32_t *csr_indices,
                                  const int64_t *csr_indptr, int64_t n, int64_t m, const double *d,
                                  double *out, void *stream) {
    return tmh::run_sparse_sandwich_direct<double>(csr_data, csr_indices, csr_indptr, n, m, d, out,
                                                   tmh::as_stream(stream));
}
}  // extern "C"
