// K1n  Dense self sandwich  out = X' diag(d) X  of an unrestricted block of AT MOST 11 columns (reference:
// ext/dense_helpers-tmpl.cpp:266-311; the reference's own benchmark designs `dense` (4M x 10) and
// `dense_cat` (3M x 5 dense columns), benchmark/generate_matrices.py:90-100).
//
// A block this narrow is a pure stream (80 bytes per row at 10 columns) and the MFMA syrk spends its
// time padding it to 16 columns and staging 32-row chunks through LDS (0.245 ms for 4M x 10 where
// the block streams in 0.06 ms).  Here a lane owns a ROW: it loads the row's m values (m strided
// loads of a wave cover 64 consecutive rows: every cache line is used completely), and keeps all
// P = m (m + 1) / 2 <= 66 weighted products in registers -- no LDS, no tiles, any order (C / F) and
// alignment.  (First attempt: a lane per PAIR of columns, one row per wave step -- three vector-memory
// instructions per row made it address-issue bound at 0.66 ms.)  The 64 lanes are summed with DPP
// moves once at the end; partial sums per workgroup in the workspace, summed in a fixed order by the
// finish kernel (double accumulation for f32 too).
#include <algorithm>

#include "common.hpp"

namespace tmh {

constexpr int SN_MAXM = 11;
constexpr int SN_THREADS = 256;

template <typename F, int M, bool STAGED>
__global__ __launch_bounds__(SN_THREADS) void syrk_narrow_kernel(const F *__restrict__ X, int64_t n, int order_f,
                                                                 const F *__restrict__ d, int64_t rows_per_wg,
                                                                 double *__restrict__ part) {
    constexpr int P = M * (M + 1) / 2;
    constexpr int NW = SN_THREADS / 64;
    __shared__ double red[NW][P];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t lo = (int64_t)blockIdx.x * rows_per_wg, hi = min(n, lo + rows_per_wg);
    const int64_t si = order_f ? n : 1, sr = order_f ? 1 : M;           // element strides: column / row
    double acc[P];
#pragma unroll
    for (int p = 0; p < P; ++p) acc[p] = 0.0;
    auto take = [&](int64_t row) {
        const F *xr = X + row * sr;
        double x[M];
#pragma unroll
        for (int i = 0; i < M; ++i) x[i] = (double)xr[i * si];
        const double dv = (double)d[row];
        int p = 0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const double t = dv * x[i];
#pragma unroll
            for (int j = 0; j <= i; ++j) acc[p + j] = __builtin_fma(t, x[j], acc[p + j]);
            p += i + 1;
        }
    };
    constexpr int VEC = 16 / (int)sizeof(F);
    if (STAGED) {
        // C order: a lane reading its own row of M elements touches M * sizeof(F) / 128 + 1 cache lines per
        // load instruction (3.7 TB/s for 10 doubles).  The workgroup copies SN_THREADS whole rows with flat
        // 16-byte loads instead (one tile ahead, in registers), lanes then read their row from LDS; the LDS
        // rows are padded to an odd length.
        typedef F vec_t __attribute__((ext_vector_type(VEC)));
        constexpr int MS = M | 1;
        constexpr int NV = (SN_THREADS * M + VEC * SN_THREADS - 1) / (VEC * SN_THREADS);   // vectors per lane
        __shared__ F tile[SN_THREADS * MS];
        vec_t x[NV];
        auto fetch = [&](int64_t r0) {
            const int nvec = (int)(min((int64_t)SN_THREADS, hi - r0) * M) / VEC;
            const vec_t *src = reinterpret_cast<const vec_t *>(X + r0 * M);
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int q = (int)threadIdx.x + u * SN_THREADS;
                if (q < nvec) x[u] = __builtin_nontemporal_load(src + q);
            }
        };
        if (lo < hi) fetch(lo);
        for (int64_t r0 = lo; r0 < hi; r0 += SN_THREADS) {
            const int rows_here = (int)min((int64_t)SN_THREADS, hi - r0);
            const int elems = rows_here * M, nvec = elems / VEC;
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int q = (int)threadIdx.x + u * SN_THREADS;
                if (q < nvec) {
                    const int e = q * VEC;
                    int rr = e / M, cc = e - rr * M;         // M is a compile-time constant
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        tile[rr * MS + cc] = x[u][k];
                        if (++cc == M) { cc = 0; ++rr; }
                    }
                }
            }
            if ((int)threadIdx.x < elems - nvec * VEC) {
                const int e = nvec * VEC + threadIdx.x;
                tile[(e / M) * MS + e % M] = X[r0 * M + e];
            }
            __syncthreads();
            if (r0 + SN_THREADS < hi) fetch(r0 + SN_THREADS);
            if ((int)threadIdx.x < rows_here) {
                const F *xr = tile + threadIdx.x * MS;
                double xx[M];
#pragma unroll
                for (int i = 0; i < M; ++i) xx[i] = (double)xr[i];
                const double dv = (double)d[r0 + threadIdx.x];
                int p = 0;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const double t = dv * xx[i];
#pragma unroll
                    for (int j = 0; j <= i; ++j) acc[p + j] = __builtin_fma(t, xx[j], acc[p + j]);
                    p += i + 1;
                }
            }
        }
    } else {
        int64_t row = lo + (int64_t)wave * 64 + lane;
        for (; row + (int64_t)NW * 64 < hi; row += (int64_t)2 * NW * 64) {   // two rows per turn: loads of both in flight
            take(row);
            take(row + (int64_t)NW * 64);
        }
        if (row < hi) take(row);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        double v = acc[p];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) red[wave][p] = v;
    }
    __syncthreads();
    if (threadIdx.x < P) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[w][threadIdx.x];
        part[(int64_t)blockIdx.x * 128 + threadIdx.x] = t;
    }
}

// fixed-order sum of the workgroups' partial sums: 16 strided slices, then the slices
template <typename F>
__global__ __launch_bounds__(128 * 8) void syrk_narrow_finish_kernel(const double *__restrict__ part, int nblk, int m,
                                                                     F *__restrict__ out) {
    __shared__ double red[8][128];
    const int p = threadIdx.x, sl = threadIdx.y;
    const int P = m * (m + 1) / 2;
    double t = 0.0;
    if (p < P) {
        // 8 partials in flight per lane: the plain loop was one dependent load after the other
        // (128 round trips for 1024 workgroups = 52 us for a 10 x 10 result)
        int b = sl;
        for (; b + 56 < nblk; b += 64) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = part[(int64_t)(b + 8 * q) * 128 + p];
            t += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; b < nblk; b += 8) t += part[(int64_t)b * 128 + p];
    }
    red[sl][p] = t;
    __syncthreads();
    if (sl != 0 || p >= P) return;
    t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][p];
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= p) ++i;
    const int j = p - i * (i + 1) / 2;
    out[i * m + j] = (F)t;
    out[j * m + i] = (F)t;
}

bool syrk_narrow_ok(int64_t m) { return m >= 1 && m <= SN_MAXM; }

template <typename F>
int run_syrk_narrow(const F *X, int64_t n, int64_t m, int order_f, const F *d, F *out, hipStream_t st) {
    TM_REQUIRE(syrk_narrow_ok(m), "the narrow syrk takes 1 .. 11 columns");
    if (n == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)(m * m), st));
        return TM_OK;
    }
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(4 * NUM_CU, ceil_div(n, 2048)));
    // staged through LDS (C order, 16-byte aligned, 3 or more columns); a workgroup's rows then start on a
    // multiple of 4 rows = a 16-byte boundary
    // (where it measured faster, scripts/dev/time_syrk_narrow.py: 8 columns 0.120 -> 0.088 ms f64, 0.157 -> 0.081 f32;
    // from 9 columns on the accumulators leave the staged form two waves per SIMD and the direct loads win)
    const bool pays = sizeof(F) == 8 ? (m >= 4 && m <= 8) : (m == 3 || m == 4 || m == 8);
    const int64_t knob = tune("syrk_narrow_staged", -1);               // 0 / 1 force, -1 = by width
    const bool staged = !order_f && m >= 3 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && (knob < 0 ? pays : knob != 0);
    const int64_t rpw = ceil_div(ceil_div(n, want), 4) * 4;
    const int nblk = (int)ceil_div(n, rpw);
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(double) * (size_t)nblk * 128 + 256, &wsv, st);
    if (rc) return rc;
    double *part = reinterpret_cast<double *>(wsv);
    prof_begin(st);
    auto go = [&](auto mc) {
        constexpr int M = decltype(mc)::value;
        if (staged)
            hipLaunchKernelGGL((syrk_narrow_kernel<F, M, true>), dim3((unsigned)nblk), dim3(SN_THREADS), 0, st, X, n,
                               order_f, d, rpw, part);
        else
            hipLaunchKernelGGL((syrk_narrow_kernel<F, M, false>), dim3((unsigned)nblk), dim3(SN_THREADS), 0, st, X, n,
                               order_f, d, rpw, part);
    };
    switch ((int)m) {
        case 1: go(std::integral_constant<int, 1>{}); break;
        case 2: go(std::integral_constant<int, 2>{}); break;
        case 3: go(std::integral_constant<int, 3>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 5: go(std::integral_constant<int, 5>{}); break;
        case 6: go(std::integral_constant<int, 6>{}); break;
        case 7: go(std::integral_constant<int, 7>{}); break;
        case 8: go(std::integral_constant<int, 8>{}); break;
        case 9: go(std::integral_constant<int, 9>{}); break;
        case 10: go(std::integral_constant<int, 10>{}); break;
        default: go(std::integral_constant<int, 11>{}); break;
    }
    prof_end(st);
    TM_LAUNCH_CHECK();
    hipLaunchKernelGGL((syrk_narrow_finish_kernel<F>), dim3(1), dim3(128, 8), 0, st, part, nblk, (int)m, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template int run_syrk_narrow<float>(const float *, int64_t, int64_t, int, const float *, float *, hipStream_t);
template int run_syrk_narrow<double>(const double *, int64_t, int64_t, int, const double *, double *, hipStream_t);

}  // namespace tmh
