// Categorical-block kernels for gfx950: the weighted-histogram family.
//
//   K4a/K4b  transpose_matvec / sandwich diagonal : res[col(k)]            += v[k]
//   K4c      cat x cat cross sandwich             : res[col_i(k),col_j(k)] += d[k]
//   K4d      cat x dense cross sandwich           : res[col(k), :]         += d[k] * M[k, :]
//   cat x sparse cross sandwich                   : res[col(k), :]         += d[k] * S[k, :]
//   K4e      matvec (gather)                      : out[k]                 += v[col(k)]
//
// Reference loops: ext/cat_split_helpers-tmpl.cpp:4-151, ext/categorical.pyx:23-218,
// categorical_matrix.py:825-838.  The reference privatises the output per OpenMP thread and
// merges; here the output (or a category-range slice of it that fits the CU's 160 KB LDS) is
// privatised per workgroup in LDS, updated with ds_add (wavefront atomics), written to the
// workspace and combined by reduce_partials_kernel.  Where the output does not fit LDS it is
// split by CATEGORY RANGE across blockIdx.y ("parts"): every part scans the 4-byte codes of
// its rows but touches the wide operand (dense row / sparse row) only for rows whose category
// falls in its range, so the wide operand is still read exactly once from HBM.
#include <algorithm>

#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

__global__ void fill_i32_kernel(int32_t *p, int64_t n, int32_t v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void scatter_iota_i32_kernel(int32_t *map, const int32_t *cols, int64_t n_cols) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_cols) map[cols[i]] = (int32_t)i;
}

int build_col_map(int32_t *map, int64_t m, const int32_t *cols, int64_t n_cols, hipStream_t st) {
    if (m > 0) {
        hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, map,
                           m, -1);
        TM_LAUNCH_CHECK();
    }
    if (n_cols > 0) {
        hipLaunchKernelGGL(scatter_iota_i32_kernel, dim3((unsigned)ceil_div(n_cols, 256)),
                           dim3(256), 0, st, map, cols, n_cols);
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

// ---------------------------------------------------------------------------------------
// 1-D / 2-D weighted histogram with LDS-privatised bins.
//   TWO = false: bin = col(codes_i[k])                       (optionally masked by col_map)
//   TWO = true : bin = (col_i(k) - i0) * j_ncol + col_j(k)   for col_i in [i0, i0 + ti)
// Each block owns rows [blockIdx.x * rows_per_block, ...) of the (possibly restricted) row
// list and the category range of part blockIdx.y.  LDS bins are flushed to
// ws[part][blockIdx.x][stride].
// ---------------------------------------------------------------------------------------
template <typename F, bool TWO>
__global__ __launch_bounds__(1024) void hist_lds_kernel(
    const int32_t *__restrict__ ci, const int32_t *__restrict__ cj, const F *__restrict__ w,
    const int32_t *__restrict__ rows, int64_t n_iter, int64_t rows_per_block, int drop_i,
    int drop_j, int i_ncol, int j_ncol, int ti, const int32_t *__restrict__ col_map,
    F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *bins = reinterpret_cast<lds_acc_t *>(smem_raw);      // doubles (common.hpp)
    const int part = blockIdx.y;
    const int i0 = part * ti;
    const int i1 = min(i0 + ti, i_ncol);
    const int nbins = TWO ? (i1 - i0) * j_ncol : (i1 - i0);
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) bins[b] = 0.0;
    __syncthreads();

    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);
    int64_t tbeg = t0;
    if (!TWO && rows == nullptr && (t0 & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(ci) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
        // streaming fast path (BASELINE configs[2]): 4 consecutive rows per thread and step,
        // one 16-byte load of codes + 4 weights in flight per lane
        typedef int i4 __attribute__((ext_vector_type(4)));
        const int64_t nvec = (t1 - t0) / 4;
        for (int64_t q = threadIdx.x; q < nvec; q += blockDim.x) {
            const int64_t k = t0 + q * 4;
            const i4 cc = *reinterpret_cast<const i4 *>(ci + k);
            F ww[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) ww[e] = w[k + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = cc[e] - drop_i;
                if (c >= i0 && c < i1 && !(col_map && col_map[c] < 0)) atomic_add(&bins[c - i0], (lds_acc_t)ww[e]);
            }
        }
        tbeg = t0 + nvec * 4;
    }
    for (int64_t t = tbeg + threadIdx.x; t < t1; t += blockDim.x) {
        const int64_t k = rows ? (int64_t)rows[t] : t;
        const int c = ci[k] - drop_i;
        if (c < i0 || c >= i1) continue;
        if (TWO) {
            const int c2 = cj[k] - drop_j;
            if (c2 < 0) continue;
            atomic_add(&bins[(c - i0) * j_ncol + c2], (lds_acc_t)w[k]);
        } else {
            if (col_map && col_map[c] < 0) continue;
            atomic_add(&bins[c - i0], (lds_acc_t)w[k]);
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)part * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) dst[b] = (F)bins[b];
}

// Fallback when one category row of bins does not fit LDS: global atomics into `out`.
template <typename F, bool TWO>
__global__ __launch_bounds__(256) void hist_global_kernel(
    const int32_t *__restrict__ ci, const int32_t *__restrict__ cj, const F *__restrict__ w,
    const int32_t *__restrict__ rows, int64_t n_iter, int drop_i, int drop_j, int j_ncol,
    const int32_t *__restrict__ col_map, F *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_iter; t += stride) {
        const int64_t k = rows ? (int64_t)rows[t] : t;
        const int c = ci[k] - drop_i;
        if (c < 0) continue;
        if (TWO) {
            const int c2 = cj[k] - drop_j;
            if (c2 < 0) continue;
            atomic_add(&out[(int64_t)c * j_ncol + c2], w[k]);
        } else {
            if (col_map && col_map[c] < 0) continue;
            atomic_add(&out[c], w[k]);
        }
    }
}

// 2-D weighted histogram over rows GROUPED BY the level of categorical i (static per pair of categoricals: perm =
// the rows in level order, ci_s / cj_s = the two column indices in that order, lptr = first position of every
// level).  Part p owns the levels [p * ti, (p + 1) * ti) -- one LDS tile of the table -- and reads ONLY the
// positions of those levels: one pass over the rows whatever the number of parts, against n_parts passes for
// hist_lds_kernel<F, true> and one device-scope atomic per row (23 G/s) for hist_global_kernel.  Per row: 12
// bytes of coalesced loads, one gather of the weight, one ds_add_f64.
template <typename F>
__global__ __launch_bounds__(1024) void hist_sorted_kernel(
    const int32_t *__restrict__ ci_s, const int32_t *__restrict__ cj_s, const int32_t *__restrict__ perm,
    const int64_t *__restrict__ lptr, const F *__restrict__ w, int i_ncol, int j_ncol, int ti,
    F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *bins = reinterpret_cast<lds_acc_t *>(smem_raw);
    const int part = blockIdx.y;
    const int i0 = part * ti;
    const int i1 = min(i0 + ti, i_ncol);
    const int nbins = (i1 - i0) * j_ncol;
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) bins[b] = 0.0;
    __syncthreads();
    const int64_t p0 = lptr[i0], p1 = lptr[i1];
    const int64_t per = (p1 - p0 + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = p0 + (int64_t)blockIdx.x * per, t1 = min(t0 + per, p1);
    const int B = blockDim.x;
    int64_t t = t0 + threadIdx.x;
    for (; t + 3 * (int64_t)B < t1; t += 4 * (int64_t)B) {          // four positions per lane in flight
        int c[4], c2[4];
        F x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            c[u] = ci_s[t + u * B];
            c2[u] = cj_s[t + u * B];
            x[u] = w[perm[t + u * B]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (c2[u] >= 0) atomic_add(&bins[(c[u] - i0) * j_ncol + c2[u]], (lds_acc_t)x[u]);
    }
    for (; t < t1; t += B) {
        const int c2 = cj_s[t];
        if (c2 >= 0) atomic_add(&bins[(ci_s[t] - i0) * j_ncol + c2], (lds_acc_t)w[perm[t]]);
    }
    __syncthreads();
    F *dst = ws + ((int64_t)part * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) dst[b] = (F)bins[b];
}

constexpr size_t HIST_LDS_MAX = 128 * 1024;

// Generic driver.  For !TWO: out[i_ncol] (accumulate or overwrite).  For TWO: out[i_ncol*j_ncol].
template <typename F, bool TWO>
static int run_hist(const int32_t *ci, const int32_t *cj, const F *w, const int32_t *rows,
                    int64_t n_iter, int drop_i, int drop_j, int64_t i_ncol, int64_t j_ncol,
                    const int32_t *col_map, F *out, bool accumulate, hipStream_t st, int parts_max = 64) {
    const int64_t total = TWO ? i_ncol * j_ncol : i_ncol;
    if (total == 0) return TM_OK;
    if (!accumulate) TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
    if (n_iter == 0) return TM_OK;
    const size_t row_bytes = sizeof(lds_acc_t) * (size_t)(TWO ? j_ncol : 1);
    if (row_bytes > HIST_LDS_MAX || i_ncol > (int64_t)INT32_MAX / (TWO ? j_ncol : 1)) {
        const int64_t nblk = std::min<int64_t>(ceil_div(n_iter, 256 * 4), 2048);
        hipLaunchKernelGGL((hist_global_kernel<F, TWO>), dim3((unsigned)nblk), dim3(256), 0, st, ci,
                           cj, w, rows, n_iter, drop_i, drop_j, (int)j_ncol, col_map, out);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    int64_t ti = std::min<int64_t>(i_ncol, (int64_t)(HIST_LDS_MAX / row_bytes));
    const int64_t n_parts = ceil_div(i_ncol, ti);
    if (n_parts > parts_max) {  // too many passes over the codes: global atomics instead
        const int64_t nblk = std::min<int64_t>(ceil_div(n_iter, 256 * 4), 2048);
        hipLaunchKernelGGL((hist_global_kernel<F, TWO>), dim3((unsigned)nblk), dim3(256), 0, st, ci,
                           cj, w, rows, n_iter, drop_i, drop_j, (int)j_ncol, col_map, out);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    ti = ceil_div(i_ncol, n_parts);  // balance the parts
    const int64_t stride = ti * (TWO ? j_ncol : 1);
    const size_t lds = (size_t)stride * sizeof(lds_acc_t);
    const int threads = lds > 64 * 1024 ? 1024 : 512;
    const int blocks_per_cu = lds > 64 * 1024 ? 1 : 2;
    int64_t nblk = std::max<int64_t>(1, (NUM_CU * blocks_per_cu) / n_parts);
    nblk = std::min<int64_t>(nblk, ceil_div(n_iter, threads * 4));
    const int64_t rows_per_block = ceil_div(ceil_div(n_iter, nblk), 4) * 4;   // keeps t0 % 4 == 0
    nblk = ceil_div(n_iter, rows_per_block);
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(F) * (size_t)(n_parts * nblk * stride) + 256, &wsv, st);
    if (rc) return rc;
    F *ws = reinterpret_cast<F *>(wsv);
    if (lds > 48 * 1024) {
        TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_lds_kernel<F, TWO>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    prof_begin(st);
    hipLaunchKernelGGL((hist_lds_kernel<F, TWO>), dim3((unsigned)nblk, (unsigned)n_parts),
                       dim3(threads), lds, st, ci, cj, w, rows, n_iter, rows_per_block, drop_i,
                       drop_j, (int)i_ncol, (int)j_ncol, (int)ti, col_map, ws, stride);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return launch_reduce_partials<F>(ws, stride, (int)nblk, (int)n_parts, out, total, accumulate,
                                     st);
}

template <typename F>
static int run_hist_sorted(const int32_t *ci_s, const int32_t *cj_s, const int32_t *perm, const int64_t *lptr,
                           int64_t n_sorted, const F *w, int64_t i_ncol, int64_t j_ncol, F *out, hipStream_t st) {
    const int64_t total = i_ncol * j_ncol;
    if (total == 0) return TM_OK;
    const size_t row_bytes = sizeof(lds_acc_t) * (size_t)j_ncol;
    TM_REQUIRE(row_bytes <= HIST_LDS_MAX && total < (int64_t)INT32_MAX,
               "cat x cat (level-sorted): one row of the table must fit the LDS tile");
    if (n_sorted == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    int64_t ti = std::min<int64_t>(i_ncol, (int64_t)(HIST_LDS_MAX / row_bytes));
    const int64_t n_parts = ceil_div(i_ncol, ti);
    ti = ceil_div(i_ncol, n_parts);
    const int64_t stride = ti * j_ncol;
    const size_t lds = (size_t)stride * sizeof(lds_acc_t);
    const int threads = lds > 64 * 1024 ? 1024 : 512;
    const int per_cu = lds > 64 * 1024 ? 1 : 2;
    // workgroups per part: two rounds over the chip, at least ~4096 positions each
    int64_t nblk = std::max<int64_t>(1, ceil_div(2 * NUM_CU * per_cu, n_parts));
    nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, n_sorted / n_parts / 4096));
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(F) * (size_t)(n_parts * nblk * stride) + 256, &wsv, st);
    if (rc) return rc;
    F *ws = reinterpret_cast<F *>(wsv);
    if (lds > 48 * 1024)
        TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hist_sorted_kernel<F>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL((hist_sorted_kernel<F>), dim3((unsigned)nblk, (unsigned)n_parts), dim3(threads), lds, st,
                       ci_s, cj_s, perm, lptr, w, (int)i_ncol, (int)j_ncol, (int)ti, ws, stride);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return launch_reduce_partials<F>(ws, stride, (int)nblk, (int)n_parts, out, total, false, st);
}

// ---------------------------------------------------------------------------------------
// K4e gather
// ---------------------------------------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(256) void cat_matvec_kernel(const int32_t *__restrict__ codes,
                                                         int64_t n, int drop_first,
                                                         const F *__restrict__ v,
                                                         const int32_t *__restrict__ col_map,
                                                         F *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int c = codes[i] - drop_first;
        if (c >= 0 && (!col_map || col_map[c] >= 0)) out[i] += v[c];
    }
}

// The same with four rows per lane and two quads in flight (16-byte loads of the codes, 16-byte accesses of out;
// codes and out 16-byte aligned): 50M rows x 10k levels took 0.49 ms lane by lane (a 4-byte load, a gather, an
// 8-byte read and an 8-byte write per lane and turn).  ASSIGN: out is fresh storage and is WRITTEN (out[i] = v[col]
// or 0) -- no zero fill before the launch, no read of out.
// LDSV: the coefficient vector (with the column selection folded in as zeros) is staged in LDS first -- 10k levels
// are 80 KB, more than the L1 holds, and 50M random 8-byte reads of it through the L1 cost 0.26 ms against 0.12 for
// the codes and the output together.
template <typename F, bool ASSIGN, bool LDSV>
__global__ __launch_bounds__(512) void cat_matvec_quad_kernel(const int32_t *__restrict__ codes, int64_t n,
                                                              int n_cols, int drop_first,
                                                              const F *__restrict__ v,
                                                              const int32_t *__restrict__ col_map,
                                                              F *__restrict__ out) {
    typedef int32_t i4 __attribute__((ext_vector_type(4)));
    constexpr int OV = 16 / (int)sizeof(F);            // out elements per 16-byte access
    typedef F ov_t __attribute__((ext_vector_type(OV)));
    extern __shared__ __attribute__((aligned(16))) unsigned char cat_mv_smem[];
    F *vl = reinterpret_cast<F *>(cat_mv_smem);
    if (LDSV) {
        for (int c = threadIdx.x; c < n_cols; c += blockDim.x) vl[c] = (!col_map || col_map[c] >= 0) ? v[c] : F(0);
        __syncthreads();
    }
    const int64_t nq = n >> 2;                          // whole quads
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto term = [&](int code) -> F {
        const int c = code - drop_first;
        if (LDSV) return c >= 0 ? vl[c] : F(0);
        return (c >= 0 && (!col_map || col_map[c] >= 0)) ? v[c] : F(0);
    };
    auto put = [&](int64_t q, const i4 cq) {
        F t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = term(cq[k]);
        ov_t *o = reinterpret_cast<ov_t *>(out + 4 * q);
#pragma unroll
        for (int h = 0; h < 4 / OV; ++h) {
            ov_t w;
            if (ASSIGN) {
#pragma unroll
                for (int e = 0; e < OV; ++e) w[e] = t[h * OV + e];
            } else {
                w = o[h];
#pragma unroll
                for (int e = 0; e < OV; ++e) w[e] += t[h * OV + e];
            }
            o[h] = w;
        }
    };
    const i4 *cq = reinterpret_cast<const i4 *>(codes);
    int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; q + stride < nq; q += 2 * stride) {
        const i4 a = __builtin_nontemporal_load(cq + q), b = __builtin_nontemporal_load(cq + q + stride);
        put(q, a);
        put(q + stride, b);
    }
    if (q < nq) put(q, __builtin_nontemporal_load(cq + q));
    // the n % 4 last rows
    const int64_t i = 4 * nq + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x);
    if (i < n) out[i] = ASSIGN ? term(codes[i]) : out[i] + term(codes[i]);
}

// ---------------------------------------------------------------------------------------
// K4d  cat x dense:   tile[col(k) - i0][jc] += d[k] * M[k, j_cols[jc]]
// Workgroup = 256 threads; LDS tile = ti x n_j.  C-ordered M: a wave scans 64 codes at a time,
// ballots the rows that fall in this part's category range and then streams each selected row
// of M with all 64 lanes (coalesced 512 B / 1 KB segments).  F-ordered M: lane <-> row, loop
// over columns (coalesced along the rows of one column).
// ---------------------------------------------------------------------------------------
template <typename F, bool ORDER_F>
__global__ __launch_bounds__(256) void cat_dense_kernel(
    const int32_t *__restrict__ codes, const F *__restrict__ d, const int32_t *__restrict__ rows,
    int64_t n_iter, int64_t rows_per_block, int drop_first, int i_ncol, int ti,
    const F *__restrict__ M, int64_t M_nrow, int64_t M_ncol, const int32_t *__restrict__ j_cols,
    int n_j, F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);
    const int part = blockIdx.y;
    const int i0 = part * ti;
    const int i1 = min(i0 + ti, i_ncol);
    const int nel = (i1 - i0) * n_j;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwave = blockDim.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);

    for (int64_t base = t0 + (int64_t)wave * 64; base < t1; base += (int64_t)nwave * 64) {
        const int64_t t = base + lane;
        int64_t k = 0;
        int c = -1;
        F dk = F(0);
        if (t < t1) {
            k = rows ? (int64_t)rows[t] : t;
            c = codes[k] - drop_first;
            if (c >= i0 && c < i1) dk = d[k];
            else c = -1;
        }
        if (ORDER_F) {
            if (c >= 0) {
                lds_acc_t *trow = tile + (c - i0) * n_j;
                for (int jc = 0; jc < n_j; ++jc) {
                    const int64_t j = j_cols ? (int64_t)j_cols[jc] : jc;
                    atomic_add(&trow[jc], (lds_acc_t)(dk * M[j * M_nrow + k]));
                }
            }
        } else {
            unsigned long long mask = __ballot(c >= 0);
            while (mask) {
                const int src = __builtin_ctzll(mask);
                mask &= mask - 1;
                const int64_t kk = __shfl(k, src, 64);
                const int cc = __shfl(c, src, 64);
                const F dd = __shfl(dk, src, 64);
                const F *mrow = M + kk * M_ncol;
                lds_acc_t *trow = tile + (cc - i0) * n_j;
                for (int jc = lane; jc < n_j; jc += 64) {
                    const int64_t j = j_cols ? (int64_t)j_cols[jc] : jc;
                    atomic_add(&trow[jc], (lds_acc_t)(dd * mrow[j]));
                }
            }
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)part * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) dst[b] = (F)tile[b];
}

// ---------------------------------------------------------------------------------------
// cat x sparse:   tile[col(k) - i0][col_map[j]] += d[k] * S[k, j]   over the CSR row of k.
// G lanes cooperate on one sparse row (G = 64 / rows-per-wave-step, chosen from nnz/row).
// ---------------------------------------------------------------------------------------
template <typename F, int G>
__global__ __launch_bounds__(256) void cat_sparse_kernel(
    const int32_t *__restrict__ codes, const F *__restrict__ d, const int32_t *__restrict__ rows,
    int64_t n_iter, int64_t rows_per_block, int drop_first, int i_ncol, int ti,
    const F *__restrict__ sdata, const int32_t *__restrict__ sind,
    const int64_t *__restrict__ sptr, const int32_t *__restrict__ col_map, int n_out_cols,
    F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);
    const int part = blockIdx.y;
    const int i0 = part * ti;
    const int i1 = min(i0 + ti, i_ncol);
    const int nel = (i1 - i0) * n_out_cols;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();

    constexpr int RPS = 64 / G;  // rows per wave step
    const int lane = threadIdx.x & 63;
    const int sub = lane / G;    // which row of the step
    const int sl = lane % G;     // lane within the row group
    const int wave = threadIdx.x >> 6;
    const int nwave = blockDim.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);

    for (int64_t base = t0 + (int64_t)wave * 64; base < t1; base += (int64_t)nwave * 64) {
        // scan 64 codes, keep the rows of this part
        const int64_t t = base + lane;
        int64_t k = 0;
        int c = -1;
        F dk = F(0);
        if (t < t1) {
            k = rows ? (int64_t)rows[t] : t;
            c = codes[k] - drop_first;
            if (c >= i0 && c < i1) dk = d[k];
            else c = -1;
        }
        unsigned long long mask = __ballot(c >= 0);
        while (mask) {
            // each G-lane group takes the next selected row
            int src = -1;
            unsigned long long m2 = mask;
#pragma unroll
            for (int s = 0; s < RPS; ++s) {
                if (m2) {
                    const int b = __builtin_ctzll(m2);
                    m2 &= m2 - 1;
                    if (s == sub) src = b;
                }
            }
            mask = m2;
            const int srcl = src < 0 ? 0 : src;
            const int64_t kk = __shfl(k, srcl, 64);
            const int cc = __shfl(c, srcl, 64);
            const F dd = __shfl(dk, srcl, 64);
            if (src >= 0) {
                lds_acc_t *trow = tile + (cc - i0) * n_out_cols;
                const int64_t p1 = sptr[kk + 1];
                for (int64_t p = sptr[kk] + sl; p < p1; p += G) {
                    const int j = sind[p];
                    const int oc = col_map ? col_map[j] : j;
                    if (oc >= 0) atomic_add(&trow[oc], (lds_acc_t)(dd * sdata[p]));
                }
            }
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)part * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) dst[b] = (F)tile[b];
}

// common launch geometry for the LDS-tile kernels: returns parts / blocks / strides
struct TilePlan {
    int64_t ti, n_parts, nblk, rows_per_block, stride;
    size_t lds;
};

static bool plan_tile(int64_t i_ncol, int64_t row_elems, size_t elem_bytes, int64_t n_iter,
                      int64_t min_rows_per_block, TilePlan *p) {
    const size_t row_bytes = elem_bytes * (size_t)row_elems;
    if (row_bytes == 0 || row_bytes > HIST_LDS_MAX) return false;
    int64_t ti = std::min<int64_t>(i_ncol, (int64_t)(HIST_LDS_MAX / row_bytes));
    p->n_parts = ceil_div(i_ncol, ti);
    p->ti = ceil_div(i_ncol, p->n_parts);
    p->stride = p->ti * row_elems;
    p->lds = (size_t)p->stride * elem_bytes;
    const int blocks_per_cu = p->lds > 64 * 1024 ? 1 : 2;
    int64_t nblk = std::max<int64_t>(1, (NUM_CU * blocks_per_cu) / p->n_parts);
    nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, ceil_div(n_iter, min_rows_per_block)));
    p->rows_per_block = ceil_div(n_iter, nblk);
    p->nblk = ceil_div(n_iter, p->rows_per_block);
    return true;
}

template <typename F>
static int run_cat_dense(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                         const F *d, const int32_t *rows, int64_t n_rows, const F *M,
                         int64_t M_ncol, int order_f, const int32_t *j_cols, int64_t n_j, F *out,
                         hipStream_t st) {
    const int64_t total = i_ncol * n_j;
    if (total == 0) return TM_OK;
    TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
    const int64_t n_iter = rows ? n_rows : n;
    if (n_iter == 0) return TM_OK;
    TilePlan p;
    if (!plan_tile(i_ncol, n_j, sizeof(lds_acc_t), n_iter, 1024, &p)) {
        set_error("cat_dense_sandwich: %lld selected dense columns exceed the LDS tile",
                  (long long)n_j);
        return TM_EUNSUPPORTED;
    }
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(F) * (size_t)(p.n_parts * p.nblk * p.stride) + 256, &wsv, st);
    if (rc) return rc;
    F *ws = reinterpret_cast<F *>(wsv);
    auto kern = order_f ? &cat_dense_kernel<F, true> : &cat_dense_kernel<F, false>;
    if (p.lds > 48 * 1024)
        TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.nblk, (unsigned)p.n_parts), dim3(256), p.lds, st,
                       codes, d, rows, n_iter, p.rows_per_block, drop_first, (int)i_ncol,
                       (int)p.ti, M, n, M_ncol, j_cols, (int)n_j, ws, p.stride);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return launch_reduce_partials<F>(ws, p.stride, (int)p.nblk, (int)p.n_parts, out, total, false,
                                     st);
}

template <typename F>
static int run_cat_sparse(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                          const F *sdata, const int32_t *sind, const int64_t *sptr, int64_t s_ncol,
                          const F *d, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                          int64_t n_cols, F *out, hipStream_t st) {
    const int64_t n_out = cols ? n_cols : s_ncol;
    const int64_t total = i_ncol * n_out;
    if (total == 0) return TM_OK;
    TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
    const int64_t n_iter = rows ? n_rows : n;
    if (n_iter == 0) return TM_OK;
    TilePlan p;
    if (!plan_tile(i_ncol, n_out, sizeof(lds_acc_t), n_iter, 1024, &p)) {
        set_error("cat_sparse_sandwich: %lld selected sparse columns exceed the LDS tile",
                  (long long)n_out);
        return TM_EUNSUPPORTED;
    }
    const size_t map_bytes = cols ? ((sizeof(int32_t) * (size_t)s_ncol + 255) / 256) * 256 : 0;
    void *wsv = nullptr;
    int rc = get_workspace(map_bytes + sizeof(F) * (size_t)(p.n_parts * p.nblk * p.stride) + 256,
                           &wsv, st);
    if (rc) return rc;
    int32_t *col_map = nullptr;
    if (cols) {
        col_map = reinterpret_cast<int32_t *>(wsv);
        rc = build_col_map(col_map, s_ncol, cols, n_cols, st);
        if (rc) return rc;
    }
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + map_bytes);
    auto kern = &cat_sparse_kernel<F, 32>;
    if (p.lds > 48 * 1024)
        TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)p.nblk, (unsigned)p.n_parts), dim3(256), p.lds, st,
                       codes, d, rows, n_iter, p.rows_per_block, drop_first, (int)i_ncol,
                       (int)p.ti, sdata, sind, sptr, col_map, (int)n_out, ws, p.stride);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return launch_reduce_partials<F>(ws, p.stride, (int)p.nblk, (int)p.n_parts, out, total, false,
                                     st);
}

template <typename F>
static int run_cat_tmv(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first, const F *v,
                       const int32_t *rows, int64_t n_rows, const int32_t *cols,
                       int64_t n_cols_sel, F *out, hipStream_t st) {
    const int64_t n_iter = rows ? n_rows : n;
    if (n_cols == 0 || n_iter == 0) return TM_OK;
    if (cols && n_cols_sel == 0) return TM_OK;
    int32_t *col_map = nullptr;
    if (cols) {
        // the col map lives at the END of the workspace so run_hist's partials (which start
        // at offset 0) cannot overlap it: reserve both up front.
        void *wsv = nullptr;
        const size_t map_bytes = ((sizeof(int32_t) * (size_t)n_cols + 255) / 256) * 256;
        const size_t part_bytes = sizeof(F) * (size_t)(2 * NUM_CU) * (size_t)n_cols + 4096;
        int rc = get_workspace(part_bytes + map_bytes, &wsv, st);
        if (rc) return rc;
        col_map = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(wsv) + part_bytes);
        rc = build_col_map(col_map, n_cols, cols, n_cols_sel, st);
        if (rc) return rc;
    }
    return run_hist<F, false>(codes, nullptr, v, rows, n_iter, drop_first, 0, n_cols, 1, col_map,
                              out, true, st);
}

template <typename F>
static int run_cat_matvec(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                          const F *v, const int32_t *cols, int64_t n_cols_sel, F *out,
                          hipStream_t st, bool assign = false) {
    if (n == 0) return TM_OK;
    if (n_cols == 0 || (cols && n_cols_sel == 0)) {
        if (assign) TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)n, st));
        return TM_OK;
    }
    int32_t *col_map = nullptr;
    if (cols) {
        void *wsv = nullptr;
        int rc = get_workspace(sizeof(int32_t) * (size_t)n_cols + 256, &wsv, st);
        if (rc) return rc;
        col_map = reinterpret_cast<int32_t *>(wsv);
        rc = build_col_map(col_map, n_cols, cols, n_cols_sel, st);
        if (rc) return rc;
    }
    const int64_t nblk = std::min<int64_t>(ceil_div(n, 256 * 4), NUM_CU * 8);
    const bool quads = ((reinterpret_cast<uintptr_t>(codes) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    // the coefficient vector in LDS when it outgrows the L1 and the rows pay for staging it in every workgroup
    const size_t vbytes = sizeof(F) * (size_t)n_cols;
    const bool ldsv = quads && vbytes > 16 * 1024 && vbytes <= 144 * 1024 && n >= 64 * n_cols;
    prof_begin(st);
    if (ldsv) {
        const int per_cu = vbytes <= 72 * 1024 ? 2 : 1;
        const int64_t nb = std::min<int64_t>(ceil_div(n, 512 * 8), NUM_CU * per_cu);
        auto go = [&](auto kern) -> int {
            if (vbytes > 48 * 1024)
                TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)vbytes));
            hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(512), vbytes, st, codes, n, (int)n_cols, drop_first, v,
                               col_map, out);
            return TM_OK;
        };
        const int rc = assign ? go(&cat_matvec_quad_kernel<F, true, true>) : go(&cat_matvec_quad_kernel<F, false, true>);
        if (rc) return rc;
    } else if (quads && assign)
        hipLaunchKernelGGL((cat_matvec_quad_kernel<F, true, false>), dim3((unsigned)nblk), dim3(256), 0, st, codes, n,
                           (int)n_cols, drop_first, v, col_map, out);
    else if (quads)
        hipLaunchKernelGGL((cat_matvec_quad_kernel<F, false, false>), dim3((unsigned)nblk), dim3(256), 0, st, codes, n,
                           (int)n_cols, drop_first, v, col_map, out);
    else {
        if (assign) TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)n, st));
        hipLaunchKernelGGL((cat_matvec_kernel<F>), dim3((unsigned)nblk), dim3(256), 0, st, codes, n,
                           drop_first, v, col_map, out);
    }
    prof_end(st);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

// ---------------------------------------------------------------------------------------
// FUSED cross terms of ALL categorical blocks of a SplitMatrix with its dense / sparse block.
// One pass over the wide operand serves every categorical: the LDS tile holds the stacked
// outputs [sum_c ncol_c][TJ] of a TJ-column slice of the wide operand (parts = column slices),
// so the dense block is read from HBM exactly once for all categoricals together (instead of
// once per categorical) and the 4-byte codes are re-read once per part.
// ---------------------------------------------------------------------------------------
constexpr int MAX_CATS = 16;
struct CatSet {
    const int32_t *codes[MAX_CATS];
    int ncol[MAX_CATS];
    int drop[MAX_CATS];
    int off[MAX_CATS];   // row offset of categorical c in the stacked output
    int n_cats;
    int total;           // sum of ncol
};

// C-ordered dense: TJ lanes per row (64 / TJ rows per wave step), lane <-> dense column.
// NC = number of categoricals when <= 4 (codes of UNR rows x NC categoricals are loaded up
// front so that ~UNR * (2 + NC) independent loads are in flight per lane), 0 = generic loop.
template <typename F, int TJ, int NC>
__global__ __launch_bounds__(1024) void multi_cat_dense_c_kernel(
    CatSet cs, const F *__restrict__ d, const F *__restrict__ M, int64_t n, int64_t m,
    int64_t rows_per_block, F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [total][TJ]
    const int nel = cs.total * TJ;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    constexpr int RPW = 64 / TJ;
    constexpr int UNR = 4;
    constexpr int NCC = NC > 0 ? NC : 1;
    const int lane = threadIdx.x & 63;
    const int sub = lane / TJ, jl = lane % TJ;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int64_t j = (int64_t)blockIdx.y * TJ + jl;
    const bool jok = j < m;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n);
    const int64_t step = (int64_t)nwave * RPW;
    for (int64_t k0 = t0 + (int64_t)wave * RPW + sub; k0 < t1; k0 += UNR * step) {
        F x[UNR];
        int cc[UNR][NCC];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t k = k0 + u * step;
            const bool ok = k < t1;
            x[u] = (ok && jok) ? d[k] * M[k * m + j] : F(0);
            if (NC > 0) {
#pragma unroll
                for (int c = 0; c < NCC; ++c) cc[u][c] = ok ? cs.codes[c][k] - cs.drop[c] : -1;
            }
        }
        if (NC > 0) {
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int c = 0; c < NCC; ++c)
                    if (jok && cc[u][c] >= 0)
                        atomic_add(&tile[(cs.off[c] + cc[u][c]) * TJ + jl], (lds_acc_t)x[u]);
        } else {
#pragma unroll 1
            for (int c = 0; c < cs.n_cats; ++c) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int64_t k = k0 + u * step;
                    const int c1 = k < t1 ? cs.codes[c][k] - cs.drop[c] : -1;
                    if (jok && c1 >= 0) atomic_add(&tile[(cs.off[c] + c1) * TJ + jl], (lds_acc_t)x[u]);
                }
            }
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) dst[b] = (F)tile[b];
}

// C-ordered dense, wide loads (the fast path for aligned operands and <= 8 categoricals):
// a wave step covers MCW_RS = 32 rows x TJ = 16 * VEC columns (VEC = columns per 16-byte load):
// 8 global_load_dwordx4 per lane bring the 32 rows in (16 lanes per row, 4 rows per instruction),
// d and the codes arrive lane <-> row with one coalesced load each and are parked in a per-wave
// LDS scratch as {d, byte offset of the level's tile row}, read back per 4-row group (4 distinct
// addresses per read).  The loads of step t + 1 are in flight while step t issues its atomics
// (two register sets, the loop is unrolled by two: no copies of in-flight loads).
// Tile layout [level][TJ] with the columns of a lane de-interleaved: column c sits at
// (c / VEC) + 16 * (c % VEC), so one ds_add covers 16 consecutive elements per row.
// The narrow kernel above spends one 8-byte load instruction per lane and element and reloads the
// codes in every lane: ~20 vector-memory instructions per 2 KB of the dense operand.
constexpr int MCW_RS = 32;
constexpr int MCW_VEC = 2;

template <typename F, int NC>
__global__ __launch_bounds__(1024) void multi_cat_dense_wide_kernel(
    CatSet cs, const F *__restrict__ d, const F *__restrict__ M, int64_t n, int64_t m,
    int64_t rows_per_block, F *__restrict__ ws, int64_t stride, const int32_t *__restrict__ rows,
    WgLogBuf *__restrict__ wglog) {
    const unsigned long long t_begin = wg_log_begin(wglog);
    // rows != NULL: `n` is the length of the row list and every position is mapped through it
    // (cost proportional to the list; the reference's `for k in rows`, ext/split.pyx:32-80)
    // VEC = 2 columns per lane for both types (16-byte loads for f64, 8-byte loads for f32): the
    // tile is made of doubles, and 32 columns per part is what fits LDS at a few hundred levels
    constexpr int VEC = MCW_VEC;
    constexpr int TJ = 16 * VEC;
    constexpr int NI = MCW_RS / 4;               // load instructions per step
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [total][TJ], lane-column de-interleaved
    const int nel = cs.total * TJ;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    // per-wave scratch: d[MCW_RS] then NC x int[MCW_RS] tile-row byte offsets (-1: no level)
    unsigned char *scr = smem_raw + (((size_t)nel * sizeof(lds_acc_t) + 15) / 16) * 16 +
                         (size_t)wave * MCW_RS * (sizeof(F) + NC * sizeof(int));
    F *sd = reinterpret_cast<F *>(scr);
    int *sc = reinterpret_cast<int *>(scr + MCW_RS * sizeof(F));
    __syncthreads();
    const int q = lane >> 4, jl = lane & 15;
    const int64_t j = (int64_t)blockIdx.y * TJ + jl * VEC;
    const int64_t jc = min(j, m - VEC);          // clamped (a part past the last column adds nothing)
    const bool jok = j < m;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n);
    const int64_t step = (int64_t)nwave * MCW_RS;
    const int lr = lane & (MCW_RS - 1);          // row of the step this lane loads d / codes for

    struct Regs { vec_t x[NI]; F dk; int code[NC]; };
    auto load = [&](int64_t k0, Regs &R) {
        // (row list: all row ids first, then the row loads -- a row id fetched inside the loop was
        // waited for before the next one went out)
        int64_t kr[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) kr[i] = min(k0 + 4 * i + q, n - 1);
        if (rows) {
#pragma unroll
            for (int i = 0; i < NI; ++i) kr[i] = (int64_t)rows[kr[i]];
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            // (streamed once: nontemporal loads keep the dense block out of the L2 -- 2.01 -> 1.71 ms at cfg4)
            R.x[i] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(M + kr[i] * m + jc));
        }
        const int64_t k = k0 + lr;
        const int64_t kp = min(k, n - 1);
        const int64_t kc = rows ? (int64_t)rows[kp] : kp;
        R.dk = d[kc];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = cs.codes[c][kc] - cs.drop[c];
            R.code[c] = (k < t1 && col >= 0) ? (cs.off[c] + col) * TJ * (int)sizeof(lds_acc_t) : -1;
        }
    };
    auto process = [&](const Regs &R) {
        __builtin_amdgcn_wave_barrier();
        if (lane < MCW_RS) {
            sd[lr] = R.dk;
#pragma unroll
            for (int c = 0; c < NC; ++c)     // rows with d == 0 contribute exactly nothing
                sc[c * MCW_RS + lr] = R.dk != F(0) ? R.code[c] : -1;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int row = 4 * i + q;
            const F dk = sd[row];
            int off[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) off[c] = sc[c * MCW_RS + row];
            vec_t x = R.x[i];
#pragma unroll
            for (int v = 0; v < VEC; ++v) x[v] *= dk;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (jok && off[c] >= 0) {
                    lds_acc_t *dst = reinterpret_cast<lds_acc_t *>(smem_raw + off[c]) + jl;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) atomic_add(dst + 16 * v, (lds_acc_t)x[v]);
                }
            }
        }
    };
    Regs ra, rb;
    int64_t k0 = t0 + (int64_t)wave * MCW_RS;
    if (k0 < t1) load(k0, ra);
    for (; k0 < t1; k0 += 2 * step) {
        const bool more = k0 + step < t1;
        if (more) load(k0 + step, rb);
        process(ra);
        if (more) {
            if (k0 + 2 * step < t1) load(k0 + 2 * step, ra);
            process(rb);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) wg_log_end(wglog, t_begin, WG_CAT_DENSE);
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) {
        const int c = b % TJ;
        dst[b] = (F)tile[(b / TJ) * TJ + (c / VEC) + 16 * (c % VEC)];
    }
}

// F-ordered dense: lane <-> row, loop over the TJ columns of the part.
template <typename F, int TJ>
__global__ __launch_bounds__(1024) void multi_cat_dense_f_kernel(
    CatSet cs, const F *__restrict__ d, const F *__restrict__ M, int64_t n, int64_t m,
    int64_t rows_per_block, F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);
    const int nel = cs.total * TJ;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    const int64_t j0 = (int64_t)blockIdx.y * TJ;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n);
    for (int64_t k = t0 + threadIdx.x; k < t1; k += blockDim.x) {
        const F dk = d[k];
        int col[MAX_CATS];
#pragma unroll
        for (int c = 0; c < MAX_CATS; ++c)
            col[c] = c < cs.n_cats ? cs.codes[c][k] - cs.drop[c] : -1;
        for (int jl = 0; jl < TJ && j0 + jl < m; ++jl) {
            const F x = dk * M[(j0 + jl) * n + k];
#pragma unroll
            for (int c = 0; c < MAX_CATS; ++c)
                if (col[c] >= 0) atomic_add(&tile[(cs.off[c] + col[c]) * TJ + jl], (lds_acc_t)x);
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) dst[b] = (F)tile[b];
}

// tmp [part][total][TJ] -> out[total][m]
template <typename F>
__global__ void multi_cat_untile_kernel(const F *__restrict__ tmp, int64_t total, int64_t m,
                                        int TJ, F *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total * m) return;
    const int64_t a = e / m, j = e % m;
    out[e] = tmp[((j / TJ) * total + a) * TJ + (j % TJ)];
}

// Sparse operand in slab-blocked column-major form (sparse.hip): part = one 32-column group,
// lane <-> nonzero of the (slab, group) stream.  The LDS tile has a row stride of
// group_cols + 1: consecutive stream entries belong to the same column (runs of ~6 rows), with an
// unpadded stride they would all fall on one bank pair.  STAGE: every wave first copies the
// codes and d of its 128-row slab into its own LDS scratch (coalesced loads), the per-nonzero
// lookups are then LDS reads instead of 64-lane global gathers.
template <typename F, bool STAGE>
__global__ __launch_bounds__(1024) void multi_cat_sparse_kernel(
    CatSet cs, const F *__restrict__ d, const F *__restrict__ vals,
    const unsigned *__restrict__ koff, const unsigned char *__restrict__ ecol,
    const int64_t *__restrict__ gptr, int n_groups, int64_t n_slabs, int64_t slabs_per_block,
    int slab_rows, int group_cols, int64_t n, F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [total][group_cols + 1]
    const int tstr = group_cols + 1;
    const int nel = cs.total * tstr;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    // per-wave scratch (STAGE): d of the slab rows, then the codes of every categorical
    F *sd = reinterpret_cast<F *>(smem_raw + (((size_t)nel * sizeof(lds_acc_t) + 15) / 16) * 16) +
            (size_t)wave * slab_rows;
    int32_t *sc = reinterpret_cast<int32_t *>(
                      reinterpret_cast<F *>(smem_raw + (((size_t)nel * sizeof(lds_acc_t) + 15) / 16) * 16) +
                      (size_t)nwave * slab_rows) +
                  (size_t)wave * cs.n_cats * slab_rows;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    const unsigned rowb = 64u * (unsigned)sizeof(F);
    // every wave walks its own slabs (s0 + wave, + nwave, ...): 16 independent streams per
    // workgroup; the pointers of the next slab are requested before the current one is used
    int64_t nb = 0, ne = 0;
    if (s0 + wave < s1) {
        nb = gptr[(s0 + wave) * n_groups + g];
        ne = gptr[(s0 + wave) * n_groups + g + 1];
    }
    for (int64_t s = s0 + wave; s < s1; s += nwave) {
        const int64_t base = nb, end = ne;
        if (s + nwave < s1) {
            nb = gptr[(s + nwave) * n_groups + g];
            ne = gptr[(s + nwave) * n_groups + g + 1];
        }
        if (STAGE) {
            __builtin_amdgcn_wave_barrier();
            for (int r = lane; r < slab_rows; r += 64) {
                const int64_t k = min(s * slab_rows + r, n - 1);
                sd[r] = d[k];
                for (int c = 0; c < cs.n_cats; ++c) sc[c * slab_rows + r] = cs.codes[c][k] - cs.drop[c];
            }
            __builtin_amdgcn_wave_barrier();
        }
        for (int64_t e0 = base + lane; e0 < end; e0 += 128) {
            const int64_t e1 = e0 + 64;
            const bool ok1 = e1 < end;
            const unsigned ko0 = koff[e0];
            const unsigned ko1 = ok1 ? koff[e1] : 0u;
            const F v0 = vals[e0];
            const F v1 = ok1 ? vals[e1] : F(0);
            const int ec0 = ecol[e0];
            const int ec1 = ok1 ? ecol[e1] : 0;
            const int r0 = (int)(ko0 / rowb), r1 = (int)(ko1 / rowb);
            const int64_t k0 = s * slab_rows + r0;
            const int64_t k1 = s * slab_rows + r1;
            const F d0 = STAGE ? sd[r0] : d[k0];
            const F d1 = ok1 ? (STAGE ? sd[r1] : d[k1]) : F(0);
            const F x0 = d0 != F(0) ? d0 * v0 : F(0);   // rows masked out by d == 0 contribute
            const F x1 = d1 != F(0) ? d1 * v1 : F(0);   // exactly nothing, whatever they hold
#pragma unroll 1
            for (int c = 0; c < cs.n_cats; ++c) {
                const int c0 = STAGE ? sc[c * slab_rows + r0] : cs.codes[c][k0] - cs.drop[c];
                const int c1 = ok1 ? (STAGE ? sc[c * slab_rows + r1] : cs.codes[c][k1] - cs.drop[c])
                                   : -1;
                if (c0 >= 0) atomic_add(&tile[(cs.off[c] + c0) * tstr + ec0], (lds_acc_t)x0);
                if (c1 >= 0) atomic_add(&tile[(cs.off[c] + c1) * tstr + ec1], (lds_acc_t)x1);
            }
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < cs.total * group_cols; b += blockDim.x)
        dst[b] = (F)tile[(b / group_cols) * tstr + (b % group_cols)];
}

// Fast variant for 1..4 categoricals: one (slab, group) block per wave step.  While a block is
// processed, the first 256 stream entries of the wave's NEXT block and the d / codes of that
// slab's 128 rows are already in flight into registers (one memory round trip per block, hidden
// behind the previous block's atomics); d and codes are then parked in per-wave LDS scratch so the
// per-nonzero lookups are LDS reads.
template <typename F, int NC>
__global__ __launch_bounds__(1024) void multi_cat_sparse_pf_kernel(
    CatSet cs, const F *__restrict__ d, const F *__restrict__ vals,
    const unsigned *__restrict__ koff, const unsigned char *__restrict__ ecol,
    const int64_t *__restrict__ gptr, int n_groups, int64_t n_slabs, int64_t slabs_per_block,
    int group_cols, int64_t n, F *__restrict__ ws, int64_t stride, WgLogBuf *__restrict__ wglog) {
    const unsigned long long t_begin = wg_log_begin(wglog);
    constexpr int SR = 128;              // slab rows (tm_slab_rows)
    constexpr int NQ = 4;                // prefetched 64-entry chunks per block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [total][group_cols + 1]
    const int tstr = group_cols + 1;
    const int nel = cs.total * tstr;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    unsigned char *scratch = smem_raw + (((size_t)nel * sizeof(lds_acc_t) + 15) / 16) * 16;
    F *sd = reinterpret_cast<F *>(scratch) + (size_t)wave * SR;
    int32_t *sc = reinterpret_cast<int32_t *>(reinterpret_cast<F *>(scratch) + (size_t)nwave * SR) +
                  (size_t)wave * NC * SR;
    __syncthreads();
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    const unsigned rowb = 64u * (unsigned)sizeof(F);

    // pointers two blocks ahead, stream + row data one block ahead
    int64_t q_base = 0, q_end = 0;
    int64_t p_base = 0, p_end = 0;
    unsigned p_ko[NQ];
    F p_v[NQ];
    int p_ec[NQ];
    F p_d[2];
    int p_c[NC][2];
    auto load_ptrs = [&](int64_t s) {
        q_base = q_end = 0;
        if (s < s1) {
            q_base = gptr[s * n_groups + g];
            q_end = gptr[s * n_groups + g + 1];
        }
    };
    auto load_block = [&](int64_t s) {     // consumes the pointers, issues the loads of block s
        p_base = q_base;
        p_end = q_end;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int64_t e = p_base + q * 64 + lane;
            const int64_t ec = min(e, max(p_end - 1, p_base));     // clamped: branch-free issue
            const bool ok = e < p_end;
            p_ko[q] = ok ? koff[ec] : 0u;
            p_v[q] = ok ? vals[ec] : F(0);
            p_ec[q] = ok ? (int)ecol[ec] : 0;
        }
        if (s < s1) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int64_t k = min(s * SR + lane + 64 * h, n - 1);
                p_d[h] = d[k];
#pragma unroll
                for (int c = 0; c < NC; ++c) p_c[c][h] = cs.codes[c][k] - cs.drop[c];
            }
        }
    };
    const int64_t sw = s0 + wave;
    load_ptrs(sw);
    load_block(sw);
    load_ptrs(sw + nwave);
    for (int64_t s = sw; s < s1; s += nwave) {
        // take over the prefetched block
        const int64_t base = p_base, end = p_end;
        unsigned ko[NQ];
        F v[NQ];
        int ec[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) { ko[q] = p_ko[q]; v[q] = p_v[q]; ec[q] = p_ec[q]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            sd[lane + 64 * h] = p_d[h];
#pragma unroll
            for (int c = 0; c < NC; ++c) sc[c * SR + lane + 64 * h] = p_c[c][h];
        }
        __builtin_amdgcn_wave_barrier();
        load_block(s + nwave);
        load_ptrs(s + 2 * nwave);
        auto scatter = [&](unsigned kk, F vv, int ee) {
            const int r = (int)(kk / rowb);
            const F dk = sd[r];
            const F x = dk != F(0) ? dk * vv : F(0);   // rows masked out by d == 0 contribute nothing
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int cc = sc[c * SR + r];
                if (cc >= 0) atomic_add(&tile[(cs.off[c] + cc) * tstr + ee], (lds_acc_t)x);
            }
        };
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (base + q * 64 + lane < end) scatter(ko[q], v[q], ec[q]);
        // blocks longer than the prefetch window (dense columns): the rest straight from memory
        for (int64_t e = base + NQ * 64 + lane; e < end; e += 64) scatter(koff[e], vals[e], (int)ecol[e]);
    }
    __syncthreads();
    if (threadIdx.x == 0) wg_log_end(wglog, t_begin, WG_CAT_SPARSE);
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < cs.total * group_cols; b += blockDim.x)
        dst[b] = (F)tile[(b / group_cols) * tstr + (b % group_cols)];
}

// The same fused categorical x sparse cross terms on the ENTRY twin of the sparse block (round 4: the stream K3
// walks, csrc/sparse_ent.hip / tabmat_amd/ext/_types.py::SlabEnt -- {value, row << 4 | column in group} in batches
// of 16 slots, blocks of a 16-column group slab after slab), so that a design whose sparse x dense term runs on the
// entry kernel needs no slab-form twin (3.4 GB at BASELINE configs[3]) for this term.  A workgroup owns TWO column
// groups (32 kernel columns: tile [total levels][33] of doubles, as the slab kernel) over a range of slabs; its
// waves alternate between the two groups and cut the range among themselves; lane <-> slot, 64 slots per step.
// The d and the codes of a slot's row are gathered from global memory (rows of a (group, slab) block lie within
// 64 rows: the lines stay in the L1 / L2); the stream of the step after next and the gathers of the next step are
// in flight while the current one is scattered (the slab kernel's lesson: it was latency-bound, not atomic-bound).
#ifndef EN_CS_U
#define EN_CS_U 4
#endif
template <typename F, int NC>
__global__ __launch_bounds__(1024) void multi_cat_sparse_ent_kernel(
    CatSet cs, const F *__restrict__ d, const F *__restrict__ vals, const unsigned short *__restrict__ meta,
    const unsigned *__restrict__ bstart, int n_groups, int64_t n, int64_t n_slabs, int64_t slabs_per_block,
    F *__restrict__ ws, int64_t stride, const unsigned *__restrict__ packed) {
    // packed (round 5, may be NULL; NC <= 3): packed[row] = the tile rows of the row's levels, 10 bits each (1023 =
    // none: missing / dropped level), built once per matrix by tm_multi_cat_pack_codes -- ONE gather per slot
    // instead of NC (the kernel is bound by its dependent loads: no gathers at all 0.60 ms against 1.02,
    // profiles/r4_catsparse.txt)
    constexpr int GC = 32, TSTR = GC + 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [total][33]
    const int nel = cs.total * TSTR;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int h = wave & 1;                               // which of the pair's two groups
    const int g = 2 * blockIdx.y + h;
    const int nwh = (nwave + 1 - h) / 2;                  // waves on this group
    const int wi = wave >> 1;
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    if (g < n_groups && s1 > s0) {
        const int64_t spw = (s1 - s0 + nwh - 1) / nwh;
        const int64_t sa = min(s0 + (int64_t)wi * spw, s1), sb = min(sa + spw, s1);
        const unsigned *brow = bstart + (int64_t)g * (n_slabs + 1);
        const int64_t e0 = (int64_t)brow[sa] * 16, e1 = (int64_t)brow[sb] * 16;     // slots of the wave
        // U x 64 slots per step and lane: the kernel is bound by the memory round trips of its two dependent
        // loads (stream -> d / codes of the row), not by the atomics -- more of them in flight per wave
        constexpr int U = EN_CS_U;
        struct Step { F v[U]; unsigned m[U]; F dk[U]; int c[U][NC]; };
        auto load_stream = [&](int64_t e, Step &t) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = e + 64 * u + lane < e1;
                const int64_t i = ok ? e + 64 * u + lane : (e1 > e0 ? e1 - 1 : 0);
                t.v[u] = ok ? vals[i] : F(0);
                t.m[u] = meta[i];
            }
        };
        // (round 6: 16-bit meta words {slab & 63, row in slab, column}: the slab of a slot from the tag and a running
        // slab, in stream order -- load_rows is called step after step; rows of slots behind the wave's stream clamped)
        unsigned cur_slab = (unsigned)sa;
        const unsigned row_max = (unsigned)(n - 1);
        auto load_rows = [&](Step &t) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned slab = cur_slab + (((t.m[u] >> 10) - cur_slab) & 63u);
                cur_slab = (unsigned)__builtin_amdgcn_readlane((int)slab, 63);
                const unsigned row = min((slab << 6) | ((t.m[u] >> 4) & 63u), row_max);
                t.dk[u] = d[row];
                if (NC <= 3 && packed != nullptr) {
                    const unsigned pk = packed[row];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const int f = (int)((pk >> (10 * c)) & 1023u);
                        t.c[u][c] = f == 1023 ? -1 : f - cs.off[c];
                    }
                    continue;
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) t.c[u][c] = cs.codes[c][row] - cs.drop[c];
            }
        };
        if (e1 > e0) {
            Step cur, nxt, nx2;
            load_stream(e0, cur);
            load_stream(e0 + 64 * U, nxt);
            load_rows(cur);
            for (int64_t e = e0; e < e1; e += 64 * U) {
                load_stream(e + 128 * U, nx2);
                load_rows(nxt);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    // rows masked out by d == 0 (and padding slots: value 0) contribute exactly nothing
                    const F x = (cur.dk[u] != F(0) && cur.v[u] != F(0)) ? cur.dk[u] * cur.v[u] : F(0);
                    const int col = 16 * h + (int)(cur.m[u] & 15u);
                    if (x != F(0)) {
#pragma unroll
                        for (int c = 0; c < NC; ++c)
                            if (cur.c[u][c] >= 0)
                                atomic_add(&tile[(cs.off[c] + cur.c[u][c]) * TSTR + col], (lds_acc_t)x);
                    }
                }
                cur = nxt;
                nxt = nx2;
            }
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < cs.total * GC; b += blockDim.x)
        dst[b] = (F)tile[(b / GC) * TSTR + (b % GC)];
}

// The same on the entry twin with the per-row operands STAGED (round 6).  The kernel above is bound
// by its dependent loads: stream -> {d, codes} of the slot's row (no gathers at all 0.60 ms against 1.02,
// profiles/r4_catsparse.txt).  A (group, slab) block only names rows of ITS 64-row slab, and a wave walks the slabs of
// its range in order, so what a slot needs of its row is known without looking at the stream: the wave loads d and the
// code word(s) of the slab's 64 rows lane <-> row (two coalesced loads), EN_CS_D slabs ahead together with the block's
// stream, parks them in 768 bytes of per-wave LDS when the slab's turn comes, and every slot reads its row's pair from
// there -- no load depends on another load.  One step of 64 slots per block (a block of more than 64 slots: further
// steps straight from memory; 3 % of the blocks at BASELINE configs[3]).  PK: the packed code word of
// tm_multi_cat_pack_codes (NC <= 3), else NC code words per row.
#ifndef EN_CS_D
#define EN_CS_D 4
#endif
#ifndef EN_CS_BOTH
#define EN_CS_BOTH 1
#endif
template <typename F, int NC, bool PK>
__global__ __launch_bounds__(1024) void multi_cat_sparse_ent_staged_kernel(
    CatSet cs, const F *__restrict__ d, const F *__restrict__ vals, const unsigned short *__restrict__ meta,
    const unsigned *__restrict__ bstart, int n_groups, int64_t n, int64_t n_slabs, int64_t slabs_per_block,
    F *__restrict__ ws, int64_t stride, const unsigned *__restrict__ packed, int tile_bytes) {
    constexpr int GC = 32, TSTR = GC + 1;
    constexpr int NCW = PK ? 1 : NC;                      // staged code words per row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [total][33]
    const int nel = cs.total * TSTR;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwave = blockDim.x >> 6;
    F *sd = reinterpret_cast<F *>(smem_raw + tile_bytes) + wave * 64;                                      // d of the slab's rows
    int *sc = reinterpret_cast<int *>(smem_raw + tile_bytes + (size_t)nwave * 64 * sizeof(F)) + wave * 64 * NCW;
    // EN_CS_BOTH: a wave walks BOTH groups of the pair over its slabs (the rows' operands are fetched and parked once per
    // slab for two blocks: 10 instead of 12 loads and 2 instead of 4 staging writes per slab and pair); else the waves
    // alternate between the two groups.
    constexpr int NG = EN_CS_BOTH ? 2 : 1;
    const int h0 = EN_CS_BOTH ? 0 : (wave & 1);
    const int g0 = 2 * blockIdx.y + h0;
    const int nwh = EN_CS_BOTH ? nwave : (nwave + 1 - h0) / 2;     // waves that share this group's slabs
    const int wi = EN_CS_BOTH ? wave : (wave >> 1);
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    if (g0 < n_groups && s1 > s0) {
        const int64_t spw = (s1 - s0 + nwh - 1) / nwh;
        const int64_t sa = min(s0 + (int64_t)wi * spw, s1), sb = min(sa + spw, s1);
        // (an odd number of groups: the last pair's second group does not exist -- its bounds are read from the first
        // group's row and its blocks count as empty)
        const bool has[2] = {true, g0 + 1 < n_groups};
        const unsigned *brow[2] = {bstart + (int64_t)g0 * (n_slabs + 1),
                                   bstart + (int64_t)(has[1] ? g0 + 1 : g0) * (n_slabs + 1)};
        // Every load of the walk is UNCONDITIONAL (clamped addresses, results masked afterwards): the compiler can only
        // count its waits (s_waitcnt vmcnt(N) for the requests of EN_CS_D slabs ago, the newer ones stay in flight)
        // over loads it knows were issued -- a first version with the loads behind `if (slot < slots of the block)`
        // waited with vmcnt(0) in front of every slab, one memory round trip per slab: 1.27 ms against 0.94 for the
        // gather kernel.  Three stages, EN_CS_D slabs apart: block bounds (two broadcast loads) -> stream + the rows'
        // operands -> scatter.
        struct Bnd { unsigned b0[NG], b1[NG]; };
        struct Pre { F v[NG]; unsigned m[NG]; F dd; int c[NCW]; unsigned b0[NG]; int nsl[NG]; };
        // (the bounds through VECTOR loads at a lane-invariant address the compiler cannot see through: scalar loads
        // share their counter with the LDS atomics, every use would wait for those too)
        int vz = 0;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
#endif
        auto request_bounds = [&](int64_t s, Bnd &q) {
            const int64_t sc_ = min(s, sb - 1) + vz;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                q.b0[k] = brow[k][sc_];
                q.b1[k] = brow[k][sc_ + 1];
            }
        };
        auto request = [&](int64_t s, const Bnd &q, Pre &p) {
            const bool live = s < sb;
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const unsigned qb0 = (unsigned)__builtin_amdgcn_readfirstlane((int)q.b0[k]);
                const unsigned qb1 = (unsigned)__builtin_amdgcn_readfirstlane((int)q.b1[k]);
                p.b0[k] = qb0;
                p.nsl[k] = (live && has[k]) ? (int)(qb1 - qb0) * 16 : 0;
                const bool ok = lane < p.nsl[k];
                const int64_t e = (int64_t)qb0 * 16 + (ok ? lane : 0);        // (slot b0 * 16 exists: slack behind the stream)
                const F v = __builtin_nontemporal_load(vals + e);
                p.m[k] = __builtin_nontemporal_load(meta + e);
                p.v[k] = ok ? v : F(0);
            }
            const int64_t row_ = min(s, sb - 1) * 64 + lane;
            const int64_t row = min(row_, n - 1);              // (ragged last slab: no slot names a row beyond n - 1)
            p.dd = d[row];
            if constexpr (PK) {
                p.c[0] = (int)packed[row];
            } else {
#pragma unroll
                for (int c = 0; c < NC; ++c) p.c[c] = cs.codes[c][row] - cs.drop[c];
            }
        };
        auto process = [&](Pre &p) {
            int any = 0;
#pragma unroll
            for (int k = 0; k < NG; ++k) any |= p.nsl[k];
            if (any == 0) return;
            sd[lane] = p.dd;
#pragma unroll
            for (int c = 0; c < NCW; ++c) sc[c * 64 + lane] = p.c[c];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                F v = p.v[k];
                unsigned m = p.m[k];
                const int cbase = 16 * (h0 + k);
                for (int off = 0; off < p.nsl[k]; off += 64) {
                    if (off > 0) {                 // a block of more than 64 slots: the rest straight from memory
                        const bool ok = off + lane < p.nsl[k];
                        const int64_t e = (int64_t)p.b0[k] * 16 + off + (ok ? lane : 0);
                        v = ok ? vals[e] : F(0);
                        m = meta[e];
                    }
                    const int r6 = (int)((m >> 4) & 63u);
                    const int col = cbase + (int)(m & 15u);
                    const F dk = sd[r6];
                    // rows masked out by d == 0 (and padding slots: value 0) contribute exactly nothing
                    const F x = (dk != F(0) && v != F(0)) ? dk * v : F(0);
                    if (x != F(0)) {
                        if constexpr (PK) {
                            const unsigned pk = (unsigned)sc[r6];
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                const int f = (int)((pk >> (10 * c)) & 1023u);
                                if (f != 1023) atomic_add(&tile[f * TSTR + col], (lds_acc_t)x);
                            }
                        } else {
#pragma unroll
                            for (int c = 0; c < NC; ++c) {
                                const int cc = sc[c * 64 + r6];
                                if (cc >= 0) atomic_add(&tile[(cs.off[c] + cc) * TSTR + col], (lds_acc_t)x);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        constexpr int DEPTH = EN_CS_D;             // slabs in flight per wave and stage
        Pre p[DEPTH];
        Bnd q[DEPTH];
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) request_bounds(sa + j, q[j]);
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            request(sa + j, q[j], p[j]);
            request_bounds(sa + DEPTH + j, q[j]);
        }
        for (int64_t s = sa; s < sb; s += DEPTH) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                process(p[j]);
                request(s + DEPTH + j, q[j], p[j]);
                request_bounds(s + 2 * DEPTH + j, q[j]);
            }
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < cs.total * GC; b += blockDim.x)
        dst[b] = (F)tile[(b / GC) * TSTR + (b % GC)];
}

// Row-list form of the fused categorical x sparse cross terms (the reference's cost for `rows=` is
// proportional to len(rows): categorical_matrix.py:825-838 works on self[rows]): the selected rows'
// entry lists come from the chunk-major twin through a {start, end} table [chunk][selected row]
// (the table the row-list K2 uses), part = one 32-column group of a 128-column chunk, tile
// [total levels][33] of doubles in LDS.  8 lanes per selected row walk its list in the group's
// chunk and keep the entries of the group; codes and d are read once per row.
// (IX = uint8_t: the columns as bytes inside their 128-column chunk, round 6)
template <typename F, typename IX>
__global__ __launch_bounds__(1024) void multi_cat_sparse_rows_kernel(
    CatSet cs, const F *__restrict__ cm_data, const IX *__restrict__ cm_ind,
    const int32_t *__restrict__ ranges, const int32_t *__restrict__ rows, const F *__restrict__ d_sel,
    int64_t n_sel, int64_t rows_per_block, F *__restrict__ ws, int64_t stride) {
    constexpr int GC = 32;                       // columns per group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [total][GC + 1]
    constexpr int tstr = GC + 1;
    const int nel = cs.total * tstr;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    const int g = blockIdx.y;                    // 32-column group; chunk = g / 4
    const int ch = g >> 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    const int lr = lane >> 3, lt = lane & 7;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_sel);
    const int32_t *rg = ranges + (int64_t)ch * n_sel * 2;
    for (int64_t k0 = t0 + wave * 8; k0 < t1; k0 += (int64_t)nwave * 8) {
        const int64_t k = k0 + lr;
        if (k >= t1) continue;
        const F dk = d_sel[k];
        if (dk == F(0)) continue;                // rows with d == 0 contribute exactly nothing
        const int e0 = rg[2 * k], e1 = rg[2 * k + 1];
        if (e0 >= e1) continue;
        const int64_t row = rows[k];
        int off[MAX_CATS];
#pragma unroll
        for (int c = 0; c < MAX_CATS; ++c) {
            off[c] = -1;
            if (c < cs.n_cats) {
                const int col = cs.codes[c][row] - cs.drop[c];
                off[c] = col >= 0 ? (cs.off[c] + col) * tstr : -1;
            }
        }
        for (int e = e0 + lt; e < e1; e += 8) {
            const int col = (int)cm_ind[e] - (sizeof(IX) == 1 ? (g & 3) : g) * GC;
            if (col < 0 || col >= GC) continue;
            const lds_acc_t x = (lds_acc_t)(dk * cm_data[e]);
#pragma unroll
            for (int c = 0; c < MAX_CATS; ++c)
                if (off[c] >= 0) atomic_add(&tile[off[c] + col], x);
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < cs.total * GC; b += blockDim.x)
        dst[b] = (F)tile[(b / GC) * tstr + (b % GC)];
}

static int make_catset(const void *const *h_codes, const int64_t *h_ncols, const int32_t *h_drop,
                       int n_cats, CatSet *cs) {
    if (n_cats < 1 || n_cats > MAX_CATS) {
        set_error("multi-cat call supports 1..%d categoricals, got %d", MAX_CATS, n_cats);
        return TM_EINVAL;
    }
    int off = 0;
    for (int c = 0; c < MAX_CATS; ++c) {
        cs->codes[c] = c < n_cats ? reinterpret_cast<const int32_t *>(h_codes[c]) : nullptr;
        cs->ncol[c] = c < n_cats ? (int)h_ncols[c] : 0;
        cs->drop[c] = c < n_cats ? h_drop[c] : 0;
        cs->off[c] = off;
        off += cs->ncol[c];
    }
    cs->n_cats = n_cats;
    cs->total = off;
    return TM_OK;
}

template <typename F>
static int run_multi_cat_dense(const void *const *h_codes, const int64_t *h_ncols,
                               const int32_t *h_drop, int n_cats, int64_t n, const F *d, const F *M,
                               int64_t m, int order_f, F *out, hipStream_t st,
                               const int32_t *rows = nullptr, int64_t n_rows = 0) {
    CatSet cs;
    int rc = make_catset(h_codes, h_ncols, h_drop, n_cats, &cs);
    if (rc) return rc;
    const int64_t total = (int64_t)cs.total * m;
    if (total == 0) return TM_OK;
    if (rows != nullptr) n = n_rows;          // positions of the row list from here on
    if (n == 0) {                             // (otherwise the untile kernel writes every element)
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    {
        // wide-load path: C-ordered, 16-byte aligned rows, <= 8 categoricals, tile + scratch in LDS
        constexpr int VEC = MCW_VEC;
        constexpr int TJW = 16 * VEC;
        const size_t tile_b = ((sizeof(lds_acc_t) * (size_t)cs.total * TJW + 15) / 16) * 16;
        const size_t lds_w = tile_b + (size_t)16 * MCW_RS * (sizeof(F) + (size_t)n_cats * sizeof(int));
        if (!order_f && n_cats <= 8 && m >= VEC && m % VEC == 0 &&
            (reinterpret_cast<uintptr_t>(M) & 15) == 0 && lds_w <= 150 * 1024 &&
            (int64_t)cs.total * TJW * (int64_t)sizeof(lds_acc_t) < (1ll << 30)) {
            const int64_t n_parts = ceil_div(m, TJW);
            const int64_t stride = (int64_t)cs.total * TJW;
            int64_t nblk = std::max<int64_t>(1, tune("catdense_rounds", 1) * NUM_CU / n_parts);
            nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, ceil_div(n, 4096)));
            const int64_t rpb = ceil_div(ceil_div(n, nblk), MCW_RS) * MCW_RS;
            nblk = ceil_div(n, rpb);
            const size_t tmp_bytes = ((sizeof(F) * (size_t)(n_parts * stride) + 255) / 256) * 256;
            void *wsv = nullptr;
            rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)(n_parts * nblk * stride) + 256, &wsv, st);
            if (rc) return rc;
            F *tmp = reinterpret_cast<F *>(wsv);
            F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
            auto gow = [&](auto kern) -> int {
                TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_w));
                prof_begin(st);
                // 16 waves own a CU; 12 leave registers for the co-resident syrk (syrk_co.hip).  With a large
                // f64 tile (cfg4: 384 levels x 32 columns = 96 KB) EIGHT waves are faster -- 1.63 against
                // 1.82 ms alone, 14.6 against 15.0 ms for the whole cfg4 step (fewer waves queue on the LDS
                // atomic pipe; 6, 10 and 12 waves gain nothing); small tiles and f32 stay at 16
                const int nw_dflt = (sizeof(F) == 8 && tile_b > 64 * 1024) ? 8 : 16;
                const int nw = (int)std::min<int64_t>(16, std::max<int64_t>(4, tune("catdense_waves", nw_dflt)));
                hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts), dim3(nw * 64), lds_w, st,
                                   cs, d, M, n, m, rpb, ws, stride, rows, wg_log_ptr());
                prof_end(st);
                TM_LAUNCH_CHECK();
                return TM_OK;
            };
            if (n_cats == 1) rc = gow(&multi_cat_dense_wide_kernel<F, 1>);
            else if (n_cats == 2) rc = gow(&multi_cat_dense_wide_kernel<F, 2>);
            else if (n_cats == 3) rc = gow(&multi_cat_dense_wide_kernel<F, 3>);
            else if (n_cats == 4) rc = gow(&multi_cat_dense_wide_kernel<F, 4>);
            else if (n_cats == 5) rc = gow(&multi_cat_dense_wide_kernel<F, 5>);
            else if (n_cats == 6) rc = gow(&multi_cat_dense_wide_kernel<F, 6>);
            else if (n_cats == 7) rc = gow(&multi_cat_dense_wide_kernel<F, 7>);
            else rc = gow(&multi_cat_dense_wide_kernel<F, 8>);
            if (rc) return rc;
            rc = launch_reduce_partials<F>(ws, stride, (int)nblk, (int)n_parts, tmp, n_parts * stride,
                                           false, st);
            if (rc) return rc;
            hipLaunchKernelGGL((multi_cat_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)),
                               dim3(256), 0, st, tmp, (int64_t)cs.total, m, TJW, out);
            TM_LAUNCH_CHECK();
            return TM_OK;
        }
    }
    if (rows != nullptr) {
        set_error("multi_cat_dense: the row-list form needs the wide-load path");
        return TM_EUNSUPPORTED;
    }
    int TJ = 64;
    while (TJ > 1 && sizeof(lds_acc_t) * (size_t)cs.total * TJ > HIST_LDS_MAX) TJ >>= 1;
    while (TJ > 1 && TJ / 2 >= m) TJ >>= 1;
    if (sizeof(lds_acc_t) * (size_t)cs.total * TJ > HIST_LDS_MAX) {
        set_error("multi_cat_dense: %d stacked categories exceed the LDS tile", cs.total);
        return TM_EUNSUPPORTED;
    }
    const int64_t n_parts = ceil_div(m, TJ);
    const int64_t stride = (int64_t)cs.total * TJ;
    const size_t lds = sizeof(lds_acc_t) * (size_t)stride;
    int64_t nblk = std::max<int64_t>(1, NUM_CU / n_parts);
    nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, ceil_div(n, 4096)));
    const int64_t rpb = ceil_div(n, nblk);
    nblk = ceil_div(n, rpb);
    const size_t tmp_bytes = ((sizeof(F) * (size_t)(n_parts * stride) + 255) / 256) * 256;
    void *wsv = nullptr;
    rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)(n_parts * nblk * stride) + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    auto go = [&](auto kern) -> int {
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts), dim3(1024), lds, st, cs, d,
                           M, n, m, rpb, ws, stride);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return TM_OK;
    };
#define TM_MCD_CASE(V)                                                                       \
    case V:                                                                                  \
        if (order_f) rc = go(&multi_cat_dense_f_kernel<F, V>);                               \
        else if (n_cats == 1) rc = go(&multi_cat_dense_c_kernel<F, V, 1>);                   \
        else if (n_cats == 2) rc = go(&multi_cat_dense_c_kernel<F, V, 2>);                   \
        else if (n_cats == 3) rc = go(&multi_cat_dense_c_kernel<F, V, 3>);                   \
        else if (n_cats == 4) rc = go(&multi_cat_dense_c_kernel<F, V, 4>);                   \
        else rc = go(&multi_cat_dense_c_kernel<F, V, 0>);                                    \
        break;
    switch (TJ) {
        TM_MCD_CASE(64)
        TM_MCD_CASE(32)
        TM_MCD_CASE(16)
        TM_MCD_CASE(8)
        TM_MCD_CASE(4)
        TM_MCD_CASE(2)
        TM_MCD_CASE(1)
        default:
            rc = TM_EINVAL;
    }
#undef TM_MCD_CASE
    if (rc) return rc;
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, (int)n_parts, tmp, n_parts * stride, false,
                                   st);
    if (rc) return rc;
    hipLaunchKernelGGL((multi_cat_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)),
                       dim3(256), 0, st, tmp, (int64_t)cs.total, m, TJ, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F, typename IX>
static int run_multi_cat_sparse_rows(const void *const *h_codes, const int64_t *h_ncols,
                                     const int32_t *h_drop, int n_cats, const F *cm_data,
                                     const IX *cm_ind, const int32_t *ranges, const int32_t *rows,
                                     int64_t n_sel, int64_t m, const F *d_sel, F *out, hipStream_t st);

template <typename F>
static int run_multi_cat_sparse(const void *const *h_codes, const int64_t *h_ncols,
                                const int32_t *h_drop, int n_cats, int64_t n, const F *d,
                                const F *vals, const unsigned *koff, const unsigned char *ecol,
                                const int64_t *gptr, int64_t m, int slab_rows, int group_cols,
                                F *out, hipStream_t st) {
    CatSet cs;
    int rc = make_catset(h_codes, h_ncols, h_drop, n_cats, &cs);
    if (rc) return rc;
    const int64_t total = (int64_t)cs.total * m;
    if (total == 0) return TM_OK;
    if (n == 0) {                             // (otherwise the untile kernel writes every element)
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    const int64_t stride = (int64_t)cs.total * group_cols;
    const size_t tile_bytes = ((sizeof(lds_acc_t) * (size_t)cs.total * (group_cols + 1) + 15) / 16) * 16;
    if (tile_bytes > HIST_LDS_MAX) {
        set_error("multi_cat_sparse: %d stacked categories exceed the LDS tile", cs.total);
        return TM_EUNSUPPORTED;
    }
    // per-wave staging of the slab's d and codes when it fits next to the tile
    // (16 waves own a CU; 12 leave registers and LDS for the co-resident syrk, syrk_co.hip)
    const int nw = (int)std::min<int64_t>(16, std::max<int64_t>(4, tune("catsparse_waves", 16)));
    const size_t stage_bytes = (size_t)nw * slab_rows * (sizeof(F) + sizeof(int32_t) * (size_t)n_cats);
    const bool stage = tile_bytes + stage_bytes <= 150 * 1024;
    const size_t lds = tile_bytes + (stage ? stage_bytes : 0);
    const int n_groups = (int)ceil_div(m, group_cols);
    const int64_t n_slabs = ceil_div(n, slab_rows);
    int64_t nblk = std::max<int64_t>(1, tune("catsparse_rounds", 1) * NUM_CU / n_groups);
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const size_t tmp_bytes = ((sizeof(F) * (size_t)(n_groups * stride) + 255) / 256) * 256;
    void *wsv = nullptr;
    rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_groups * nblk * stride) + 256,
                       &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    if (stage && n_cats <= 8 && slab_rows == 128) {
        auto kpf = n_cats == 1   ? &multi_cat_sparse_pf_kernel<F, 1>
                   : n_cats == 2 ? &multi_cat_sparse_pf_kernel<F, 2>
                   : n_cats == 3 ? &multi_cat_sparse_pf_kernel<F, 3>
                   : n_cats == 4 ? &multi_cat_sparse_pf_kernel<F, 4>
                   : n_cats == 5 ? &multi_cat_sparse_pf_kernel<F, 5>
                   : n_cats == 6 ? &multi_cat_sparse_pf_kernel<F, 6>
                   : n_cats == 7 ? &multi_cat_sparse_pf_kernel<F, 7>
                                 : &multi_cat_sparse_pf_kernel<F, 8>;
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kpf),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(st);
        hipLaunchKernelGGL(kpf, dim3((unsigned)nblk, (unsigned)n_groups), dim3(nw * 64), lds, st, cs, d,
                           vals, koff, ecol, gptr, n_groups, n_slabs, spb, group_cols, n, ws, stride,
                           wg_log_ptr());
        prof_end(st);
        TM_LAUNCH_CHECK();
    } else {
        auto kern = stage ? &multi_cat_sparse_kernel<F, true> : &multi_cat_sparse_kernel<F, false>;
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_groups), dim3(nw * 64), lds, st, cs,
                           d, vals, koff, ecol, gptr, n_groups, n_slabs, spb, slab_rows, group_cols,
                           n, ws, stride);
        prof_end(st);
        TM_LAUNCH_CHECK();
    }
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_groups, tmp,
                                   (int64_t)n_groups * stride, false, st);
    if (rc) return rc;
    // tmp [group][total][group_cols] -> out[total][m]
    hipLaunchKernelGGL((multi_cat_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)),
                       dim3(256), 0, st, tmp, (int64_t)cs.total, m, group_cols, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

// packed[row] = sum_c field_c << (10 c), field_c = tile row of the row's level in categorical c (cs.off[c] + code -
// drop) or 1023 when the row has none there (missing code, dropped first level)
__global__ __launch_bounds__(256) void multi_cat_pack_codes_kernel(CatSet cs, int64_t n, unsigned *__restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        unsigned pk = 0u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            unsigned f = 1023u;
            if (c < cs.n_cats) {
                const int col = cs.codes[c][r] - cs.drop[c];
                if (col >= 0 && col < cs.ncol[c]) f = (unsigned)(cs.off[c] + col);
            }
            pk |= f << (10 * c);
        }
        out[r] = pk;
    }
}

// entry-twin form (see multi_cat_sparse_ent_kernel): out [total levels][mk], mk = 16 * groups kernel columns
template <typename F>
static int run_multi_cat_sparse_ent(const void *const *h_codes, const int64_t *h_ncols, const int32_t *h_drop,
                                    int n_cats, int64_t n, const F *d, const F *vals, const unsigned short *meta,
                                    const unsigned *bstart, int64_t mk, F *out, hipStream_t st,
                                    const unsigned *packed = nullptr, int64_t n_slots = 0) {
    CatSet cs;
    int rc = make_catset(h_codes, h_ncols, h_drop, n_cats, &cs);
    if (rc) return rc;
    const int64_t total = (int64_t)cs.total * mk;
    if (total == 0) return TM_OK;
    if (n == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    TM_REQUIRE(mk % 16 == 0 && n_cats >= 1 && n_cats <= 8, "kernel columns in groups of 16, 1 .. 8 categoricals");
    TM_REQUIRE(n < (1ll << 28), "at most 2^28 - 1 rows per block");
    constexpr int GC = 32;
    const int64_t stride = (int64_t)cs.total * GC;
    const size_t lds = ((sizeof(lds_acc_t) * (size_t)cs.total * (GC + 1) + 15) / 16) * 16;
    if (lds > HIST_LDS_MAX) {
        set_error("multi_cat_sparse_ent: %d stacked categories exceed the LDS tile", cs.total);
        return TM_EUNSUPPORTED;
    }
    const int nw = (int)std::min<int64_t>(16, std::max<int64_t>(4, tune("catsparse_waves", 16)));
    const int n_groups = (int)(mk / 16);
    const int n_pairs = (n_groups + 1) / 2;
    const int64_t n_slabs = ceil_div(n, 64);
    int64_t nblk = std::max<int64_t>(1, tune("catsparse_rounds", 1) * NUM_CU / n_pairs);
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const size_t tmp_bytes = ((sizeof(F) * (size_t)(n_pairs * stride) + 255) / 256) * 256;
    void *wsv = nullptr;
    rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_pairs * nblk * stride) + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    auto kern = n_cats == 1   ? &multi_cat_sparse_ent_kernel<F, 1>
                : n_cats == 2 ? &multi_cat_sparse_ent_kernel<F, 2>
                : n_cats == 3 ? &multi_cat_sparse_ent_kernel<F, 3>
                : n_cats == 4 ? &multi_cat_sparse_ent_kernel<F, 4>
                : n_cats == 5 ? &multi_cat_sparse_ent_kernel<F, 5>
                : n_cats == 6 ? &multi_cat_sparse_ent_kernel<F, 6>
                : n_cats == 7 ? &multi_cat_sparse_ent_kernel<F, 7>
                              : &multi_cat_sparse_ent_kernel<F, 8>;
    TM_REQUIRE(packed == nullptr || (n_cats <= 3 && cs.total < 1023), "packed codes: at most 3 categoricals, 1022 levels");
    // staged form (round 6): d and the code word(s) of a slab's rows parked in per-wave LDS next to the tile
    const int ncw = packed != nullptr ? 1 : n_cats;
    const size_t lds_staged = lds + (size_t)nw * 64 * (sizeof(F) + 4 * (size_t)ncw);
    // ... where a (group, slab) block fills a good part of a 64-lane step: the staged walk takes one step (and one
    // fetch of the rows' operands) per BLOCK, the gather kernel 256 slots per step whatever the blocks hold -- 512
    // columns @ 5 %: 59 slots per block, staged -16 %; 512 @ 4 % (48): -7 %; 512 @ 3 % (38): +10 %; 2048 @ 1.25 % (18):
    // +85 % -- the staged walk costs per block, the gather walk per slot; crossover ~44 (profiles/r6_catsparse.txt).  n_slots = slots of the whole stream (0: unknown -> gather kernel).
    const double fill = (double)n_slots / ((double)std::max(n_groups, 1) * (double)n_slabs);
    if (tune("catsparse_staged", 1) != 0 && lds_staged <= 156 * 1024 &&
        fill >= (double)tune("catsparse_staged_fill", 44)) {
        using KS = void (*)(CatSet, const F *, const F *, const unsigned short *, const unsigned *, int, int64_t, int64_t,
                            int64_t, F *, int64_t, const unsigned *, int);
        KS ks = nullptr;
        if (packed != nullptr) {
            ks = n_cats == 1 ? &multi_cat_sparse_ent_staged_kernel<F, 1, true>
                 : n_cats == 2 ? &multi_cat_sparse_ent_staged_kernel<F, 2, true>
                               : &multi_cat_sparse_ent_staged_kernel<F, 3, true>;
        } else {
            ks = n_cats == 1   ? &multi_cat_sparse_ent_staged_kernel<F, 1, false>
                 : n_cats == 2 ? &multi_cat_sparse_ent_staged_kernel<F, 2, false>
                 : n_cats == 3 ? &multi_cat_sparse_ent_staged_kernel<F, 3, false>
                 : n_cats == 4 ? &multi_cat_sparse_ent_staged_kernel<F, 4, false>
                 : n_cats == 5 ? &multi_cat_sparse_ent_staged_kernel<F, 5, false>
                 : n_cats == 6 ? &multi_cat_sparse_ent_staged_kernel<F, 6, false>
                 : n_cats == 7 ? &multi_cat_sparse_ent_staged_kernel<F, 7, false>
                               : &multi_cat_sparse_ent_staged_kernel<F, 8, false>;
        }
        if (lds_staged > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(ks),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_staged));
        prof_begin(st);
        hipLaunchKernelGGL(ks, dim3((unsigned)nblk, (unsigned)n_pairs), dim3(nw * 64), lds_staged, st, cs, d, vals,
                           meta, bstart, n_groups, n, n_slabs, spb, ws, stride, packed, (int)lds);
        prof_end(st);
        TM_LAUNCH_CHECK();
    } else {
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_pairs), dim3(nw * 64), lds, st, cs, d, vals, meta,
                           bstart, n_groups, n, n_slabs, spb, ws, stride, packed);
        prof_end(st);
        TM_LAUNCH_CHECK();
    }
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_pairs, tmp, (int64_t)n_pairs * stride, false, st);
    if (rc) return rc;
    // tmp [pair][total][32] -> out[total][mk]
    hipLaunchKernelGGL((multi_cat_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, tmp,
                       (int64_t)cs.total, mk, GC, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

using namespace tmh;

#define TM_CHECK_COMMON(nn) \
    TM_REQUIRE((nn) >= 0, "negative size")

extern "C" {

int tm_cat_transpose_matvec_f32(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                                const float *v, const int32_t *rows, int64_t n_rows,
                                const int32_t *cols, int64_t n_cols_sel, float *out, void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_tmv<float>(codes, n, n_cols, drop_first, v, rows, n_rows, cols, n_cols_sel, out,
                              as_stream(stream));
}
int tm_cat_transpose_matvec_f64(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                                const double *v, const int32_t *rows, int64_t n_rows,
                                const int32_t *cols, int64_t n_cols_sel, double *out,
                                void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_tmv<double>(codes, n, n_cols, drop_first, v, rows, n_rows, cols, n_cols_sel,
                               out, as_stream(stream));
}

int tm_cat_matvec_f32(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                      const float *v, const int32_t *cols, int64_t n_cols_sel, float *out,
                      void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_matvec<float>(codes, n, n_cols, drop_first, v, cols, n_cols_sel, out,
                                 as_stream(stream));
}
int tm_cat_matvec_f64(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                      const double *v, const int32_t *cols, int64_t n_cols_sel, double *out,
                      void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_matvec<double>(codes, n, n_cols, drop_first, v, cols, n_cols_sel, out,
                                  as_stream(stream));
}

int tm_cat_matvec_assign_f32(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                             const float *v, const int32_t *cols, int64_t n_cols_sel, float *out,
                             void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_matvec<float>(codes, n, n_cols, drop_first, v, cols, n_cols_sel, out,
                                 as_stream(stream), true);
}
int tm_cat_matvec_assign_f64(const int32_t *codes, int64_t n, int64_t n_cols, int drop_first,
                             const double *v, const int32_t *cols, int64_t n_cols_sel, double *out,
                             void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_matvec<double>(codes, n, n_cols, drop_first, v, cols, n_cols_sel, out,
                                  as_stream(stream), true);
}

int tm_cat_cat_sandwich_f32(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                            const float *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                            int64_t j_ncol, int i_drop_first, int j_drop_first, float *out,
                            void *stream) {
    TM_CHECK_COMMON(n);
    return run_hist<float, true>(i_codes, j_codes, d, rows, rows ? n_rows : n, i_drop_first,
                                 j_drop_first, i_ncol, j_ncol, nullptr, out, false,
                                 as_stream(stream));
}
int tm_cat_cat_sandwich_f64(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                            const double *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                            int64_t j_ncol, int i_drop_first, int j_drop_first, double *out,
                            void *stream) {
    TM_CHECK_COMMON(n);
    return run_hist<double, true>(i_codes, j_codes, d, rows, rows ? n_rows : n, i_drop_first,
                                  j_drop_first, i_ncol, j_ncol, nullptr, out, false,
                                  as_stream(stream));
}

// (the caller vouches that no cell of the table collects more than a few thousand rows: global atomics
// on one address serialise)
int tm_cat_cat_sandwich_atomic_f32(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                                   const float *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                                   int64_t j_ncol, int i_drop_first, int j_drop_first, float *out,
                                   void *stream) {
    TM_CHECK_COMMON(n);
    return run_hist<float, true>(i_codes, j_codes, d, rows, rows ? n_rows : n, i_drop_first,
                                 j_drop_first, i_ncol, j_ncol, nullptr, out, false,
                                 as_stream(stream), 12);
}
int tm_cat_cat_sandwich_atomic_f64(const int32_t *i_codes, const int32_t *j_codes, int64_t n,
                                   const double *d, const int32_t *rows, int64_t n_rows, int64_t i_ncol,
                                   int64_t j_ncol, int i_drop_first, int j_drop_first, double *out,
                                   void *stream) {
    TM_CHECK_COMMON(n);
    return run_hist<double, true>(i_codes, j_codes, d, rows, rows ? n_rows : n, i_drop_first,
                                  j_drop_first, i_ncol, j_ncol, nullptr, out, false,
                                  as_stream(stream), 12);
}

int tm_cat_cat_sandwich_sorted_f32(const int32_t *ci_sorted, const int32_t *cj_sorted, const int32_t *perm,
                                   const int64_t *lptr, int64_t n_sorted, const float *d, int64_t i_ncol,
                                   int64_t j_ncol, float *out, void *stream) {
    TM_CHECK_COMMON(n_sorted);
    return run_hist_sorted<float>(ci_sorted, cj_sorted, perm, lptr, n_sorted, d, i_ncol, j_ncol, out,
                                  as_stream(stream));
}
int tm_cat_cat_sandwich_sorted_f64(const int32_t *ci_sorted, const int32_t *cj_sorted, const int32_t *perm,
                                   const int64_t *lptr, int64_t n_sorted, const double *d, int64_t i_ncol,
                                   int64_t j_ncol, double *out, void *stream) {
    TM_CHECK_COMMON(n_sorted);
    return run_hist_sorted<double>(ci_sorted, cj_sorted, perm, lptr, n_sorted, d, i_ncol, j_ncol, out,
                                   as_stream(stream));
}

int tm_cat_dense_sandwich_f32(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                              const float *d, const int32_t *rows, int64_t n_rows, const float *M,
                              int64_t M_ncol, int order_f, const int32_t *j_cols, int64_t n_j,
                              float *out, void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_dense<float>(codes, n, i_ncol, drop_first, d, rows, n_rows, M, M_ncol, order_f,
                                j_cols, j_cols ? n_j : M_ncol, out, as_stream(stream));
}
int tm_cat_dense_sandwich_f64(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                              const double *d, const int32_t *rows, int64_t n_rows,
                              const double *M, int64_t M_ncol, int order_f, const int32_t *j_cols,
                              int64_t n_j, double *out, void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_dense<double>(codes, n, i_ncol, drop_first, d, rows, n_rows, M, M_ncol, order_f,
                                 j_cols, j_cols ? n_j : M_ncol, out, as_stream(stream));
}

int tm_cat_sparse_sandwich_f32(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                               const float *csr_data, const int32_t *csr_indices,
                               const int64_t *csr_indptr, int64_t s_ncol, const float *d,
                               const int32_t *rows, int64_t n_rows, const int32_t *cols,
                               int64_t n_cols, float *out, void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_sparse<float>(codes, n, i_ncol, drop_first, csr_data, csr_indices, csr_indptr,
                                 s_ncol, d, rows, n_rows, cols, n_cols, out, as_stream(stream));
}
int tm_cat_sparse_sandwich_f64(const int32_t *codes, int64_t n, int64_t i_ncol, int drop_first,
                               const double *csr_data, const int32_t *csr_indices,
                               const int64_t *csr_indptr, int64_t s_ncol, const double *d,
                               const int32_t *rows, int64_t n_rows, const int32_t *cols,
                               int64_t n_cols, double *out, void *stream) {
    TM_CHECK_COMMON(n);
    return run_cat_sparse<double>(codes, n, i_ncol, drop_first, csr_data, csr_indices, csr_indptr,
                                  s_ncol, d, rows, n_rows, cols, n_cols, out, as_stream(stream));
}

int tm_multi_cat_dense_sandwich_f32(const void *const *h_codes, const int64_t *h_ncols,
                                    const int32_t *h_drop_first, int n_cats, int64_t n,
                                    const float *d, const float *M, int64_t M_ncol, int order_f,
                                    float *out, void *stream) {
    return run_multi_cat_dense<float>(h_codes, h_ncols, h_drop_first, n_cats, n, d, M, M_ncol,
                                      order_f, out, as_stream(stream));
}
int tm_multi_cat_dense_sandwich_f64(const void *const *h_codes, const int64_t *h_ncols,
                                    const int32_t *h_drop_first, int n_cats, int64_t n,
                                    const double *d, const double *M, int64_t M_ncol, int order_f,
                                    double *out, void *stream) {
    return run_multi_cat_dense<double>(h_codes, h_ncols, h_drop_first, n_cats, n, d, M, M_ncol,
                                       order_f, out, as_stream(stream));
}
int tm_multi_cat_dense_sandwich_rows_f32(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n,
                                         const float *d, const float *M, int64_t M_ncol,
                                         const int32_t *rows, int64_t n_rows, float *out,
                                         void *stream) {
    return run_multi_cat_dense<float>(h_codes, h_ncols, h_drop_first, n_cats, n, d, M, M_ncol, 0, out,
                                      as_stream(stream), rows, n_rows);
}
int tm_multi_cat_dense_sandwich_rows_f64(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n,
                                         const double *d, const double *M, int64_t M_ncol,
                                         const int32_t *rows, int64_t n_rows, double *out,
                                         void *stream) {
    return run_multi_cat_dense<double>(h_codes, h_ncols, h_drop_first, n_cats, n, d, M, M_ncol, 0, out,
                                       as_stream(stream), rows, n_rows);
}
int tm_multi_cat_sparse_sandwich_ent_f32(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n, const float *d,
                                         const float *vals, const uint16_t *meta, const uint32_t *bstart,
                                         int64_t n_slots, int64_t mk, float *out, void *stream) {
    return run_multi_cat_sparse_ent<float>(h_codes, h_ncols, h_drop_first, n_cats, n, d, vals, meta, bstart, mk,
                                           out, as_stream(stream), nullptr, n_slots);
}
int tm_multi_cat_sparse_sandwich_ent_f64(const void *const *h_codes, const int64_t *h_ncols,
                                         const int32_t *h_drop_first, int n_cats, int64_t n, const double *d,
                                         const double *vals, const uint16_t *meta, const uint32_t *bstart,
                                         int64_t n_slots, int64_t mk, double *out, void *stream) {
    return run_multi_cat_sparse_ent<double>(h_codes, h_ncols, h_drop_first, n_cats, n, d, vals, meta, bstart, mk,
                                            out, as_stream(stream), nullptr, n_slots);
}
int tm_multi_cat_pack_codes(const void *const *h_codes, const int64_t *h_ncols, const int32_t *h_drop_first, int n_cats,
                            int64_t n, uint32_t *packed, void *stream) {
    CatSet cs;
    int rc = make_catset(h_codes, h_ncols, h_drop_first, n_cats, &cs);
    if (rc) return rc;
    TM_REQUIRE(n_cats <= 3 && cs.total < 1023, "packed codes: at most 3 categoricals, 1022 stacked levels");
    if (n == 0) return TM_OK;
    hipLaunchKernelGGL(multi_cat_pack_codes_kernel, dim3((unsigned)std::min<int64_t>(4 * NUM_CU, ceil_div(n, 256))),
                       dim3(256), 0, as_stream(stream), cs, n, packed);
    TM_LAUNCH_CHECK();
    return TM_OK;
}
int tm_multi_cat_sparse_sandwich_entp_f32(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n, const float *d,
                                          const float *vals, const uint16_t *meta, const uint32_t *bstart,
                                          int64_t n_slots, int64_t mk, const uint32_t *packed, float *out, void *stream) {
    return run_multi_cat_sparse_ent<float>(h_codes, h_ncols, h_drop_first, n_cats, n, d, vals, meta, bstart, mk, out,
                                           as_stream(stream), packed, n_slots);
}
int tm_multi_cat_sparse_sandwich_entp_f64(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n, const double *d,
                                          const double *vals, const uint16_t *meta, const uint32_t *bstart,
                                          int64_t n_slots, int64_t mk, const uint32_t *packed, double *out, void *stream) {
    return run_multi_cat_sparse_ent<double>(h_codes, h_ncols, h_drop_first, n_cats, n, d, vals, meta, bstart, mk,
                                            out, as_stream(stream), packed, n_slots);
}
int tm_multi_cat_sparse_sandwich_slab_f32(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n,
                                          const float *d, const float *vals, const uint32_t *koff,
                                          const uint8_t *ecol, const int64_t *gptr, int64_t m,
                                          float *out, void *stream) {
    return run_multi_cat_sparse<float>(h_codes, h_ncols, h_drop_first, n_cats, n, d, vals, koff,
                                       ecol, gptr, m, tm_slab_rows(), tm_slab_group_cols(), out,
                                       as_stream(stream));
}
int tm_multi_cat_sparse_sandwich_slab_f64(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats, int64_t n,
                                          const double *d, const double *vals, const uint32_t *koff,
                                          const uint8_t *ecol, const int64_t *gptr, int64_t m,
                                          double *out, void *stream) {
    return run_multi_cat_sparse<double>(h_codes, h_ncols, h_drop_first, n_cats, n, d, vals, koff,
                                        ecol, gptr, m, tm_slab_rows(), tm_slab_group_cols(), out,
                                        as_stream(stream));
}

}  // extern "C"

namespace tmh {
template <typename F, typename IX>
static int run_multi_cat_sparse_rows(const void *const *h_codes, const int64_t *h_ncols,
                                     const int32_t *h_drop, int n_cats, const F *cm_data,
                                     const IX *cm_ind, const int32_t *ranges, const int32_t *rows,
                                     int64_t n_sel, int64_t m, const F *d_sel, F *out, hipStream_t st) {
    CatSet cs;
    int rc = make_catset(h_codes, h_ncols, h_drop, n_cats, &cs);
    if (rc) return rc;
    const int64_t total = (int64_t)cs.total * m;
    if (total == 0) return TM_OK;
    if (n_sel == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    constexpr int GC = 32;
    const size_t lds = ((sizeof(lds_acc_t) * (size_t)cs.total * (GC + 1) + 15) / 16) * 16;
    if (lds > HIST_LDS_MAX) {
        set_error("multi_cat_sparse_rows: %d stacked categories exceed the LDS tile", cs.total);
        return TM_EUNSUPPORTED;
    }
    const int n_groups = (int)ceil_div(m, GC);
    const int64_t stride = (int64_t)cs.total * GC;
    int64_t nblk = std::max<int64_t>(1, tune("catsparse_rounds", 1) * NUM_CU / n_groups);
    nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, ceil_div(n_sel, 2048)));
    const int64_t rpb = ceil_div(n_sel, nblk);
    nblk = ceil_div(n_sel, rpb);
    const size_t tmp_bytes = ((sizeof(F) * (size_t)(n_groups * stride) + 255) / 256) * 256;
    void *wsv = nullptr;
    rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_groups * nblk * stride) + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    auto kern = &multi_cat_sparse_rows_kernel<F, IX>;
    if (lds > 48 * 1024)
        TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_groups), dim3(1024), lds, st, cs, cm_data,
                       cm_ind, ranges, rows, d_sel, n_sel, rpb, ws, stride);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_groups, tmp, (int64_t)n_groups * stride, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((multi_cat_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       st, tmp, (int64_t)cs.total, m, GC, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}
}  // namespace tmh

extern "C" {
int tm_multi_cat_sparse_sandwich_rows_f32(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats,
                                          const float *cm_data, const int32_t *cm_indices,
                                          const int32_t *row_ranges, const int32_t *rows,
                                          int64_t n_sel, int64_t m, const float *d_sel, float *out,
                                          void *stream) {
    return tmh::run_multi_cat_sparse_rows<float, int32_t>(h_codes, h_ncols, h_drop_first, n_cats, cm_data,
                                                 cm_indices, row_ranges, rows, n_sel, m, d_sel, out,
                                                 tmh::as_stream(stream));
}
int tm_multi_cat_sparse_sandwich_rows_f64(const void *const *h_codes, const int64_t *h_ncols,
                                          const int32_t *h_drop_first, int n_cats,
                                          const double *cm_data, const int32_t *cm_indices,
                                          const int32_t *row_ranges, const int32_t *rows,
                                          int64_t n_sel, int64_t m, const double *d_sel, double *out,
                                          void *stream) {
    return tmh::run_multi_cat_sparse_rows<double, int32_t>(h_codes, h_ncols, h_drop_first, n_cats, cm_data,
                                                  cm_indices, row_ranges, rows, n_sel, m, d_sel, out,
                                                  tmh::as_stream(stream));
}
int tm_multi_cat_sparse_sandwich_rows_u8_f32(const void *const *h_codes, const int64_t *h_ncols,
                                             const int32_t *h_drop_first, int n_cats,
                                             const float *cm_data, const uint8_t *cm_col8,
                                             const int32_t *row_ranges, const int32_t *rows,
                                             int64_t n_sel, int64_t m, const float *d_sel, float *out,
                                             void *stream) {
    return tmh::run_multi_cat_sparse_rows<float, uint8_t>(h_codes, h_ncols, h_drop_first, n_cats, cm_data,
                                                          cm_col8, row_ranges, rows, n_sel, m, d_sel, out,
                                                          tmh::as_stream(stream));
}
int tm_multi_cat_sparse_sandwich_rows_u8_f64(const void *const *h_codes, const int64_t *h_ncols,
                                             const int32_t *h_drop_first, int n_cats,
                                             const double *cm_data, const uint8_t *cm_col8,
                                             const int32_t *row_ranges, const int32_t *rows,
                                             int64_t n_sel, int64_t m, const double *d_sel, double *out,
                                             void *stream) {
    return tmh::run_multi_cat_sparse_rows<double, uint8_t>(h_codes, h_ncols, h_drop_first, n_cats, cm_data,
                                                           cm_col8, row_ranges, rows, n_sel, m, d_sel, out,
                                                           tmh::as_stream(stream));
}
}  // extern "C"
