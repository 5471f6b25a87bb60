// Categorical cross terms for categoricals with MANY levels (reference: ext/split.pyx:32-80
// sandwich_cat_dense -> ext/cat_split_helpers-tmpl.cpp:97-151, and the scipy product behind
// CategoricalMatrix._cross_sparse, categorical_matrix.py:825-838).
//
// The LDS-tile kernels of cat.hip hold [levels][columns] in LDS; with thousands of levels the
// tile no longer fits and they fall back to passes per level range or to the one-hot gather
// (4.7 ms for 2M rows x 128 columns against 10 500 levels, 5.4 ms per pass for cat x sparse).
// Here the cost does not depend on the number of levels: the rows are grouped by level ONCE
// (CategoricalMatrix._det_plan: a stable device sort -> `perm`; every level's run cut into blocks
// of at most tm_cat_det_block_rows() rows -> `bstart`; the blocks of level c are
// cat_bptr[c] .. cat_bptr[c + 1]), a workgroup sums  d[k] * Y[k, :]  over the rows of ONE block
// -- all of one level, so the accumulator is a single row: registers for a dense Y, an LDS row of
// doubles for a sparse Y -- and a second kernel adds the blocks of each level in a fixed order
// (run-to-run reproducible, like the reference's thread-owned partial sums).
// Rows with d == 0 are skipped without touching Y (a row the caller excluded may hold inf / NaN).
#include "common.hpp"

namespace tmh {

constexpr int CS_WAVES = 4;

// ---- dense Y (C-ordered, 16-byte aligned rows): lane <-> VEC adjacent columns of a 64 * VEC chunk
// `val` != NULL: entry i additionally carries a value (a SPARSE column instead of a one-hot level:
// the column-sorted form of the sparse x dense term, see tm_csc_dense_sandwich_sorted_*).
template <typename F>
__global__ __launch_bounds__(CS_WAVES * 64) void cat_dense_sorted_kernel(
    const int32_t *__restrict__ perm, const int64_t *__restrict__ bstart, const F *__restrict__ d,
    const F *__restrict__ val, const F *__restrict__ Y, int64_t ld, int m,
    double *__restrict__ partial) {
    constexpr int VEC = 16 / (int)sizeof(F);
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ double red[CS_WAVES][64 * VEC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    const int j = (blockIdx.y * 64 + lane) * VEC;          // first column of this lane
    const int jc = min(j, m - VEC);
    const int64_t i0 = bstart[b], i1 = bstart[b + 1];
    double acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = 0.0;
    // 4 rows per wave and step: their (dependent) loads perm -> d, Y row are issued together
    for (int64_t i = i0 + wave * 4; i < i1; i += CS_WAVES * 4) {
        int64_t k[4];
        F dk[4];
        vec_t y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = perm[min(i + u, i1 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) dk[u] = i + u < i1 ? d[k[u]] : F(0);
        if (val != nullptr) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (dk[u] != F(0)) dk[u] *= val[min(i + u, i1 - 1)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) y[u][v] = F(0);
            if (dk[u] != F(0))
                y[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(Y + k[u] * ld + jc));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (dk[u] != F(0)) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) acc[v] += (double)dk[u] * (double)y[u][v];
            }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) red[wave][lane * VEC + v] = acc[v];
    __syncthreads();
    if (wave == 0 && j < m) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const int c = lane * VEC + v;
            partial[b * m + j + v] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        }
    }
}

// ---- sparse Y (CSR): an LDS row of doubles over the output columns [c0, c0 + mc)
template <typename F>
__global__ __launch_bounds__(CS_WAVES * 64) void cat_sparse_sorted_kernel(
    const int32_t *__restrict__ perm, const int64_t *__restrict__ bstart, const F *__restrict__ d,
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    int m, int mc, double *__restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *acc = reinterpret_cast<double *>(smem_raw);          // [mc]
    const int64_t b = blockIdx.x;
    const int c0 = blockIdx.y * mc;
    const int c1 = min(c0 + mc, m);
    for (int c = threadIdx.x; c < mc; c += blockDim.x) acc[c] = 0.0;
    __syncthreads();
    const int64_t i0 = bstart[b], i1 = bstart[b + 1];
    // 32 lanes per row, two rows per wave at a time.  The chain perm -> {d, indptr} -> entries is
    // three dependent memory round trips: each half-wave resolves the first two for 32 of its rows
    // AT ONCE (lane <-> row) and then walks those rows with the values broadcast from the lanes.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 5, sl = lane & 31;
    constexpr int NH = CS_WAVES * 2;                       // half-waves of the workgroup
    const int h = wave * 2 + sub;
    for (int64_t base = i0 + h; base < i1; base += (int64_t)NH * 32) {
        const int64_t i = base + (int64_t)sl * NH;         // this lane's row of the batch
        int64_t k = 0, p0 = 0, p1 = 0;
        F dk = F(0);
        if (i < i1) {
            k = perm[i];
            dk = d[k];
            p0 = ptr[k];
            p1 = ptr[k + 1];
        }
        if (dk == F(0)) p1 = p0;                           // d == 0: the row is not read
        const int nrow = (int)min((int64_t)32, (i1 - base + NH - 1) / NH);
        for (int t = 0; t < nrow; ++t) {
            const int src = sub * 32 + t;
            const int64_t q0 = __shfl(p0, src, 64), q1 = __shfl(p1, src, 64);
            const double dd = (double)__shfl(dk, src, 64);
            for (int64_t p = q0 + sl; p < q1; p += 32) {
                const int c = ind[p];
                if (c >= c0 && c < c1) atomicAdd(&acc[c - c0], dd * (double)data[p]);
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < c1 - c0; c += blockDim.x) partial[b * m + c0 + c] = acc[c];
}

// out[c][j] = sum of the partial rows of level c's blocks, in block order
template <typename F>
__global__ __launch_bounds__(256) void cat_sorted_final_kernel(const double *__restrict__ partial,
                                                               const int64_t *__restrict__ cat_bptr,
                                                               int m, F *__restrict__ out) {
    const int64_t c = blockIdx.x;
    const int64_t b0 = cat_bptr[c], b1 = cat_bptr[c + 1];
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        double s = 0.0;
        for (int64_t b = b0; b < b1; ++b) s += partial[b * m + j];
        out[c * m + j] = (F)s;
    }
}

template <typename F>
static int run_cat_dense_sorted(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                const int64_t *cat_bptr, int64_t n_cols, const F *d, const F *val,
                                const F *Y, int64_t m, F *out, hipStream_t st) {
    if (n_cols == 0 || m == 0) return TM_OK;
    constexpr int VEC = 16 / (int)sizeof(F);
    if ((reinterpret_cast<uintptr_t>(Y) & 15) != 0 || m % VEC != 0 || m < VEC) {
        set_error("tm_cat_dense_sandwich_sorted: Y must be C-ordered with 16-byte aligned rows");
        return TM_EUNSUPPORTED;
    }
    TM_REQUIRE(m < (1 << 24) && n_blocks < (1ll << 31), "cat_dense_sorted: operand too large");
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(double) * (size_t)(std::max<int64_t>(n_blocks, 1) * m) + 256, &wsv, st);
    if (rc) return rc;
    double *partial = reinterpret_cast<double *>(wsv);
    if (n_blocks > 0) {
        prof_begin(st);
        hipLaunchKernelGGL((cat_dense_sorted_kernel<F>),
                           dim3((unsigned)n_blocks, (unsigned)ceil_div(m, 64 * VEC)), dim3(CS_WAVES * 64),
                           0, st, perm, bstart, d, val, Y, m, (int)m, partial);
        prof_end(st);
        TM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((cat_sorted_final_kernel<F>), dim3((unsigned)n_cols), dim3(256), 0, st, partial,
                       cat_bptr, (int)m, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_cat_sparse_sorted(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                 const int64_t *cat_bptr, int64_t n_cols, const F *d, const F *data,
                                 const int32_t *ind, const int64_t *ptr, int64_t m, F *out,
                                 hipStream_t st) {
    if (n_cols == 0 || m == 0) return TM_OK;
    TM_REQUIRE(m < (1 << 24) && n_blocks < (1ll << 31), "cat_sparse_sorted: operand too large");
    const int mc = (int)std::min<int64_t>(m, 8192);          // 64 KB of doubles per pass
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(double) * (size_t)(std::max<int64_t>(n_blocks, 1) * m) + 256, &wsv, st);
    if (rc) return rc;
    double *partial = reinterpret_cast<double *>(wsv);
    if (n_blocks > 0) {
        const size_t lds = sizeof(double) * (size_t)mc;
        auto kern = &cat_sparse_sorted_kernel<F>;
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)n_blocks, (unsigned)ceil_div(m, mc)), dim3(CS_WAVES * 64),
                           lds, st, perm, bstart, d, data, ind, ptr, (int)m, mc, partial);
        prof_end(st);
        TM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((cat_sorted_final_kernel<F>), dim3((unsigned)n_cols), dim3(256), 0, st, partial,
                       cat_bptr, (int)m, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {
int tm_cat_dense_sandwich_sorted_f32(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                     const int64_t *cat_bptr, int64_t n_cols, const float *d,
                                     const float *Y, int64_t m, float *out, void *stream) {
    return tmh::run_cat_dense_sorted<float>(perm, bstart, n_blocks, cat_bptr, n_cols, d, nullptr, Y, m,
                                            out, tmh::as_stream(stream));
}
int tm_cat_dense_sandwich_sorted_f64(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                     const int64_t *cat_bptr, int64_t n_cols, const double *d,
                                     const double *Y, int64_t m, double *out, void *stream) {
    return tmh::run_cat_dense_sorted<double>(perm, bstart, n_blocks, cat_bptr, n_cols, d, nullptr, Y, m,
                                             out, tmh::as_stream(stream));
}
int tm_csc_dense_sandwich_sorted_f32(const int32_t *rows, const float *vals, const int64_t *bstart,
                                     int64_t n_blocks, const int64_t *col_bptr, int64_t n_cols,
                                     const float *d, const float *Y, int64_t m, float *out,
                                     void *stream) {
    return tmh::run_cat_dense_sorted<float>(rows, bstart, n_blocks, col_bptr, n_cols, d, vals, Y, m, out,
                                            tmh::as_stream(stream));
}
int tm_csc_dense_sandwich_sorted_f64(const int32_t *rows, const double *vals, const int64_t *bstart,
                                     int64_t n_blocks, const int64_t *col_bptr, int64_t n_cols,
                                     const double *d, const double *Y, int64_t m, double *out,
                                     void *stream) {
    return tmh::run_cat_dense_sorted<double>(rows, bstart, n_blocks, col_bptr, n_cols, d, vals, Y, m, out,
                                             tmh::as_stream(stream));
}
int tm_cat_sparse_sandwich_sorted_f32(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                      const int64_t *cat_bptr, int64_t n_cols, const float *d,
                                      const float *csr_data, const int32_t *csr_indices,
                                      const int64_t *csr_indptr, int64_t m, float *out, void *stream) {
    return tmh::run_cat_sparse_sorted<float>(perm, bstart, n_blocks, cat_bptr, n_cols, d, csr_data,
                                             csr_indices, csr_indptr, m, out, tmh::as_stream(stream));
}
int tm_cat_sparse_sandwich_sorted_f64(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                      const int64_t *cat_bptr, int64_t n_cols, const double *d,
                                      const double *csr_data, const int32_t *csr_indices,
                                      const int64_t *csr_indptr, int64_t m, double *out, void *stream) {
    return tmh::run_cat_sparse_sorted<double>(perm, bstart, n_blocks, cat_bptr, n_cols, d, csr_data,
                                              csr_indices, csr_indptr, m, out, tmh::as_stream(stream));
}
}  // extern "C"
