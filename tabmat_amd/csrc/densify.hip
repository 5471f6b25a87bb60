// A narrow column selection of a SplitMatrix sandwich (`cols=`: a solver's active set) as ONE
// row-major dense block: the selected columns of the dense blocks are gathered, the selected
// columns of the sparse blocks are written out densely, and the tuned unrestricted kernels (MFMA
// syrk, fused categorical x dense) run on the result -- the reference's restricted loops
// (ext/dense.pyx:24-54, ext/sparse.pyx:17-77 and 211-260 with `cols`) cost in proportion to the
// selection, which the generic restricted kernels here do not (they stream everything and drop).
#include "common.hpp"

namespace tmh {

// T[r, colmap[c]] += value for every stored entry (r, c) whose column is selected (colmap >= 0).
// 8 lanes per row.
template <typename F>
__global__ __launch_bounds__(256) void csr_densify_cols_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ indices,
    const int64_t *__restrict__ indptr, int64_t n, const int32_t *__restrict__ colmap,
    F *__restrict__ T, int64_t ld) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    if (r >= n) return;
    const int g = threadIdx.x & 7;
    const int64_t e1 = indptr[r + 1];
    for (int64_t e = indptr[r] + g; e < e1; e += 8) {
        const int t = colmap[indices[e]];
        if (t >= 0) atomic_add(T + r * ld + t, data[e]);     // (duplicate entries add up)
    }
}

// The same from the CSC form (rows / values sorted by column, CsrDev.csc_blocks): only the entries
// of the selected columns are touched.  seg[q] = {first entry, end} of selected column q, tcol[q]
// its column of T.
template <typename F>
__global__ __launch_bounds__(256) void csc_densify_cols_kernel(
    const int32_t *__restrict__ rows, const F *__restrict__ vals, const int64_t *__restrict__ seg,
    const int32_t *__restrict__ tcol, F *__restrict__ T, int64_t ld) {
    const int64_t e0 = seg[2 * blockIdx.y], e1 = seg[2 * blockIdx.y + 1];
    const int t = tcol[blockIdx.y];
    for (int64_t e = e0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < e1;
         e += (int64_t)gridDim.x * blockDim.x)
        atomic_add(T + (int64_t)rows[e] * ld + t, vals[e]);
}

// T[r, t0 + q] = X[r, cols[q]] for a C- or F-ordered dense block (16 rows x 16 selected columns
// per 256-thread block)
template <typename F>
__global__ __launch_bounds__(256) void dense_gather_cols_kernel(
    const F *__restrict__ X, int64_t n, int64_t m, int order_f, const int32_t *__restrict__ cols,
    int64_t n_sel, F *__restrict__ T, int64_t ld, int64_t t0) {
    const int64_t q = (int64_t)blockIdx.y * 16 + (threadIdx.x & 15);
    const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    if (q >= n_sel || r >= n) return;
    const int64_t c = cols[q];
    T[r * ld + t0 + q] = order_f ? X[c * n + r] : X[r * m + c];
}

// F-ordered source: 64 consecutive rows of a column are one 512-byte read; the 64 x 32 tile is
// turned in LDS so that T is written in 32-element row segments.
template <typename F>
__global__ __launch_bounds__(256) void dense_gather_cols_f_kernel(
    const F *__restrict__ X, int64_t n, const int32_t *__restrict__ cols, int64_t n_sel,
    F *__restrict__ T, int64_t ld, int64_t t0) {
    __shared__ F tile[64][33];
    const int64_t r0 = (int64_t)blockIdx.x * 64, q0 = (int64_t)blockIdx.y * 32;
    const int rl = threadIdx.x & 63;
    for (int qq = threadIdx.x >> 6; qq < 32; qq += 4) {
        F v = F(0);
        if (q0 + qq < n_sel && r0 + rl < n) v = X[(int64_t)cols[q0 + qq] * n + r0 + rl];
        tile[rl][qq] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 32; e += 256) {
        const int rr = e >> 5, qq = e & 31;
        if (r0 + rr < n && q0 + qq < n_sel) T[(r0 + rr) * ld + t0 + q0 + qq] = tile[rr][qq];
    }
}

}  // namespace tmh

using namespace tmh;

// int64 -> int32 indices on the device (bad[0] |= 1 when a value lies outside [0, limit)); int32 -> int64
__global__ void index_narrow_kernel(const int64_t *__restrict__ src, int64_t count, int64_t limit,
                                    int32_t *__restrict__ dst, int32_t *__restrict__ bad) {
    bool off = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = src[i];
        off |= v < 0 || v >= limit;
        dst[i] = (int32_t)v;
    }
    if (__any(off) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}
__global__ void index_widen_kernel(const int32_t *__restrict__ src, int64_t count, int64_t *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = (int64_t)src[i];
}

extern "C" {
int tm_index_narrow_i64(const int64_t *src, int64_t count, int64_t limit, int32_t *dst, int32_t *bad, void *stream) {
    using namespace tmh;
    if (count <= 0) return TM_OK;
    TM_REQUIRE(limit >= 0 && limit <= (1ll << 31), "index limit beyond int32");
    hipLaunchKernelGGL(index_narrow_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(count, 256), 4096)), dim3(256), 0,
                       as_stream(stream), src, count, limit, dst, bad);
    TM_LAUNCH_CHECK();
    return TM_OK;
}
int tm_index_widen_i32(const int32_t *src, int64_t count, int64_t *dst, void *stream) {
    using namespace tmh;
    if (count <= 0) return TM_OK;
    hipLaunchKernelGGL(index_widen_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(count, 256), 4096)), dim3(256), 0,
                       as_stream(stream), src, count, dst);
    TM_LAUNCH_CHECK();
    return TM_OK;
}
}  // extern "C"

extern "C" {
#define TM_DENSIFY(SUF, F)                                                                              \
    int tm_csr_densify_cols_##SUF(const F *data, const int32_t *indices, const int64_t *indptr,         \
                                  int64_t n, const int32_t *colmap, F *T, int64_t ld, void *stream) {   \
        if (n <= 0) return TM_OK;                                                                       \
        hipStream_t st = as_stream(stream);                                                             \
        hipLaunchKernelGGL((csr_densify_cols_kernel<F>), dim3((unsigned)ceil_div(n * 8, 256)),          \
                           dim3(256), 0, st, data, indices, indptr, n, colmap, T, ld);                  \
        TM_LAUNCH_CHECK();                                                                              \
        return TM_OK;                                                                                   \
    }                                                                                                   \
    int tm_csc_densify_cols_##SUF(const int32_t *rows, const F *vals, const int64_t *seg,               \
                                  const int32_t *tcol, int64_t n_sel, int64_t max_len, F *T,            \
                                  int64_t ld, void *stream) {                                           \
        if (n_sel <= 0 || max_len <= 0) return TM_OK;                                                   \
        hipStream_t st = as_stream(stream);                                                             \
        const int64_t gx = std::max<int64_t>(1, std::min<int64_t>(ceil_div(max_len, 1024), 1024));      \
        hipLaunchKernelGGL((csc_densify_cols_kernel<F>), dim3((unsigned)gx, (unsigned)n_sel),           \
                           dim3(256), 0, st, rows, vals, seg, tcol, T, ld);                             \
        TM_LAUNCH_CHECK();                                                                              \
        return TM_OK;                                                                                   \
    }                                                                                                   \
    int tm_dense_gather_cols_##SUF(const F *X, int64_t n, int64_t m, int order_f, const int32_t *cols,  \
                                   int64_t n_sel, F *T, int64_t ld, int64_t t0, void *stream) {         \
        if (n <= 0 || n_sel <= 0) return TM_OK;                                                         \
        hipStream_t st = as_stream(stream);                                                             \
        if (order_f)                                                                                    \
            hipLaunchKernelGGL((dense_gather_cols_f_kernel<F>),                                         \
                               dim3((unsigned)ceil_div(n, 64), (unsigned)ceil_div(n_sel, 32)),          \
                               dim3(256), 0, st, X, n, cols, n_sel, T, ld, t0);                         \
        else                                                                                            \
            hipLaunchKernelGGL((dense_gather_cols_kernel<F>),                                           \
                               dim3((unsigned)ceil_div(n, 16), (unsigned)ceil_div(n_sel, 16)),          \
                               dim3(256), 0, st, X, n, m, order_f, cols, n_sel, T, ld, t0);             \
        TM_LAUNCH_CHECK();                                                                              \
        return TM_OK;                                                                                   \
    }
TM_DENSIFY(f32, float)
TM_DENSIFY(f64, double)
#undef TM_DENSIFY
}  // extern "C"
