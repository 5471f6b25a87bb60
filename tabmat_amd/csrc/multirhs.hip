// Multi-right-hand-side matvec / transpose-matvec of dense and sparse blocks (2-D `vec`):
// the reference hands these to scipy.sparse / NumPy BLAS (sparse_matrix.py:252-268,
// dense_matrix.py:212-217 with a 2-D operand).  V is (m, K) resp. (n, K) row-major, the result
// (n_rows, K) resp. (n_cols, K) row-major and ACCUMULATED into (like the 1-D entry points).
// Mapping: lane <-> right-hand side q (coalesced reads of V rows), 256 / KP rows or columns per
// workgroup; K > 64 is processed in passes of 64.  HBM-bound streaming kernels; the transpose
// forms accumulate with hardware f64 / f32 global atomics into the small (n_cols, K) result.
#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

__device__ __forceinline__ int kp_of(int K) { return K; }

template <typename F>
__global__ __launch_bounds__(256) void csr_matvec_multi_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ V, int K, int KP, int q0, const int32_t *__restrict__ rows, int64_t n_rows,
    const int32_t *__restrict__ colmap, F *__restrict__ out) {
    const int q = q0 + (threadIdx.x % KP);
    const int64_t ci = (int64_t)blockIdx.x * (256 / KP) + threadIdx.x / KP;
    if (ci >= n_rows || q >= K) return;
    const int64_t r = rows ? (int64_t)rows[ci] : ci;
    F acc = F(0);
    for (int64_t e = ptr[r]; e < ptr[r + 1]; ++e) {
        const int c = ind[e];
        if (colmap && colmap[c] < 0) continue;
        acc = fma(data[e], V[(int64_t)c * K + q], acc);
    }
    out[ci * K + q] += acc;
}

template <typename F>
__global__ __launch_bounds__(256) void csr_rmatvec_multi_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ V, int K, int KP, int q0, const int32_t *__restrict__ rows, int64_t n_rows,
    const int32_t *__restrict__ colmap, F *__restrict__ out) {
    const int q = q0 + (threadIdx.x % KP);
    const int64_t ci = (int64_t)blockIdx.x * (256 / KP) + threadIdx.x / KP;
    if (ci >= n_rows || q >= K) return;
    const int64_t r = rows ? (int64_t)rows[ci] : ci;
    const F vr = V[r * K + q];
    for (int64_t e = ptr[r]; e < ptr[r + 1]; ++e) {
        const int c = ind[e];
        const int pos = colmap ? colmap[c] : c;
        if (pos >= 0) atomic_add(out + (int64_t)pos * K + q, data[e] * vr);
    }
}

// out[ci, q] += sum_cj X[r, c] V[c, q]
template <typename F>
__global__ __launch_bounds__(256) void dense_matvec_multi_kernel(
    const F *__restrict__ X, int64_t n, int64_t m, int order_f, const F *__restrict__ V, int K, int KP,
    int q0, const int32_t *__restrict__ rows, int64_t n_rows, const int32_t *__restrict__ cols,
    int64_t n_cols, F *__restrict__ out) {
    const int q = q0 + (threadIdx.x % KP);
    const int64_t ci = (int64_t)blockIdx.x * (256 / KP) + threadIdx.x / KP;
    if (ci >= n_rows || q >= K) return;
    const int64_t r = rows ? (int64_t)rows[ci] : ci;
    F acc = F(0);
    for (int64_t cj = 0; cj < n_cols; ++cj) {
        const int64_t c = cols ? (int64_t)cols[cj] : cj;
        const F x = order_f ? X[c * n + r] : X[r * m + c];
        acc = fma(x, V[c * K + q], acc);
    }
    out[ci * K + q] += acc;
}

// out[cj, q] += sum_ci X[r, c] V[r, q]; blockIdx.y = chunk of rows, one atomic per (cj, q, chunk)
template <typename F>
__global__ __launch_bounds__(256) void dense_rmatvec_multi_kernel(
    const F *__restrict__ X, int64_t n, int64_t m, int order_f, const F *__restrict__ V, int K, int KP,
    int q0, const int32_t *__restrict__ rows, int64_t n_rows, int64_t rows_per_chunk,
    const int32_t *__restrict__ cols, int64_t n_cols, F *__restrict__ out) {
    const int q = q0 + (threadIdx.x % KP);
    const int64_t cj = (int64_t)blockIdx.x * (256 / KP) + threadIdx.x / KP;
    if (cj >= n_cols || q >= K) return;
    const int64_t c = cols ? (int64_t)cols[cj] : cj;
    const int64_t i0 = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t i1 = min(i0 + rows_per_chunk, n_rows);
    F acc = F(0);
    for (int64_t ci = i0; ci < i1; ++ci) {
        const int64_t r = rows ? (int64_t)rows[ci] : ci;
        const F x = order_f ? X[c * n + r] : X[r * m + c];
        acc = fma(x, V[r * K + q], acc);
    }
    atomic_add(out + cj * K + q, acc);
}

static inline int pow2_ge(int k) {
    int p = 1;
    while (p < k && p < 64) p <<= 1;
    return p;
}

template <typename F>
static int run_csr_multi(bool transpose, const F *data, const int32_t *ind, const int64_t *ptr,
                         int64_t n, int64_t m, const F *V, int64_t K, const int32_t *rows,
                         int64_t n_rows, const int32_t *cols, int64_t n_cols, F *out, hipStream_t st) {
    if (rows == nullptr) n_rows = n;
    if (K == 0 || n_rows == 0 || (cols != nullptr && n_cols == 0)) return TM_OK;
    int32_t *colmap = nullptr;
    if (cols != nullptr) {
        void *wsv = nullptr;
        int rc = get_workspace(sizeof(int32_t) * (size_t)m + 256, &wsv, st);
        if (rc) return rc;
        colmap = reinterpret_cast<int32_t *>(wsv);
        rc = build_col_map(colmap, m, cols, n_cols, st);
        if (rc) return rc;
    }
    const int KP = pow2_ge((int)std::min<int64_t>(K, 64));
    const unsigned grid = (unsigned)ceil_div(n_rows, 256 / KP);
    for (int q0 = 0; q0 < K; q0 += 64) {
        // the matvec form uses the map only as a mask (v / V keep their full length m)
        if (transpose)
            hipLaunchKernelGGL((csr_rmatvec_multi_kernel<F>), dim3(grid), dim3(256), 0, st, data, ind,
                               ptr, V, (int)K, KP, q0, rows, n_rows, colmap, out);
        else
            hipLaunchKernelGGL((csr_matvec_multi_kernel<F>), dim3(grid), dim3(256), 0, st, data, ind,
                               ptr, V, (int)K, KP, q0, rows, n_rows, colmap, out);
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

template <typename F>
static int run_dense_multi(bool transpose, const F *X, int64_t n, int64_t m, int order_f, const F *V,
                           int64_t K, const int32_t *rows, int64_t n_rows, const int32_t *cols,
                           int64_t n_cols, F *out, hipStream_t st) {
    if (rows == nullptr) n_rows = n;
    if (cols == nullptr) n_cols = m;
    if (K == 0 || n_rows == 0 || n_cols == 0) return TM_OK;
    const int KP = pow2_ge((int)std::min<int64_t>(K, 64));
    for (int q0 = 0; q0 < K; q0 += 64) {
        if (transpose) {
            const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(1024, ceil_div(n_rows, 4096)));
            const int64_t rpc = ceil_div(n_rows, chunks);
            hipLaunchKernelGGL((dense_rmatvec_multi_kernel<F>),
                               dim3((unsigned)ceil_div(n_cols, 256 / KP), (unsigned)ceil_div(n_rows, rpc)),
                               dim3(256), 0, st, X, n, m, order_f, V, (int)K, KP, q0, rows, n_rows, rpc,
                               cols, n_cols, out);
        } else {
            hipLaunchKernelGGL((dense_matvec_multi_kernel<F>), dim3((unsigned)ceil_div(n_rows, 256 / KP)),
                               dim3(256), 0, st, X, n, m, order_f, V, (int)K, KP, q0, rows, n_rows, cols,
                               n_cols, out);
        }
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

}  // namespace tmh

extern "C" {

#define TM_MULTI(SUF, F)                                                                              \
    int tm_csr_matvec_multi_##SUF(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n,    \
                                  int64_t m, const F *V, int64_t K, const int32_t *rows,              \
                                  int64_t n_rows, const int32_t *cols, int64_t n_cols, F *out,        \
                                  void *stream) {                                                     \
        return tmh::run_csr_multi<F>(false, data, ind, ptr, n, m, V, K, rows, n_rows, cols, n_cols,   \
                                     out, tmh::as_stream(stream));                                    \
    }                                                                                                 \
    int tm_csr_rmatvec_multi_##SUF(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n,   \
                                   int64_t m, const F *V, int64_t K, const int32_t *rows,             \
                                   int64_t n_rows, const int32_t *cols, int64_t n_cols, F *out,       \
                                   void *stream) {                                                    \
        return tmh::run_csr_multi<F>(true, data, ind, ptr, n, m, V, K, rows, n_rows, cols, n_cols,    \
                                     out, tmh::as_stream(stream));                                    \
    }                                                                                                 \
    int tm_dense_matvec_multi_##SUF(const F *X, int64_t n, int64_t m, int order_f, const F *V,        \
                                    int64_t K, const int32_t *rows, int64_t n_rows,                   \
                                    const int32_t *cols, int64_t n_cols, F *out, void *stream) {      \
        return tmh::run_dense_multi<F>(false, X, n, m, order_f, V, K, rows, n_rows, cols, n_cols,     \
                                       out, tmh::as_stream(stream));                                  \
    }                                                                                                 \
    int tm_dense_rmatvec_multi_##SUF(const F *X, int64_t n, int64_t m, int order_f, const F *V,       \
                                     int64_t K, const int32_t *rows, int64_t n_rows,                  \
                                     const int32_t *cols, int64_t n_cols, F *out, void *stream) {     \
        return tmh::run_dense_multi<F>(true, X, n, m, order_f, V, K, rows, n_rows, cols, n_cols, out, \
                                       tmh::as_stream(stream));                                       \
    }
TM_MULTI(f32, float)
TM_MULTI(f64, double)

}  // extern "C"
