// int64-index forms of the sparse block's entry points (reference: ext/sparse.pyx:13-15 `win_integral` -- the Cython
// functions take int32 OR int64 CSC / CSR index arrays; SURVEY.md 8b "Int in {i32, i64}").
//
// Every kernel of this library reads int32 column indices (they are below the column count, which fits int32 for any
// block the path can hold) and int64 row pointers.  A binding that holds int64 column indices should narrow them ONCE
// per block (tm_index_narrow_i64) and call the int32 entry points, as tabmat_amd/ext/_types.py does.  These `_i64`
// symbols are the drop-in for a caller that cannot keep a converted copy: they narrow the indices on the device into
// a scratch buffer of their own (per device and stream, grown on demand, separate from the kernels' workspace) and run
// the int32 kernel -- at the price of one extra pass over the index array (12 bytes per nonzero) per call.  An index
// outside [0, m) is CLAMPED into the range (the kernels index LDS tiles with it) and remembered in a device flag that
// tm_index_check_i64 reads back: the calls themselves stay asynchronous.
#include <map>
#include <mutex>

#include "common.hpp"

namespace tmh {

struct IdxScratch {
    int32_t *ptr = nullptr;
    size_t bytes = 0;
    int32_t *bad = nullptr;        // 4 bytes of device memory: != 0 once an index was out of range
};

__global__ void index_narrow_clamp_kernel(const int64_t *__restrict__ src, int64_t count, int64_t limit,
                                          int32_t *__restrict__ dst, int32_t *__restrict__ bad) {
    bool off = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t v = src[i];
        const bool o = v < 0 || v >= limit;
        off |= o;
        dst[i] = o ? 0 : (int32_t)v;
    }
    if (__any(off) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}
static std::map<std::pair<int, hipStream_t>, IdxScratch> g_idx;
static std::mutex g_idx_mu;

// int32 copy of `count` int64 indices (all < limit) in library-owned device memory, ordered on `st`
static int narrow_indices(const int64_t *src, int64_t count, int64_t limit, const int32_t **out, hipStream_t st) {
    *out = nullptr;
    if (count <= 0) return TM_OK;
    TM_REQUIRE(limit >= 0 && limit <= (1ll << 31), "more than 2^31 columns");
    int dev = 0;
    TM_HIP(hipGetDevice(&dev));
    IdxScratch *sc;
    {
        std::lock_guard<std::mutex> lk(g_idx_mu);
        sc = &g_idx[std::make_pair(dev, st)];
    }
    const size_t need = sizeof(int32_t) * (size_t)count + 256;
    if (need > sc->bytes) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            set_error("index scratch of %zu bytes needed while the stream is being captured: run the op once before", need);
            return TM_ENOMEM;
        }
        if (sc->ptr) {
            TM_HIP(hipStreamSynchronize(st));
            TM_HIP(hipFree(sc->ptr));
            sc->ptr = nullptr;
            sc->bytes = 0;
        }
        TM_HIP(hipMalloc(reinterpret_cast<void **>(&sc->ptr), need + need / 4));
        sc->bytes = need + need / 4;
    }
    if (sc->bad == nullptr) {
        TM_HIP(hipMalloc(reinterpret_cast<void **>(&sc->bad), 256));
        TM_HIP(hipMemsetAsync(sc->bad, 0, 256, st));
    }
    hipLaunchKernelGGL(index_narrow_clamp_kernel, dim3((unsigned)std::min<int64_t>(ceil_div(count, 256), 4096)),
                       dim3(256), 0, st, src, count, limit, sc->ptr, sc->bad);
    TM_LAUNCH_CHECK();
    *out = sc->ptr;
    return TM_OK;
}

}  // namespace tmh

using namespace tmh;

extern "C" {

/* bad[0] = 1 when any `_i64` call issued on `stream` since the last check met a column index outside [0, m) (such
 * indices were clamped to 0: the results of those calls are wrong); synchronises the stream, clears the flag. */
int tm_index_check_i64(void *stream, int32_t *h_bad) {
    TM_REQUIRE(h_bad != nullptr, "h_bad is NULL");
    *h_bad = 0;
    int dev = 0;
    TM_HIP(hipGetDevice(&dev));
    hipStream_t st = as_stream(stream);
    IdxScratch *sc = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_idx_mu);
        auto it = g_idx.find(std::make_pair(dev, st));
        if (it != g_idx.end()) sc = &it->second;
    }
    if (sc == nullptr || sc->bad == nullptr) return TM_OK;
    TM_HIP(hipMemcpyAsync(h_bad, sc->bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    TM_HIP(hipMemsetAsync(sc->bad, 0, sizeof(int32_t), st));
    TM_HIP(hipStreamSynchronize(st));
    return TM_OK;
}

#define TM_I64_FORMS(SUF, F)                                                                                          \
    int tm_sparse_sandwich_i64_##SUF(const F *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr,        \
                                     int64_t n, int64_t m, int64_t nnz, const F *d, const int32_t *rows,              \
                                     int64_t n_rows, const int32_t *cols, int64_t n_cols, F *out, void *stream) {     \
        const int32_t *ind = nullptr;                                                                                 \
        int rc = narrow_indices(csr_indices, nnz, m, &ind, as_stream(stream));                                        \
        if (rc) return rc;                                                                                            \
        return tm_sparse_sandwich_##SUF(csr_data, ind, csr_indptr, n, m, d, rows, n_rows, cols, n_cols, out, stream); \
    }                                                                                                                 \
    int tm_csr_dense_sandwich_i64_##SUF(const F *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr,     \
                                        int64_t n, int64_t m, int64_t nnz, const F *B, int64_t r, int order_f,        \
                                        const F *d, const int32_t *rows, int64_t n_rows, const int32_t *A_cols,       \
                                        int64_t nA, const int32_t *B_cols, int64_t nB, F *out, void *stream) {        \
        const int32_t *ind = nullptr;                                                                                 \
        int rc = narrow_indices(csr_indices, nnz, m, &ind, as_stream(stream));                                        \
        if (rc) return rc;                                                                                            \
        return tm_csr_dense_sandwich_##SUF(csr_data, ind, csr_indptr, n, m, B, r, order_f, d, rows, n_rows, A_cols,   \
                                           nA, B_cols, nB, out, stream);                                              \
    }                                                                                                                 \
    int tm_csr_matvec_i64_##SUF(const F *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,  \
                                int64_t m, int64_t nnz, const F *v, const int32_t *rows, int64_t n_rows,              \
                                const int32_t *cols, int64_t n_cols, F *out, void *stream) {                          \
        const int32_t *ind = nullptr;                                                                                 \
        int rc = narrow_indices(csr_indices, nnz, m, &ind, as_stream(stream));                                        \
        if (rc) return rc;                                                                                            \
        return tm_csr_matvec_##SUF(csr_data, ind, csr_indptr, n, m, v, rows, n_rows, cols, n_cols, out, stream);      \
    }                                                                                                                 \
    int tm_csr_rmatvec_i64_##SUF(const F *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n, \
                                 int64_t m, int64_t nnz, const F *v, const int32_t *rows, int64_t n_rows,             \
                                 const int32_t *cols, int64_t n_cols, F *out, void *stream) {                         \
        const int32_t *ind = nullptr;                                                                                 \
        int rc = narrow_indices(csr_indices, nnz, m, &ind, as_stream(stream));                                        \
        if (rc) return rc;                                                                                            \
        return tm_csr_rmatvec_##SUF(csr_data, ind, csr_indptr, n, m, v, rows, n_rows, cols, n_cols, out, stream);     \
    }                                                                                                                 \
    int tm_csr_col_sq_i64_##SUF(const F *csr_data, const int64_t *csr_indices, const int64_t *csr_indptr, int64_t n,  \
                                int64_t m, int64_t nnz, const F *w, F *out, void *stream) {                           \
        const int32_t *ind = nullptr;                                                                                 \
        int rc = narrow_indices(csr_indices, nnz, m, &ind, as_stream(stream));                                        \
        if (rc) return rc;                                                                                            \
        return tm_csr_col_sq_##SUF(csr_data, ind, csr_indptr, n, m, w, out, stream);                                  \
    }

TM_I64_FORMS(f32, float)
TM_I64_FORMS(f64, double)

}  // extern "C"
