// Deterministic weighted histogram for categorical blocks (K4a: X' v = out[c] = sum of v over the
// rows of category c).  The reference made this product deterministic on purpose
// (ext/cat_split_helpers-tmpl.cpp:33-38, CHANGELOG.rst: thread-owned partial sums combined in a
// fixed order); the fast path here (hist_lds_kernel, cat.hip) adds with LDS float atomics, whose
// order -- and therefore the last bits of a weighted sum -- changes from run to run.
// This path trades ~0.1 ms at 10M rows for bitwise reproducibility: the rows of a block are
// grouped by category ONCE (a stable device sort at ingest: `perm`), every category's run is cut
// into fixed blocks of DET_BLOCK rows (`bstart`), a workgroup sums one block in a fixed order
// (thread-strided, then a fixed tree) into a double, and one thread per category adds the block
// sums in order.
#include "common.hpp"

namespace tmh {

constexpr int DET_BLOCK = 4096;

template <typename F>
__global__ __launch_bounds__(256) void det_block_sum_kernel(const int32_t *__restrict__ perm,
                                                            const int64_t *__restrict__ bstart,
                                                            const F *__restrict__ v,
                                                            double *__restrict__ partial) {
    __shared__ double red[256];
    const int64_t b0 = bstart[blockIdx.x], b1 = bstart[blockIdx.x + 1];
    double acc = 0.0;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += 256) acc += (double)v[perm[i]];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

template <typename F>
__global__ __launch_bounds__(256) void det_final_kernel(const double *__restrict__ partial,
                                                        const int64_t *__restrict__ cat_bptr,
                                                        int64_t n_cols, F *__restrict__ out,
                                                        int accumulate) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    double s = 0.0;
    for (int64_t b = cat_bptr[c]; b < cat_bptr[c + 1]; ++b) s += partial[b];
    out[c] = accumulate ? (F)((double)out[c] + s) : (F)s;
}

template <typename F>
static int run_det(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                   const int64_t *cat_bptr, int64_t n_cols, const F *v, F *out, int accumulate,
                   hipStream_t st) {
    if (n_cols == 0) return TM_OK;
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(double) * (size_t)std::max<int64_t>(n_blocks, 1) + 256, &wsv, st);
    if (rc) return rc;
    double *partial = reinterpret_cast<double *>(wsv);
    if (n_blocks > 0) {
        hipLaunchKernelGGL((det_block_sum_kernel<F>), dim3((unsigned)n_blocks), dim3(256), 0, st, perm,
                           bstart, v, partial);
        TM_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL((det_final_kernel<F>), dim3((unsigned)ceil_div(n_cols, 256)), dim3(256), 0, st,
                       partial, cat_bptr, n_cols, out, accumulate);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_cat_det_block_rows(void) { return tmh::DET_BLOCK; }

int tm_cat_transpose_matvec_det_f32(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                    const int64_t *cat_bptr, int64_t n_cols, const float *v,
                                    float *out, int accumulate, void *stream) {
    return tmh::run_det<float>(perm, bstart, n_blocks, cat_bptr, n_cols, v, out, accumulate,
                               tmh::as_stream(stream));
}
int tm_cat_transpose_matvec_det_f64(const int32_t *perm, const int64_t *bstart, int64_t n_blocks,
                                    const int64_t *cat_bptr, int64_t n_cols, const double *v,
                                    double *out, int accumulate, void *stream) {
    return tmh::run_det<double>(perm, bstart, n_blocks, cat_bptr, n_cols, v, out, accumulate,
                                tmh::as_stream(stream));
}

}  // extern "C"
