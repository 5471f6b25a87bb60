// StandardizedMatrix on device blocks (reference: src/tabmat/standardized_mat.py:123-230):
// the O(p^2) rank-one corrections of the sandwich and the sums over the per-call vector, so that a
// standardized sandwich / transpose_matvec never leaves the device.
//   self[i, j] = mult[j] * mat[i, j] + shift[j]
//   sandwich   = outer(mult, mult) * inner + outer(m, shift) + outer(shift, m) + outer(shift, shift) * S
//   with inner = mat' diag(d) mat,  m = mult * (mat' d),  S = sum(d[rows])
#include "common.hpp"

namespace tmh {

constexpr int SUM_BLOCKS = 256;

// partial[b] = sum of v over the rows of block b (grid-stride, fixed order inside a block)
template <typename F>
__global__ __launch_bounds__(256) void vec_sum_partial_kernel(const F *__restrict__ v,
                                                              const int32_t *__restrict__ rows,
                                                              int64_t n, double *__restrict__ partial) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        acc += (double)v[rows ? (int64_t)rows[i] : i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(64) void vec_sum_final_kernel(const double *__restrict__ partial, int nb,
                                                           double *__restrict__ out) {
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < nb; ++b) s += partial[b];   // fixed order: run-to-run reproducible
        out[0] = s;
    }
}

// out[i, j] = inner[i, j] * mi * mj + m[i] * shift[j] + shift[i] * m[j] + shift[i] * shift[j] * S
// (in place on `inner`); m[i] = xtd[i] * mult[i]; mult == NULL: all ones; diag_only: `inner` holds
// only the diagonal (a categorical block's sandwich) as a length-k vector in inner_diag.
__global__ void standardize_sandwich_kernel(double *__restrict__ out, const double *__restrict__ inner_diag,
                                            const double *__restrict__ xtd,
                                            const double *__restrict__ shift,
                                            const double *__restrict__ mult,
                                            const double *__restrict__ sum_d, int64_t k) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * k) return;
    const int64_t i = e / k, j = e % k;
    const double mi = mult ? mult[i] : 1.0, mj = mult ? mult[j] : 1.0;
    double in = inner_diag ? (i == j ? inner_diag[i] : 0.0) : out[e];
    out[e] = in * mi * mj + xtd[i] * mi * shift[j] + shift[i] * xtd[j] * mj +
             shift[i] * shift[j] * sum_d[0];
}

// The same result from a PARTLY CENTRED inner product (tm_dense_sandwich*_centered_*): column i of the inner
// matrix was taken as x_i - c_i (c_i = 0: as it is), so that self[:, i] = mult_i (x_i - c_i) + delta_i with
// delta_i = shift_i + c_i mult_i -- for the centre a standardized column has, c_i = -shift_i / mult_i, delta is
// rounding-sized and the mean-sized rank-one terms of the reference's formula never form.
//   xtd[i]   = sum_r d_r (x_ri - c_i)                      (centred column sums)
//   group[i] = id of the block whose self term was computed centred (-1: none).  inner[i][j] is the centred
//              product where group[i] == group[j] >= 0 and the RAW product sum_r d_r x_ri x_rj elsewhere (cross
//              terms with sparse / categorical blocks, blocks without a centred kernel); those entries are
//              centred here: S'_ij = S_ij - c_i t_j - c_j t_i + c_i c_j S with the raw t = xtd + c S.
//   out[i][j] = S'_ij mi mj + mi xtd_i delta_j + delta_i mj xtd_j + delta_i delta_j S
__global__ void standardize_sandwich_centered_kernel(double *__restrict__ out, const double *__restrict__ xtd,
                                                     const double *__restrict__ center,
                                                     const int32_t *__restrict__ group,
                                                     const double *__restrict__ shift,
                                                     const double *__restrict__ mult,
                                                     const double *__restrict__ sum_d, int64_t k) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= k * k) return;
    const int64_t i = e / k, j = e % k;
    const double mi = mult ? mult[i] : 1.0, mj = mult ? mult[j] : 1.0;
    const double ci = center[i], cj = center[j], S = sum_d[0];
    const double di = __builtin_fma(ci, mi, shift[i]), dj = __builtin_fma(cj, mj, shift[j]);
    double in = out[e];
    if (!(group[i] >= 0 && group[i] == group[j])) {
        const double ti = __builtin_fma(ci, S, xtd[i]), tj = __builtin_fma(cj, S, xtd[j]);
        in = in - ci * tj - cj * ti + ci * cj * S;
    }
    out[e] = in * mi * mj + xtd[i] * mi * dj + di * xtd[j] * mj + di * dj * S;
}

template <typename F>
static int run_vec_sum(const F *v, const int32_t *rows, int64_t n, double *out, hipStream_t st) {
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(double) * SUM_BLOCKS + 256, &wsv, st);
    if (rc) return rc;
    double *partial = reinterpret_cast<double *>(wsv);
    const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(SUM_BLOCKS, ceil_div(n, 256)));
    hipLaunchKernelGGL((vec_sum_partial_kernel<F>), dim3(nb), dim3(256), 0, st, v, rows, n, partial);
    TM_LAUNCH_CHECK();
    hipLaunchKernelGGL(vec_sum_final_kernel, dim3(1), dim3(64), 0, st, partial, nb, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_vec_sum_f32(const float *v, const int32_t *rows, int64_t n, double *out, void *stream) {
    return tmh::run_vec_sum<float>(v, rows, n, out, tmh::as_stream(stream));
}
int tm_vec_sum_f64(const double *v, const int32_t *rows, int64_t n, double *out, void *stream) {
    return tmh::run_vec_sum<double>(v, rows, n, out, tmh::as_stream(stream));
}

int tm_standardize_sandwich_f64(double *inout, const double *inner_diag, const double *xtd,
                                const double *shift, const double *mult, const double *sum_d,
                                int64_t k, void *stream) {
    if (k == 0) return TM_OK;
    hipLaunchKernelGGL(tmh::standardize_sandwich_kernel, dim3((unsigned)tmh::ceil_div(k * k, 256)),
                       dim3(256), 0, tmh::as_stream(stream), inout, inner_diag, xtd, shift, mult,
                       sum_d, k);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

int tm_standardize_sandwich_centered_f64(double *inout, const double *xtd, const double *center,
                                         const int32_t *group, const double *shift, const double *mult,
                                         const double *sum_d, int64_t k, void *stream) {
    if (k == 0) return TM_OK;
    hipLaunchKernelGGL(tmh::standardize_sandwich_centered_kernel, dim3((unsigned)tmh::ceil_div(k * k, 256)),
                       dim3(256), 0, tmh::as_stream(stream), inout, xtd, center, group, shift, mult, sum_d, k);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // extern "C"
