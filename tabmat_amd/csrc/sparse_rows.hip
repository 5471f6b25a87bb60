// K3 with a ROW LIST: out = A[rows, :]' diag(d[rows]) B[rows, :] at a cost proportional to the
// number of selected rows (reference: the `for Ci in rows` loops of _csr_dense{C,F}_sandwich,
// ext/sparse_helpers-tmpl.cpp:67-131).  The slab kernels walk every row of the block (a row
// restriction there is d = 0 on the other rows: the stream and the dense block are still read in
// full); for a short list -- glum's hessian updates pass the rows whose weights changed -- this
// kernel is the cheaper one: a wave takes a selected row, keeps its 128 dense values in registers
// (two coalesced 512-byte loads) and adds value x d x row to the LDS tile row of every nonzero of
// that row in the workgroup's 128-column chunk (chunk-major twin + {start, end} table of the
// selected rows, the same table the row-restricted K2 uses).  Two rows per wave are in flight.
#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr int RK_TS = 128;          // sparse columns per chunk = LDS tile rows
constexpr int RK_W = 128;           // dense columns per part
constexpr int RK_WAVES = 16;

// IX = int32_t: column indices of the whole block; IX = uint8_t: the column inside the 128-column chunk (the byte
// twin K2b streams -- round 6: the int32 chunk-major columns are no longer kept in HBM)
template <typename F, typename IX>
__global__ __launch_bounds__(RK_WAVES * 64) void csr_dense_rows_kernel(
    const F *__restrict__ data, const IX *__restrict__ ind, const int32_t *__restrict__ ranges,
    int64_t n_sel, const int32_t *__restrict__ rows, const F *__restrict__ d_sel,
    const F *__restrict__ B, int64_t ldb, int order_f, int64_t n, int nB, int64_t rows_per_block,
    F *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);   // [RK_TS][RK_W] doubles
    const int chunk = blockIdx.y;
    const int j0 = blockIdx.z * RK_W;
    for (int b = threadIdx.x; b < RK_TS * RK_W; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_sel);
    const int32_t *rg = ranges + (int64_t)chunk * 2 * n_sel;
    const int c0 = j0 + lane, c1 = j0 + 64 + lane;        // this lane's two dense columns
    auto bval = [&](int64_t row, int c) -> F {
        if (c >= nB) return F(0);
        return order_f ? B[(int64_t)c * ldb + row] : B[row * ldb + c];
    };
    // two selected rows per wave and step: their loads are issued together.  The entries of a row
    // are loaded one per lane (coalesced) and broadcast with v_readlane; everything about an
    // entry is then wave-uniform and the loop over the entries has a scalar trip count.
    auto rl = [](F v, int l) -> F {
        if constexpr (sizeof(F) == 8) {
            const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
            return __builtin_bit_cast(F, ((unsigned long long)hi << 32) | lo);
        } else {
            return __builtin_bit_cast(F, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
        }
    };
    auto scatter_row = [&](int e0, int ne, F dk, F x0, F x1) {
        for (int base = 0; base < ne; base += 64) {          // > 64 entries of a row in one chunk: rare
            const int cnt = min(ne - base, 64);
            const int ci = lane < cnt ? (int)ind[e0 + base + lane] - (sizeof(IX) == 1 ? 0 : chunk * RK_TS) : 0;
            const F cv = lane < cnt ? data[e0 + base + lane] * dk : F(0);
            for (int e = 0; e < cnt; ++e) {
                const int col = __builtin_amdgcn_readlane(ci, e);
                const F v = rl(cv, e);
                atomic_add(tile + col * RK_W + lane, (lds_acc_t)(v * x0));
                atomic_add(tile + col * RK_W + 64 + lane, (lds_acc_t)(v * x1));
            }
        }
    };
    for (int64_t k = t0 + wave * 2; k < t1; k += RK_WAVES * 2) {
        const int64_t k1 = min(k + 1, t1 - 1);
        const bool two = k + 1 < t1;
        const int64_t ra = rows[k], rb = rows[k1];
        const F da = d_sel[k], db = two ? d_sel[k1] : F(0);
        const int a0 = __builtin_amdgcn_readfirstlane(rg[2 * k]);
        const int na = __builtin_amdgcn_readfirstlane(rg[2 * k + 1]) - a0;
        const int b0 = __builtin_amdgcn_readfirstlane(rg[2 * k1]);
        const int nb = two ? __builtin_amdgcn_readfirstlane(rg[2 * k1 + 1]) - b0 : 0;
        const F xa0 = bval(ra, c0), xa1 = bval(ra, c1), xb0 = bval(rb, c0), xb1 = bval(rb, c1);
        if (da != F(0)) scatter_row(a0, na, da, xa0, xa1);
        if (db != F(0)) scatter_row(b0, nb, db, xb0, xb1);
    }
    __syncthreads();
    // ws: [part = chunk * n_dense_parts + z][block][RK_TS * RK_W]
    F *dst = ws + (((int64_t)chunk * gridDim.z + blockIdx.z) * gridDim.x + blockIdx.x) * (RK_TS * RK_W);
    for (int b = threadIdx.x; b < RK_TS * RK_W; b += blockDim.x) dst[b] = (F)tile[b];
}

// tmp [chunk][zpart][RK_TS][RK_W] -> out[m][nB]
template <typename F>
__global__ void rows_untile_kernel(const F *__restrict__ tmp, int64_t m, int64_t nB, int nz,
                                   F *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * nB) return;
    const int64_t i = e / nB, j = e % nB;
    out[e] = tmp[(((i / RK_TS) * nz + j / RK_W) * RK_TS + i % RK_TS) * RK_W + j % RK_W];
}

template <typename F, typename IX>
static int run_csr_dense_rows(const F *data, const IX *ind, const int32_t *ranges, int64_t n_sel,
                              const int32_t *rows, const F *d_sel, int64_t n, int64_t m, const F *B,
                              int64_t nB, int order_f, F *out, hipStream_t st) {
    const int64_t total = m * nB;
    if (total == 0) return TM_OK;
    if (n_sel == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    const int nch = (int)ceil_div(m, RK_TS);
    const int nz = (int)ceil_div(nB, RK_W);
    int64_t nblk = std::max<int64_t>(1, NUM_CU / ((int64_t)nch * nz));
    nblk = std::min<int64_t>(nblk, ceil_div(n_sel, RK_WAVES * 2 * 4));
    const int64_t rpb = ceil_div(n_sel, nblk);
    nblk = ceil_div(n_sel, rpb);
    const int64_t stride = (int64_t)RK_TS * RK_W;
    const int n_parts = nch * nz;
    const size_t tmp_bytes = (sizeof(F) * (size_t)(n_parts * stride) + 255) / 256 * 256;
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_parts * nblk * stride) + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    const size_t lds = sizeof(lds_acc_t) * (size_t)stride;
    auto kern = &csr_dense_rows_kernel<F, IX>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)nch, (unsigned)nz), dim3(RK_WAVES * 64), lds,
                       st, data, ind, ranges, n_sel, rows, d_sel, B, order_f ? n : nB, order_f, n,
                       (int)nB, rpb, ws);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_parts, tmp, n_parts * stride, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((rows_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                       tmp, m, nB, nz, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_csr_dense_sandwich_rows_f32(const float *cm_data, const int32_t *cm_indices,
                                   const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                   const float *d_sel, int64_t n, int64_t m, const float *B, int64_t r,
                                   int order_f, float *out, void *stream) {
    return tmh::run_csr_dense_rows<float, int32_t>(cm_data, cm_indices, row_ranges, n_sel, rows, d_sel, n, m, B, r,
                                          order_f, out, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_rows_f64(const double *cm_data, const int32_t *cm_indices,
                                   const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                   const double *d_sel, int64_t n, int64_t m, const double *B, int64_t r,
                                   int order_f, double *out, void *stream) {
    return tmh::run_csr_dense_rows<double, int32_t>(cm_data, cm_indices, row_ranges, n_sel, rows, d_sel, n, m, B, r,
                                           order_f, out, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_rows_u8_f32(const float *cm_data, const uint8_t *cm_col8,
                                      const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                      const float *d_sel, int64_t n, int64_t m, const float *B, int64_t r,
                                      int order_f, float *out, void *stream) {
    return tmh::run_csr_dense_rows<float, uint8_t>(cm_data, cm_col8, row_ranges, n_sel, rows, d_sel, n, m, B, r,
                                                   order_f, out, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_rows_u8_f64(const double *cm_data, const uint8_t *cm_col8,
                                      const int32_t *row_ranges, int64_t n_sel, const int32_t *rows,
                                      const double *d_sel, int64_t n, int64_t m, const double *B, int64_t r,
                                      int order_f, double *out, void *stream) {
    return tmh::run_csr_dense_rows<double, uint8_t>(cm_data, cm_col8, row_ranges, n_sel, rows, d_sel, n, m, B, r,
                                                    order_f, out, tmh::as_stream(stream));
}

}  // extern "C"
