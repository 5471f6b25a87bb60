// All categorical x categorical blocks of a SplitMatrix sandwich (and the categorical diagonals)
// in ONE pass over the codes (reference: ext/split.pyx:83-111 sandwich_cat_cat ->
// ext/cat_split_helpers-tmpl.cpp:44-94, and ext/categorical.pyx:183-218 for the diagonals).
//
// A design with k categoricals has k (k - 1) / 2 pair tables.  One launch per table (hist_lds_kernel)
// is ~30 us of launches for a few KB of work: 30 categoricals of 12 levels = 435 tables = 11.7 ms for
// 2M rows whose codes are 0.26 GB.  Here the tables are packed into BUNDLES that fit one LDS tile
// of doubles (16 384 bins); a workgroup streams rows (lane <-> row: d and all k codes in registers),
// and for every pair of its bundle adds d into  tile[offset + c_i * L_j + c_j]  (diagonal: offset +
// c_i).  The pair loop is unrolled over the KT (KT + 1) / 2 slots of a KT-categorical call; a slot's
// tile offset (or -1: not in this bundle) is a scalar load, so absent pairs cost a scalar branch.
// A second kernel scatters every table (and its mirror) into the p x p result.
#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr int CP_MAX = 32;
constexpr int CP_BINS = 16384;          // doubles per bundle tile (128 KB)
constexpr int CP_MAX_TABLES = 128;      // tables per bundle (two register pages of 64)

struct CatSetK {
    const int32_t *codes[CP_MAX];
    int ncol[CP_MAX];
    int drop[CP_MAX];
    int n_cats;
};

template <typename F, int KT>
__global__ __launch_bounds__(1024) void multi_cat_pairs_kernel(
    CatSetK cs, const F *__restrict__ d, const int32_t *__restrict__ rows, int64_t n_iter,
    int64_t rows_per_block, const int32_t *__restrict__ pair_list, int bins,
    double *__restrict__ ws) {
    // pair_list: word 0 = words per bundle row (a multiple of 4), words 1..3 unused; then per bundle
    // a row [used-categorical mask, number of tables, 0, 0, then 4 words per table {i, j, tile
    // offset, L_j}].  The table loop is a scalar loop; the codes sit in a register VECTOR indexed
    // with the (uniform) i / j of the table.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *tile = reinterpret_cast<double *>(smem_raw);
    for (int b = threadIdx.x; b < bins; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    constexpr int UNR = KT <= 8 ? 4 : 2;           // rows per thread and step (loads in flight)
    const int stride = __builtin_amdgcn_readfirstlane(pair_list[0]);      // words per bundle row
    const int32_t *pl = pair_list + 4 + (int64_t)blockIdx.y * stride;
    const unsigned used = (unsigned)__builtin_amdgcn_readfirstlane(pl[0]);
    const int np = __builtin_amdgcn_readfirstlane(pl[1]);
    const int4 *tb4 = reinterpret_cast<const int4 *>(pl + 4);
    // the bundle's table list lives in REGISTERS (lane p of page g = table 64 g + p), read back with
    // v_readlane in the table loop: a scalar load per table there made every iteration wait for
    // memory AND for the LDS atomics in flight (SMEM and LDS share lgkmcnt)
    constexpr int NPAGE = CP_MAX_TABLES / 64;
    const int lane = threadIdx.x & 63;
    int4 page[NPAGE];
#pragma unroll
    for (int g = 0; g < NPAGE; ++g) page[g] = 64 * g + lane < np ? tb4[64 * g + lane] : int4{0, 0, 0, 0};
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);
    for (int64_t tb = t0; tb < t1; tb += (int64_t)UNR * blockDim.x) {
        int64_t k[UNR];
        double dk[UNR];
        // (a VECTOR per row: dynamic extraction with a uniform index is a v_movrels, an int array
        // of 16 or 32 would go to scratch)
        typedef int cvec_t __attribute__((ext_vector_type(KT)));
        cvec_t c[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t t = tb + (int64_t)u * blockDim.x + threadIdx.x;
            k[u] = t < t1 ? (rows ? (int64_t)rows[t] : t) : -1;
        }
        // ALL code loads of the step are issued before the first one is used (row index clamped
        // instead of a lane branch around the load: a guarded load is waited for inside its branch,
        // which made the ~10 categoricals of a bundle ten dependent memory round trips)
        int64_t kc[UNR];
        bool live[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            kc[u] = k[u] >= 0 ? k[u] : 0;
            dk[u] = (double)d[kc[u]];
        }
        static_for<KT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (i < cs.n_cats && (used >> i & 1u)) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) c[u][i] = __builtin_nontemporal_load(cs.codes[i] + kc[u]);
            }
        });
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (k[u] < 0) dk[u] = 0.0;
            live[u] = dk[u] != 0.0;
        }
        static_for<KT>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const bool on = i < cs.n_cats && (used >> i & 1u);
#pragma unroll
            for (int u = 0; u < UNR; ++u) c[u][i] = (on && live[u]) ? c[u][i] - cs.drop[i] : -1;
        });
        static_for<NPAGE>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            const int cnt = min(np - 64 * g, 64);
            for (int p = 0; p < cnt; ++p) {
                const int i = __builtin_amdgcn_readlane(page[g].x, p), j = __builtin_amdgcn_readlane(page[g].y, p);
                const int off = __builtin_amdgcn_readlane(page[g].z, p), lj = __builtin_amdgcn_readlane(page[g].w, p);
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int ci = c[u][i], cj = c[u][j];
                    if (ci >= 0 && cj >= 0) atomicAdd(tile + off + (i == j ? ci : ci * lj + cj), dk[u]);
                }
            }
        });
    }
    __syncthreads();
    double *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * bins;
    for (int b = threadIdx.x; b < bins; b += blockDim.x) dst[b] = tile[b];
}

// desc[pair] = {table offset, L_i, L_j, first position of block i, of block j, diagonal flag}
__global__ __launch_bounds__(256) void multi_cat_pairs_scatter_kernel(
    const double *__restrict__ tables, const int64_t *__restrict__ desc,
    const int64_t *__restrict__ pos, double *__restrict__ out, int64_t p) {
    const int64_t *q = desc + (int64_t)blockIdx.x * 6;
    const int64_t toff = q[0], li = q[1], lj = q[2], pi = q[3], pj = q[4];
    const bool diag = q[5] != 0;
    const int64_t total = diag ? li : li * lj;
    for (int64_t e = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.y * blockDim.x) {
        const double v = tables[toff + e];
        if (diag) {
            const int64_t r = pos[pi + e];
            out[r * p + r] = v;
        } else {
            const int64_t r = pos[pi + e / lj], cc = pos[pj + e % lj];
            out[r * p + cc] = v;
            out[cc * p + r] = v;
        }
    }
}

template <typename F>
static int run_multi_cat_pairs(const void *const *h_codes, const int64_t *h_ncols,
                               const int32_t *h_drop, int n_cats, int64_t n, const F *d,
                               const int32_t *rows, int64_t n_rows, const int32_t *pair_list,
                               int n_bundles, int64_t bins, const int64_t *desc, int64_t n_pairs,
                               const int64_t *pos, double *tables, double *out, int64_t p,
                               hipStream_t st) {
    if (n_cats < 1 || n_cats > CP_MAX || bins < 1 || bins > CP_BINS || n_bundles < 1) {
        set_error("tm_multi_cat_pairs: 1..%d categoricals, 1..%d bins per bundle", CP_MAX, CP_BINS);
        return TM_EINVAL;
    }
    CatSetK cs;
    for (int c = 0; c < CP_MAX; ++c) {
        cs.codes[c] = c < n_cats ? reinterpret_cast<const int32_t *>(h_codes[c]) : nullptr;
        cs.ncol[c] = c < n_cats ? (int)h_ncols[c] : 0;
        cs.drop[c] = c < n_cats ? h_drop[c] : 0;
    }
    cs.n_cats = n_cats;
    const int64_t n_iter = rows ? n_rows : n;
    if (n_iter == 0) {
        TM_HIP(hipMemsetAsync(tables, 0, sizeof(double) * (size_t)(n_bundles * bins), st));
    } else {
        int64_t nblk = std::max<int64_t>(1, (int64_t)NUM_CU / n_bundles);
        nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, ceil_div(n_iter, 8192)));
        const int64_t rpb = ceil_div(n_iter, nblk);
        nblk = ceil_div(n_iter, rpb);
        void *wsv = nullptr;
        int rc = get_workspace(sizeof(double) * (size_t)(n_bundles * nblk * bins) + 256, &wsv, st);
        if (rc) return rc;
        double *ws = reinterpret_cast<double *>(wsv);
        const size_t lds = sizeof(double) * (size_t)bins;
        auto go = [&](auto kern) -> int {
            if (lds > 48 * 1024)
                TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            prof_begin(st);
            hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_bundles), dim3(1024), lds, st, cs,
                               d, rows, n_iter, rpb, pair_list, (int)bins, ws);
            prof_end(st);
            TM_LAUNCH_CHECK();
            return TM_OK;
        };
        if (n_cats <= 8) rc = go(&multi_cat_pairs_kernel<F, 8>);
        else if (n_cats <= 16) rc = go(&multi_cat_pairs_kernel<F, 16>);
        else rc = go(&multi_cat_pairs_kernel<F, 32>);
        if (rc) return rc;
        rc = launch_reduce_partials<double>(ws, bins, (int)nblk, n_bundles, tables, n_bundles * bins,
                                            false, st);
        if (rc) return rc;
    }
    if (out != nullptr && n_pairs > 0) {
        hipLaunchKernelGGL(multi_cat_pairs_scatter_kernel, dim3((unsigned)n_pairs, 4), dim3(256), 0, st,
                           tables, desc, pos, out, p);
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

}  // namespace tmh

extern "C" {
int tm_multi_cat_pairs_max_bins(void) { return tmh::CP_BINS; }
int tm_multi_cat_pairs_max_tables(void) { return tmh::CP_MAX_TABLES; }
int tm_multi_cat_pairs_f32(const void *const *h_codes, const int64_t *h_ncols, const int32_t *h_drop,
                           int n_cats, int64_t n, const float *d, const int32_t *rows, int64_t n_rows,
                           const int32_t *pair_list, int n_bundles, int64_t bins, const int64_t *desc,
                           int64_t n_pairs, const int64_t *pos, double *tables, double *out, int64_t p,
                           void *stream) {
    return tmh::run_multi_cat_pairs<float>(h_codes, h_ncols, h_drop, n_cats, n, d, rows, n_rows, pair_list,
                                           n_bundles, bins, desc, n_pairs, pos, tables, out, p,
                                           tmh::as_stream(stream));
}
int tm_multi_cat_pairs_f64(const void *const *h_codes, const int64_t *h_ncols, const int32_t *h_drop,
                           int n_cats, int64_t n, const double *d, const int32_t *rows, int64_t n_rows,
                           const int32_t *pair_list, int n_bundles, int64_t bins, const int64_t *desc,
                           int64_t n_pairs, const int64_t *pos, double *tables, double *out, int64_t p,
                           void *stream) {
    return tmh::run_multi_cat_pairs<double>(h_codes, h_ncols, h_drop, n_cats, n, d, rows, n_rows, pair_list,
                                            n_bundles, bins, desc, n_pairs, pos, tables, out, p,
                                            tmh::as_stream(stream));
}
}  // extern "C"
