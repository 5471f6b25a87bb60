// K1c  Dense self sandwich  out = X' diag(d) X  for a C-ordered f64 block of <= 128 columns, built to
// run on the SAME compute units, at the same time, as an LDS- or HBM-bound partner kernel
// (reference: ext/dense_helpers-tmpl.cpp:266-311, the dense term of split_matrix.py:337-354).
//
// The step of a SplitMatrix sandwich is a chain of kernels that each saturate a different pipe of
// the CU: the MFMA syrk (matrix pipe), the sparse self sandwich (LDS atomics + integer VALU), the
// categorical cross terms (HBM + LDS atomics).  The plain syrk_kernel (dense.hip) is sized to own
// a CU: 2 workgroups x 37 KB of LDS, two 144-register waves per SIMD.  This variant is sized to be
// a GUEST:
//   * LDS: 2 x 12 rows x 1152 B + d = 27.8 KB -> fits beside a 128 KB tile (160 KB per CU,
//     allocated in 1280-byte granules on gfx950);
//   * registers: <= 168 per lane (amdgpu_waves_per_eu(3)): one wave per SIMD beside three
//     112-register waves of the partner;
//   * LDS instructions: the 8 column blocks are VIRTUAL blocks made of the even / odd columns of a
//     32-column group, so one ds_read_b128 fetches the MFMA fragments of two blocks (4 reads + d
//     per 9 MFMAs and wave, against 8 + d) -- the partner lives on the LDS pipe;
//   * with one wave per SIMD the fragments of the next row group are read before the MFMAs of
//     the current one are issued (software prefetch instead of a second wave);
//   * work is handed out in items of CO_CPI chunks through an atomic counter, because a guest does
//     not know how many of its workgroups share a CU with the partner and how many run alone.
// The column sums X' d fall out of the A-side fragments (2 v_add_f64 per row group and wave), so
// StandardizedMatrix.sandwich needs no second pass over the block (standardized_mat.py:149-150).
#include <algorithm>

#include "common.hpp"

namespace tmh {

constexpr int CO_RS = 12;                 // rows per chunk (3 MFMA row groups of 4)
constexpr int CO_W = 128;                 // padded columns
constexpr int CO_LDW = CO_W + 16;         // LDS row stride in doubles (1152 B)
constexpr int CO_CPI = 64;                // chunks per work item (768 rows)
constexpr int CO_ITEM_ROWS = CO_CPI * CO_RS;
constexpr int CO_T = 36;                  // lower-triangular 16 x 16 tiles of 8 blocks
constexpr int CO_NWAVES = 4;
constexpr int CO_THREADS = CO_NWAVES * 64;
constexpr int CO_CHUNK = CO_RS * CO_LDW;  // doubles per LDS buffer
constexpr size_t CO_LDS = sizeof(double) * (size_t)(2 * CO_CHUNK + 2 * CO_RS) + 32 + sizeof(double) * CO_W;   // + slot, centres

typedef double co_acc_t __attribute__((ext_vector_type(4)));
typedef double co_vec2 __attribute__((ext_vector_type(2)));

constexpr int co_tri_row(int t) {
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    return r;
}

// actual column of virtual index v (block b = v >> 4 holds the even (b even) / odd (b odd)
// columns of the 32-column group b >> 1)
__host__ __device__ constexpr int co_actual_col(int v) {
    return 32 * ((v >> 4) >> 1) + 2 * (v & 15) + ((v >> 4) & 1);
}

template <int WID, int S>
__device__ __forceinline__ void co_mfma_set(const double (&xa)[8], const double (&xb)[8],
                                            co_acc_t (&acc)[9]) {
    if constexpr (S < 9) {
        constexpr int t = WID + S * CO_NWAVES;
        constexpr int bi = co_tri_row(t);
        constexpr int bj = t - bi * (bi + 1) / 2;
        acc[S] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[bi], xb[bj], acc[S], 0, 0, 0);
        co_mfma_set<WID, S + 1>(xa, xb, acc);
    }
}

// CEN: the columns are centred on the way into LDS (a template parameter, not a run-time branch: with the test inside
// the load loop the uncentred kernel went from 3.37 to 4.93 ms, profiles/r5_bench_cfg4_kernel_stats.txt of the first run)
// ODD: an odd number of columns -- the pair of columns that straddles the end of a row holds the next row's first
// entry in its second half (the padded column: its tiles are dropped), except behind the LAST row, where nothing may
// be read: that one pair is fetched as a single element.
template <bool CEN, bool ODD>
__global__ __launch_bounds__(CO_THREADS) __attribute__((amdgpu_waves_per_eu(3)))
void syrk_co_kernel(const double *__restrict__ X, int64_t n, int64_t m, int n_cols,
                    const double *__restrict__ d, int n_items, unsigned *__restrict__ counter,
                    double *__restrict__ part, double *__restrict__ cpart, WgLogBuf *__restrict__ log,
                    const unsigned *__restrict__ only_if, const double *__restrict__ center) {
    // (only_if: the int8 syrk's hand-over -- this launch does the work only when the weights were
    // screened OUT of the int8 kernel's envelope, syrk_i8.hip)
    if (only_if != nullptr && *only_if == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double *lds = reinterpret_cast<double *>(smem_raw);          // [2][CO_RS][CO_LDW]
    double *dl = lds + 2 * CO_CHUNK;                             // [2][CO_RS]
    unsigned *slot = reinterpret_cast<unsigned *>(dl + 2 * CO_RS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long t_begin = wg_log_begin(log);

    co_acc_t acc[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) acc[s] = co_acc_t{0, 0, 0, 0};
    double cs0 = 0.0, cs1 = 0.0;          // column sums of d * X: virtual blocks 2 wave, 2 wave + 1

    co_vec2 stage[3];
    double dstage = 0.0;
    // center != NULL: the columns are centred on the way into LDS (x - c; this thread always stages the same
    // two columns), so the product and the column sums are those of X - 1 c' (StandardizedMatrix.sandwich,
    // standardized_mat.py:123-172, without the mean-sized cancellation)
    // (the centres sit in LDS and are read where a chunk is staged: held in registers over the chunk loop they
    // were the four registers too many for the 168 this kernel is capped at)
    double *cl = reinterpret_cast<double *>(slot + 4);          // [CO_W]
    if constexpr (CEN) {
        if (tid < CO_W) cl[tid] = tid < n_cols ? center[tid] : 0.0;
    }

    // ---- the workgroup's chunk stream: items of CO_CPI chunks, ids from the atomic counter
    unsigned idL = blockIdx.x;            // item of the next chunk to load
    int oL = 0;                           // its chunk offset inside the item
    unsigned idNext;                      // the item after idL
    if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
    __syncthreads();
    idNext = *slot;
    bool pending = false;

    auto load_chunk = [&]() -> unsigned {
        const unsigned id = idL;
        const int64_t tb = (int64_t)id * CO_ITEM_ROWS + (int64_t)oL * CO_RS;
        if (tid < CO_RS) {
            const int64_t t = tb + tid;
            dstage = t < n ? d[t] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = tid + i * CO_THREADS;
            const int r = q >> 6, c = (q & 63) * 2;
            const int64_t t = tb + r;
            co_vec2 v = co_vec2{0.0, 0.0};
            if (ODD && t + 1 >= n && c + 1 >= n_cols) {
                if (t < n && c < n_cols) v[0] = X[t * m + c];
            } else if (t < n && c < n_cols) {
                v = __builtin_nontemporal_load(reinterpret_cast<const co_vec2 *>(X + t * m + c));
            }
            stage[i] = v;
        }
        if (++oL == CO_CPI) {             // the stream moves on to the next item
            oL = 0;
            idL = idNext;
            if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
            pending = true;               // idNext is re-read behind the next barrier
        }
        return id;
    };
    auto store_chunk = [&](int buf) {
        double *lb = lds + buf * CO_CHUNK;
        if (tid < CO_RS) dl[buf * CO_RS + tid] = dstage;
        co_vec2 cen = co_vec2{0.0, 0.0};
        if constexpr (CEN) cen = *reinterpret_cast<const co_vec2 *>(cl + (tid & 63) * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int q = tid + i * CO_THREADS;
            // (rows beyond n hold 0 - c: their d is 0, so they add nothing)
            *reinterpret_cast<co_vec2 *>(lb + (q >> 6) * CO_LDW + (q & 63) * 2) = CEN ? stage[i] - cen : stage[i];
        }
    };

    // fragments of one row group: lane (k = lane >> 4, i = lane & 15) reads row 4 g + k, the 16
    // bytes at columns 32 q + 2 i, + 1 -> virtual blocks 2 q (even) and 2 q + 1 (odd), position i
    auto read_frag = [&](const double *lb, const double *db, int g, double (&xb)[8], double &dv) {
        const int rl = 4 * g + (lane >> 4);
        dv = db[rl];
        const double *lrow = lb + rl * CO_LDW + 2 * (lane & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const co_vec2 v = *reinterpret_cast<const co_vec2 *>(lrow + 32 * q);
            xb[2 * q] = v[0];
            xb[2 * q + 1] = v[1];
        }
    };

    auto run = [&](auto wid) {
        constexpr int WID = decltype(wid)::value;
        unsigned id_c = load_chunk();          // chunk 0
        store_chunk(0);
        unsigned id_c1 = load_chunk();         // chunk 1 (in registers)
        __syncthreads();
        if (pending) { idNext = *slot; pending = false; }
        int buf = 0;
        while (id_c < (unsigned)n_items) {
            const double *lb = lds + buf * CO_CHUNK;
            const double *db = dl + buf * CO_RS;
            double xb0[8], xb1[8], xa[8], dv0, dv1;
            read_frag(lb, db, 0, xb0, dv0);
            read_frag(lb, db, 1, xb1, dv1);
            // group 0
#pragma unroll
            for (int b = 0; b < 8; ++b) xa[b] = dv0 * xb0[b];
            cs0 += xa[2 * WID];
            cs1 += xa[2 * WID + 1];
            co_mfma_set<WID, 0>(xa, xb0, acc);
            // staging of the next chunk between the groups: the LDS writes and the global loads
            // of the chunk after it overlap with the matrix pipe
            store_chunk(buf ^ 1);
            const unsigned id_c2 = load_chunk();
            read_frag(lb, db, 2, xb0, dv0);
            // group 1
#pragma unroll
            for (int b = 0; b < 8; ++b) xa[b] = dv1 * xb1[b];
            cs0 += xa[2 * WID];
            cs1 += xa[2 * WID + 1];
            co_mfma_set<WID, 0>(xa, xb1, acc);
            // group 2
#pragma unroll
            for (int b = 0; b < 8; ++b) xa[b] = dv0 * xb0[b];
            cs0 += xa[2 * WID];
            cs1 += xa[2 * WID + 1];
            co_mfma_set<WID, 0>(xa, xb0, acc);
            __syncthreads();
            if (pending) { idNext = *slot; pending = false; }
            id_c = id_c1;
            id_c1 = id_c2;
            buf ^= 1;
        }
        // ---- partial tiles [t][16][16] (virtual indices) and column sums
        double *dst = part + (int64_t)blockIdx.x * (CO_T * 256);
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int t = WID + s * CO_NWAVES;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[t * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[s][r];
        }
        // lanes i, i + 16, i + 32, i + 48 hold the sums over rows = 0..3 mod 4
        cs0 += __shfl_xor(cs0, 16, 64);
        cs0 += __shfl_xor(cs0, 32, 64);
        cs1 += __shfl_xor(cs1, 16, 64);
        cs1 += __shfl_xor(cs1, 32, 64);
        if (lane < 16) {
            double *cd = cpart + (int64_t)blockIdx.x * CO_W;
            cd[32 * WID + 2 * lane] = cs0;
            cd[32 * WID + 2 * lane + 1] = cs1;
        }
    };
    if (wave == 0) run(std::integral_constant<int, 0>{});
    else if (wave == 1) run(std::integral_constant<int, 1>{});
    else if (wave == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
    if (tid == 0) wg_log_end(log, t_begin, WG_SYRK_CO);
}

// Sum of the partial tiles in a fixed order, un-permuted and mirrored into out; a quarter tile per
// block (64 elements), thread (e, s) sums partials s, s + 16, ...
__global__ __launch_bounds__(1024) void syrk_co_finish_kernel(const double *__restrict__ part,
                                                              int nblk, int n_cols,
                                                              double *__restrict__ out, int64_t ldo,
                                                              const unsigned *__restrict__ only_if) {
    if (only_if != nullptr && *only_if == 0) return;
    __shared__ double red[16][64];
    const int e = blockIdx.y * 64 + threadIdx.x, s = threadIdx.y, t = blockIdx.x;
    double a = 0.0;
    for (int b = s; b < nblk; b += 16) a += part[((int64_t)b * CO_T + t) * 256 + e];
    red[s][threadIdx.x] = a;
    __syncthreads();
    if (s == 0) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; w += 4)
            v += (red[w][threadIdx.x] + red[w + 1][threadIdx.x]) +
                 (red[w + 2][threadIdx.x] + red[w + 3][threadIdx.x]);
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        const int bj = t - bi * (bi + 1) / 2;
        const int ci = co_actual_col(16 * bi + (e >> 4)), cj = co_actual_col(16 * bj + (e & 15));
        // (a diagonal tile holds (i, j) and (j, i) as separately rounded sums: the lower one is
        // mirrored, like every other tile, so that the result is exactly symmetric)
        if (ci < n_cols && cj < n_cols && (bi != bj || (e >> 4) >= (e & 15))) {
            out[(int64_t)ci * ldo + cj] = v;
            if (ci != cj) out[(int64_t)cj * ldo + ci] = v;
        }
    }
}

__global__ __launch_bounds__(CO_W) void syrk_co_colsum_kernel(const double *__restrict__ cpart,
                                                              int nblk, int n_cols,
                                                              double *__restrict__ colsum,
                                                              const unsigned *__restrict__ only_if) {
    if (only_if != nullptr && *only_if == 0u) return;
    const int c = threadIdx.x;
    double a = 0.0;
    for (int b = 0; b < nblk; ++b) a += cpart[(int64_t)b * CO_W + c];
    if (c < n_cols) colsum[c] = a;
}

// bytes of workspace a call needs (grid-dependent upper bound)
size_t syrk_co_ws_bytes() {
    const size_t grid = (size_t)std::max<int64_t>(1, tune("co_grid", 3 * NUM_CU));
    return 256 + sizeof(double) * grid * (CO_T * 256 + CO_W);
}

// ldx / ldo: row strides (in elements) of X and out -- a 128-column panel of a wider block runs in place
static int run_syrk_co_impl(const double *X, int64_t ldx, int64_t n, int64_t m, const double *d, double *out,
                            int64_t ldo, double *colsum, const unsigned *only_if, void *ws_given,
                            hipStream_t st, const double *center) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    TM_REQUIRE(m == 0 || syrk_co_ok(X, m),
               "the co-resident syrk takes a 16-byte aligned C-ordered block of <= 128 columns");
    if (m == 0) return TM_OK;
    if (n == 0) {
        TM_HIP(hipMemset2DAsync(out, sizeof(double) * (size_t)ldo, 0, sizeof(double) * (size_t)m, (size_t)m, st));
        if (colsum) TM_HIP(hipMemsetAsync(colsum, 0, sizeof(double) * (size_t)m, st));
        return TM_OK;
    }
    const int64_t n_items64 = ceil_div(n, CO_ITEM_ROWS);
    TM_REQUIRE(n_items64 < (1ll << 31), "too many rows");
    const int n_items = (int)n_items64;
    const int grid = (int)std::min<int64_t>(n_items, tune("co_grid", 3 * NUM_CU));
    const size_t part_bytes = sizeof(double) * (size_t)grid * CO_T * 256;
    const size_t cpart_bytes = sizeof(double) * (size_t)grid * CO_W;
    void *wsv = ws_given;          // (a caller that keeps live data in the stream's workspace passes its own region)
    if (wsv == nullptr) {
        int rc = get_workspace(256 + part_bytes + cpart_bytes, &wsv, st);
        if (rc) return rc;
    }
    unsigned *counter = reinterpret_cast<unsigned *>(wsv);
    double *part = reinterpret_cast<double *>(reinterpret_cast<char *>(wsv) + 256);
    double *cpart = part + (size_t)grid * CO_T * 256;
    TM_HIP(hipMemsetAsync(counter, 0, 256, st));
    auto kern = (m & 1) ? (center ? &syrk_co_kernel<true, true> : &syrk_co_kernel<false, true>)
                        : (center ? &syrk_co_kernel<true, false> : &syrk_co_kernel<false, false>);
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)CO_LDS));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(CO_THREADS), CO_LDS, st, X, n, ldx,
                       (int)m, d, n_items, counter, part, cpart, wg_log_ptr(), only_if, center);
    prof_end(st);
    TM_LAUNCH_CHECK();
    hipLaunchKernelGGL(syrk_co_finish_kernel, dim3(CO_T, 4), dim3(64, 16), 0, st, part, grid, (int)m,
                       out, ldo, only_if);
    TM_LAUNCH_CHECK();
    if (colsum) {
        hipLaunchKernelGGL(syrk_co_colsum_kernel, dim3(1), dim3(CO_W), 0, st, cpart, grid, (int)m,
                           colsum, only_if);
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

int run_syrk_co(const double *X, int64_t n, int64_t m, const double *d, double *out, double *colsum,
                hipStream_t st, const double *center) {
    return run_syrk_co_impl(X, m, n, m, d, out, m, colsum, nullptr, nullptr, st, center);
}

// the same launches, live only when *flag != 0 (device memory): the int8 syrk's fallback
int run_syrk_co_flagged(const double *X, int64_t ldx, int64_t n, int64_t m, const double *d, double *out,
                        int64_t ldo, double *colsum, const unsigned *flag, void *ws, hipStream_t st,
                        const double *center) {
    return run_syrk_co_impl(X, ldx, n, m, d, out, ldo, colsum, flag, ws, st, center);
}

}  // namespace tmh

using namespace tmh;

extern "C" {

int tm_dense_sandwich_co_f64(const double *X, int64_t n, int64_t m, const double *d, double *out,
                             double *colsum, void *stream) {
    return run_syrk_co(X, n, m, d, out, colsum, as_stream(stream));
}

int tm_dense_sandwich_co_centered_f64(const double *X, int64_t n, int64_t m, const double *d, const double *center,
                                      double *out, double *colsum, void *stream) {
    return run_syrk_co(X, n, m, d, out, colsum, as_stream(stream), center);
}

}  // extern "C"
