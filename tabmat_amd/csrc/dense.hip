// Dense-block kernels for gfx950.
//
//   K1  dense sandwich  out = X[rows,cols]^T diag(d[rows]) X[rows,cols]
//       (reference: ext/dense_helpers-tmpl.cpp:266-311) as an MFMA row-weighted syrk:
//       every workgroup streams a contiguous slab of rows through LDS (coalesced 16-byte
//       global loads, double-buffered), each of its 4 waves keeps a fixed subset of the
//       lower-triangular 16x16 output tiles in MFMA accumulators for the whole slab
//       (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, K = 4 rows per instruction), and
//       the per-workgroup partial k x k matrices are summed in a second, deterministic pass.
//       Only tiles with bj <= bi are computed (the reference's jmaxinner, 148-151).
//   K5  restricted / unrestricted dense matvec and transpose-matvec
//       (reference: ext/dense_helpers-tmpl.cpp:314-417 and the BLAS gemv of
//       dense_matrix.py:212-217): HBM-bound streaming kernels.
#include <algorithm>
#include <utility>

#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

// -------------------------------------------------------------------------------------------
// MFMA traits.  A/B operands are one element per lane: A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15]; so for X^T D X the fragment of a 16-column block for a group of
// 4 rows is simply  frag[lane] = X[row0 + (lane>>4)][16*b + (lane&15)]  for both operands
// (the A side additionally scaled by d[row]).
// -------------------------------------------------------------------------------------------
template <typename F>
struct Mfma;

template <>
struct Mfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    static __device__ __forceinline__ int crow(int lane, int reg) { return (lane >> 4) + 4 * reg; }
};

template <>
struct Mfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = 4 * (lane >> 4) + reg
    static __device__ __forceinline__ int crow(int lane, int reg) { return 4 * (lane >> 4) + reg; }
};

constexpr int tri_row(int t) {
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= t) ++r;
    return r;
}

enum LoadMode { LOAD_C_VEC = 0, LOAD_C_GEN = 1, LOAD_F_GEN = 2, LOAD_F_VEC = 3 };

// RECT = false: lower-triangular tiles of one NBLK-block panel (syrk proper).
// RECT = true : the NBLK/2 x NBLK/2 off-diagonal tiles (row blocks NBLK/2.., column blocks
//               0..NBLK/2-1) of a virtual panel made of two 8-block column panels.
template <bool RECT, int NBLK>
constexpr int tile_bi(int t) { return RECT ? NBLK / 2 + t / (NBLK / 2) : tri_row(t); }
template <bool RECT, int NBLK>
constexpr int tile_bj(int t) {
    return RECT ? t % (NBLK / 2) : t - tri_row(t) * (tri_row(t) + 1) / 2;
}

template <typename F, int NBLK, bool RECT = false>
struct SyrkCfg {
    static constexpr int NWAVES = 4;
    static constexpr int THREADS = NWAVES * 64;
    static constexpr int W = NBLK * 16;                 // padded number of columns
    static constexpr int LDW = W + 16;                  // LDS row stride (bank-conflict free)
    static constexpr int RS = (W * (int)sizeof(F) >= 1024) ? 16 : 32;  // rows per chunk
    static constexpr int T = RECT ? (NBLK / 2) * (NBLK / 2) : NBLK * (NBLK + 1) / 2;  // tiles
    static constexpr int MAXT = (T + NWAVES - 1) / NWAVES;
    static constexpr int VEC = 16 / (int)sizeof(F);
    static constexpr int ELEMS = RS * W;
    static constexpr int PER_THREAD = ELEMS / THREADS;  // elements staged per thread (scalar modes)
    static constexpr int NVEC = ELEMS / VEC;            // 16-byte vectors per chunk
    static constexpr int VITER = (NVEC + THREADS - 1) / THREADS;
    static constexpr int NSTAGE = (PER_THREAD > VITER * VEC) ? PER_THREAD : VITER * VEC;
    // LOAD_F_VEC keeps the chunk COLUMN-major in LDS ([W][LDR], rows fastest) so that the 16-byte
    // vectors of an F-ordered X (2 / 4 consecutive rows of one column) are stored as they come.
    // LDR = RS + 2 (f64) / RS + 4 (f32): the MFMA fragment read lane -> (column 16b + (lane&15),
    // row 4g + (lane>>4)) is then bank-conflict free (18 i + g, resp. 20 i + g, distinct mod 32 / 64).
    static constexpr int LDR = RS + (sizeof(F) == 8 ? 2 : 4);
    static constexpr int CHUNK_ELEMS = (RS * LDW > W * LDR) ? RS * LDW : W * LDR;
    static constexpr size_t LDS = sizeof(F) * (size_t)(2 * CHUNK_ELEMS + 2 * RS);
};

template <typename F, int NBLK, bool RECT, int WID, int... S>
__device__ __forceinline__ void syrk_wave_step(
    const F (&xa)[NBLK], const F (&xb)[NBLK],
    typename Mfma<F>::acc_t (&acc)[SyrkCfg<F, NBLK, RECT>::MAXT], std::integer_sequence<int, S...>) {
    using C = SyrkCfg<F, NBLK, RECT>;
    (([&] {
         constexpr int t = WID + S * C::NWAVES;
         if constexpr (t < C::T) {
             constexpr int bi = tile_bi<RECT, NBLK>(t);
             constexpr int bj = tile_bj<RECT, NBLK>(t);
             acc[S] = Mfma<F>::mma(xa[bi], xb[bj], acc[S]);
         }
     }()),
     ...);
}

template <typename F, int NBLK, bool RECT, int WID, int... S>
__device__ __forceinline__ void syrk_wave_store(
    const typename Mfma<F>::acc_t (&acc)[SyrkCfg<F, NBLK, RECT>::MAXT], F *__restrict__ dst,
    int lane, std::integer_sequence<int, S...>) {
    using C = SyrkCfg<F, NBLK, RECT>;
    (([&] {
         constexpr int t = WID + S * C::NWAVES;
         if constexpr (t < C::T) {
#pragma unroll
             for (int r = 0; r < 4; ++r)
                 dst[t * 256 + Mfma<F>::crow(lane, r) * 16 + (lane & 15)] = acc[S][r];
         }
     }()),
     ...);
}

// HALF = 0 / 1: first / second half of the chunk's row groups (the staging of the next chunk is
// issued between the two halves, see syrk_kernel).
template <typename F, int NBLK, bool RECT, int WID, int HALF, bool FORD>
__device__ __forceinline__ void syrk_wave_main(
    const F *__restrict__ lbuf, const F *__restrict__ dbuf,
    typename Mfma<F>::acc_t (&acc)[SyrkCfg<F, NBLK, RECT>::MAXT], int lane) {
    using C = SyrkCfg<F, NBLK, RECT>;
    constexpr int NGH = C::RS / 8;          // row groups (of 4 rows) per half
    constexpr int GB = HALF * NGH;
    constexpr int NG = GB + NGH;
    // No fragment prefetch across row groups: it would cost the second wave per SIMD (the kernel
    // is tuned to <= 256 registers, amdgpu_waves_per_eu(2)), and two resident workgroups per CU
    // hide the LDS latency better than a software pipeline inside one wave does
    // (cfg2: 7.4 ms with one wave per SIMD and prefetch, 5.8 ms with two waves and none).
#pragma unroll 2
    for (int g = GB; g < NG; ++g) {
        const int rl = 4 * g + (lane >> 4);
        const F dv = dbuf[rl];
        const F *lrow = FORD ? lbuf + (lane & 15) * C::LDR + rl : lbuf + rl * C::LDW + (lane & 15);
        F xa[NBLK], xb[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            xb[b] = lrow[FORD ? 16 * b * C::LDR : 16 * b];
            xa[b] = dv * xb[b];
        }
        syrk_wave_step<F, NBLK, RECT, WID>(xa, xb, acc, std::make_integer_sequence<int, C::MAXT>{});
    }
}

// One workgroup: rows [blockIdx.x * rows_per_block, +rows_per_block) of the row list.
// XT (round 5): the extras of the load stage -- per-column centres and the odd-width row-end pair -- are compiled in
// only where a call needs them (run-time tests in the load loop cost the plain kernel 6-9 % at 32 / 64 columns).
template <typename F, int NBLK, int MODE, bool RECT, bool XT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void syrk_kernel(const F *__restrict__ X, int64_t n, int64_t m,
                                                   const F *__restrict__ d,
                                                   const int32_t *__restrict__ rows,
                                                   int64_t n_iter, int64_t rows_per_block,
                                                   const int32_t *__restrict__ cols, int n_cols,
                                                   F *__restrict__ ws, int64_t coff0, int64_t coff1,
                                                   const F *__restrict__ center) {
    // center (may be NULL; indexed by the column of X): the columns are centred on the way into LDS, x - c: the
    // product of X - 1 c' (what StandardizedMatrix.sandwich is after, standardized_mat.py:123-172, without
    // subtracting mean-sized rank-one terms from the raw product).  Uniform branches: the uncentred path is
    // the code it was.
    // coff0 / coff1 (LOAD_C_VEC only): first column of X behind the virtual columns 0..127 and
    // 128..255 -- a 128-column panel of a wider block, or the two panels of a rectangular pass,
    // read with the same 16-byte loads as a whole block (0 / 128 = the block itself)
    using C = SyrkCfg<F, NBLK, RECT>;
    using acc_t = typename Mfma<F>::acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr bool FORD = MODE == LOAD_F_VEC;
    F *lds = reinterpret_cast<F *>(smem_raw);            // [2][RS][LDW]  (LOAD_F_VEC: [2][W][LDR])
    F *dl = lds + 2 * C::CHUNK_ELEMS;                    // [2][RS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);
    const int nchunk = (int)((t1 - t0 + C::RS - 1) / C::RS);

    acc_t acc[C::MAXT];
#pragma unroll
    for (int s = 0; s < C::MAXT; ++s) acc[s] = acc_t{0, 0, 0, 0};

    F stage[C::NSTAGE];
    F dstage = F(0);

    auto load_chunk = [&](int ch) {
        const int64_t tb = t0 + (int64_t)ch * C::RS;
        if (tid < C::RS) {
            const int64_t t = tb + tid;
            dstage = F(0);
            if (t < t1) dstage = d[rows ? (int64_t)rows[t] : t];
        }
        if (MODE == LOAD_C_VEC) {
            // 16-byte vector loads, row-major: q -> (row, vec-column)
            constexpr int VPR = C::W / C::VEC;
#pragma unroll
            for (int i = 0; i < C::VITER; ++i) {
                const int q = tid + i * C::THREADS;
                const int r = q / VPR;
                const int c = (q % VPR) * C::VEC;
                const int64_t t = tb + r;
                typedef F vec_t __attribute__((ext_vector_type(C::VEC)));
                vec_t v;
#pragma unroll
                for (int e = 0; e < C::VEC; ++e) v[e] = F(0);
                if (q < C::NVEC && t < t1 && c < n_cols) {
                    const int64_t row = rows ? (int64_t)rows[t] : t;
                    // (streamed once: nontemporal, 3.68 -> 3.60 ms at cfg4)
                    const int64_t xc = c < 128 ? coff0 + c : coff1 + c - 128;
                    if (XT && sizeof(F) == 8 && (m & 1) && xc + C::VEC > m && row + 1 >= n) {
                        // odd width (round 5): the pair that straddles the end of the LAST row is one element
                        // (elsewhere its second half is the next row's first entry: the padded column, dropped)
                        v[0] = X[row * m + xc];
                    } else {
                        v = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(X + row * m + xc));
                    }
                    if (XT && center != nullptr) {
                        if (sizeof(F) == 8 && (m & 1) && xc + C::VEC > m) v[0] -= center[xc];
                        else v -= *reinterpret_cast<const vec_t *>(center + xc);
                    }
                }
#pragma unroll
                for (int e = 0; e < C::VEC; ++e) stage[i * C::VEC + e] = v[e];
            }
        } else if (MODE == LOAD_F_VEC) {
            // 16-byte vector loads down the columns of an F-ordered X: q -> (column, row vector)
            constexpr int VPC = C::RS / C::VEC;
#pragma unroll
            for (int i = 0; i < C::VITER; ++i) {
                const int q = tid + i * C::THREADS;
                const int c = q / VPC;
                const int64_t t = tb + (q % VPC) * C::VEC;
                typedef F vec_t __attribute__((ext_vector_type(C::VEC)));
                vec_t v;
#pragma unroll
                for (int e = 0; e < C::VEC; ++e) v[e] = F(0);
                if (q < C::NVEC && c < n_cols) {
                    const F *src = X + (int64_t)c * n + t;
                    if (t + C::VEC <= t1) {
                        v = *reinterpret_cast<const vec_t *>(src);
                    } else {
#pragma unroll
                        for (int e = 0; e < C::VEC; ++e)
                            if (t + e < t1) v[e] = src[e];
                    }
                    if (XT && center != nullptr) {
                        const F cc = center[c];
#pragma unroll
                        for (int e = 0; e < C::VEC; ++e)
                            if (t + e < t1) v[e] -= cc;
                    }
                }
#pragma unroll
                for (int e = 0; e < C::VEC; ++e) stage[i * C::VEC + e] = v[e];
            }
        } else if (MODE == LOAD_C_GEN) {
#pragma unroll
            for (int i = 0; i < C::PER_THREAD; ++i) {
                const int q = tid + i * C::THREADS;
                const int r = q / C::W;
                const int c = q % C::W;
                const int64_t t = tb + r;
                F v = F(0);
                if (t < t1 && c < n_cols) {
                    const int64_t row = rows ? (int64_t)rows[t] : t;
                    const int64_t col = cols ? (int64_t)cols[c] : c;
                    v = X[row * m + col];
                    if (XT && center != nullptr) v -= center[col];
                }
                stage[i] = v;
            }
        } else {  // LOAD_F_GEN: consecutive threads walk down one column
#pragma unroll
            for (int i = 0; i < C::PER_THREAD; ++i) {
                const int q = tid + i * C::THREADS;
                const int c = q / C::RS;
                const int r = q % C::RS;
                const int64_t t = tb + r;
                F v = F(0);
                if (t < t1 && c < n_cols) {
                    const int64_t row = rows ? (int64_t)rows[t] : t;
                    const int64_t col = cols ? (int64_t)cols[c] : c;
                    v = X[col * n + row];
                    if (XT && center != nullptr) v -= center[col];
                }
                stage[i] = v;
            }
        }
    };

    auto store_chunk = [&](int buf) {
        F *lb = lds + buf * C::CHUNK_ELEMS;
        if (tid < C::RS) dl[buf * C::RS + tid] = dstage;
        if (MODE == LOAD_F_VEC) {
            constexpr int VPC = C::RS / C::VEC;
#pragma unroll
            for (int i = 0; i < C::VITER; ++i) {
                const int q = tid + i * C::THREADS;
                typedef F vec_t __attribute__((ext_vector_type(C::VEC)));
                vec_t v;
#pragma unroll
                for (int e = 0; e < C::VEC; ++e) v[e] = stage[i * C::VEC + e];
                if (q < C::NVEC)
                    *reinterpret_cast<vec_t *>(lb + (q / VPC) * C::LDR + (q % VPC) * C::VEC) = v;
            }
        } else if (MODE == LOAD_C_VEC) {
            constexpr int VPR = C::W / C::VEC;
#pragma unroll
            for (int i = 0; i < C::VITER; ++i) {
                const int q = tid + i * C::THREADS;
                const int r = q / VPR;
                const int c = (q % VPR) * C::VEC;
                typedef F vec_t __attribute__((ext_vector_type(C::VEC)));
                vec_t v;
#pragma unroll
                for (int e = 0; e < C::VEC; ++e) v[e] = stage[i * C::VEC + e];
                if (q < C::NVEC) *reinterpret_cast<vec_t *>(lb + r * C::LDW + c) = v;
            }
        } else if (MODE == LOAD_C_GEN) {
#pragma unroll
            for (int i = 0; i < C::PER_THREAD; ++i) {
                const int q = tid + i * C::THREADS;
                lb[(q / C::W) * C::LDW + (q % C::W)] = stage[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < C::PER_THREAD; ++i) {
                const int q = tid + i * C::THREADS;
                lb[(q % C::RS) * C::LDW + (q / C::RS)] = stage[i];
            }
        }
    };

    // One barrier per chunk.  The staging of chunk ch + 1 (registers -> LDS, other buffer) and the
    // global loads of chunk ch + 2 are issued BETWEEN the two halves of chunk ch's MFMA work, so
    // the LDS writes overlap with the matrix pipe instead of sitting in front of the barrier, and
    // every global load has a whole chunk of compute to land.
    // The whole chunk loop is instantiated once per wave id (wave-uniform dispatch to the
    // statically scheduled tile set of the wave, taken ONCE): the accumulators then stay in place
    // for the whole kernel; every wave executes the same number of barriers.
    F *dst = ws + (int64_t)blockIdx.x * (C::T * 256);  // [tile][16][16]
    auto run = [&](auto wid) {
        constexpr int WID = decltype(wid)::value;
        if (nchunk > 0) {
            load_chunk(0);
            store_chunk(0);
            if (nchunk > 1) load_chunk(1);
        }
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int buf = ch & 1;
            const F *lb = lds + buf * C::CHUNK_ELEMS;
            const F *db = dl + buf * C::RS;
            syrk_wave_main<F, NBLK, RECT, WID, 0, FORD>(lb, db, acc, lane);
            if (ch + 1 < nchunk) store_chunk(buf ^ 1);
            if (ch + 2 < nchunk) load_chunk(ch + 2);
            syrk_wave_main<F, NBLK, RECT, WID, 1, FORD>(lb, db, acc, lane);
            __syncthreads();
        }
        syrk_wave_store<F, NBLK, RECT, WID>(acc, dst, lane, std::make_integer_sequence<int, C::MAXT>{});
    };
    if (wave == 0) run(std::integral_constant<int, 0>{});
    else if (wave == 1) run(std::integral_constant<int, 1>{});
    else if (wave == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
}

// Sum the per-workgroup partial tiles in fixed order (double accumulation): a quarter tile per
// block (64 elements), thread (e, s) sums partials b = s, s + 16, ...; 4 x T blocks of 16 waves
// (36 blocks of 4 x 256 threads left most of the chip idle: 52 us for 38 MB).
template <typename F>
__global__ __launch_bounds__(1024) void syrk_reduce_kernel(const F *__restrict__ part, int nblk,
                                                           int T, F *__restrict__ tmp) {
    __shared__ double red[16][64];
    const int e = blockIdx.y * 64 + threadIdx.x, s = threadIdx.y, t = blockIdx.x;
    double acc = 0.0;
    for (int b = s; b < nblk; b += 16) acc += (double)part[((int64_t)b * T + t) * 256 + e];
    red[s][threadIdx.x] = acc;
    __syncthreads();
    if (s == 0) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; w += 4)
            v += (red[w][threadIdx.x] + red[w + 1][threadIdx.x]) +
                 (red[w + 2][threadIdx.x] + red[w + 3][threadIdx.x]);
        tmp[t * 256 + e] = (F)v;
    }
}

// out[pos[i] * ldo + pos[j]] = tile-major tmp at (max(i,j), min(i,j))   (mirror + scatter)
template <typename F>
__global__ void syrk_mirror_kernel(const F *__restrict__ tmp, int nv,
                                   const int32_t *__restrict__ pos, F *__restrict__ out,
                                   int64_t ldo) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= nv || j >= nv) return;
    const int hi = max(i, j), lo = min(i, j);
    const int bi = hi >> 4, bj = lo >> 4;
    const int t = bi * (bi + 1) / 2 + bj;
    const int64_t oi = pos ? pos[i] : i;
    const int64_t oj = pos ? pos[j] : j;
    out[oi * ldo + oj] = tmp[t * 256 + (hi & 15) * 16 + (lo & 15)];
}

// RECT result (virtual rows half..half+nb-1 x virtual cols 0..na-1) -> both off-diagonal blocks
template <typename F>
__global__ void syrk_rect_scatter_kernel(const F *__restrict__ tmp, int half, int na, int nb,
                                         const int32_t *__restrict__ pos, F *__restrict__ out,
                                         int64_t ldo) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;   // column in panel a
    const int i = blockIdx.y;                              // row in panel b
    if (i >= nb || j >= na) return;
    const int nbh = half >> 4;
    const int t = (i >> 4) * nbh + (j >> 4);
    const F v = tmp[t * 256 + (i & 15) * 16 + (j & 15)];
    const int64_t oi = pos[half + i], oj = pos[j];
    out[oi * ldo + oj] = v;
    out[oj * ldo + oi] = v;
}

template <typename F, int NBLK, bool RECT = false>
static int launch_syrk(const F *X, int64_t n, int64_t m, int order_f, const F *d,
                       const int32_t *rows, int64_t n_iter, const int32_t *cols, int n_cols,
                       const int32_t *pos, F *out, int64_t ldo, char *wsbase, size_t ws_off,
                       hipStream_t st, int64_t coff0 = 0, int64_t coff1 = 128, const F *center = nullptr) {
    using C = SyrkCfg<F, NBLK, RECT>;
    const int blocks_per_cu = (2 * C::LDS <= LDS_BYTES) ? 2 : 1;
    int64_t nblk = std::min<int64_t>((int64_t)NUM_CU * blocks_per_cu,
                                     std::max<int64_t>(1, ceil_div(n_iter, 4 * C::RS)));
    int64_t rpb = ceil_div(ceil_div(n_iter, nblk), C::RS) * C::RS;
    nblk = ceil_div(n_iter, rpb);
    F *part = reinterpret_cast<F *>(wsbase + ws_off);   // [nblk][T][256]
    F *tmp = part + (size_t)nblk * C::T * 256;          // [T][256]
    // (float64 rows of an odd width start at 8-byte aligned addresses every other row: the 16-byte loads take
    // that -- 4M x 63: 1.2 ms through the element loads, see profiles/r5_odd_widths.txt)
    const bool vec_ok = !order_f && cols == nullptr && (m % C::VEC == 0 || sizeof(F) == 8) &&
                        ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    // (the 256-column f32 panel has no registers to spare for the column-major addressing)
    const bool fvec_ok = order_f && cols == nullptr && rows == nullptr && (n % C::VEC == 0) &&
                         ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                         !(sizeof(F) == 4 && NBLK == 16 && !RECT);
    const int mode = order_f ? (fvec_ok ? LOAD_F_VEC : LOAD_F_GEN) : (vec_ok ? LOAD_C_VEC : LOAD_C_GEN);
    auto go = [&](auto kern) -> int {
        if (C::LDS > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS));
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(C::THREADS), C::LDS, st, X, n, m, d,
                           rows, n_iter, rpb, cols, n_cols, part, coff0, coff1, center);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return TM_OK;
    };
    int rc;
    const bool xt = center != nullptr || (sizeof(F) == 8 && (m & 1) && mode == LOAD_C_VEC);
    if (xt) {
        if (mode == LOAD_C_VEC) rc = go(&syrk_kernel<F, NBLK, LOAD_C_VEC, RECT, true>);
        else if (mode == LOAD_C_GEN) rc = go(&syrk_kernel<F, NBLK, LOAD_C_GEN, RECT, true>);
        else if (mode == LOAD_F_VEC) rc = go(&syrk_kernel<F, NBLK, LOAD_F_VEC, RECT, true>);
        else rc = go(&syrk_kernel<F, NBLK, LOAD_F_GEN, RECT, true>);
    } else {
        if (mode == LOAD_C_VEC) rc = go(&syrk_kernel<F, NBLK, LOAD_C_VEC, RECT, false>);
        else if (mode == LOAD_C_GEN) rc = go(&syrk_kernel<F, NBLK, LOAD_C_GEN, RECT, false>);
        else if (mode == LOAD_F_VEC) rc = go(&syrk_kernel<F, NBLK, LOAD_F_VEC, RECT, false>);
        else rc = go(&syrk_kernel<F, NBLK, LOAD_F_GEN, RECT, false>);
    }
    if (rc) return rc;
    hipLaunchKernelGGL((syrk_reduce_kernel<F>), dim3(C::T, 4), dim3(64, 16), 0, st, part, (int)nblk,
                       C::T, tmp);
    TM_LAUNCH_CHECK();
    if (RECT) {
        const int half = C::W / 2;
        hipLaunchKernelGGL((syrk_rect_scatter_kernel<F>),
                           dim3((unsigned)ceil_div(half, 64), (unsigned)(n_cols - half)), dim3(64), 0,
                           st, tmp, half, half, n_cols - half, pos, out, ldo);
    } else {
        hipLaunchKernelGGL((syrk_mirror_kernel<F>),
                           dim3((unsigned)ceil_div(n_cols, 64), (unsigned)n_cols), dim3(64), 0, st,
                           tmp, n_cols, pos, out, ldo);
    }
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static size_t syrk_ws_bytes(int W) {
    return sizeof(F) * (size_t)(2 * NUM_CU + 1) * W * W + 4096;
}

template <typename F>
static int syrk_dispatch(const F *X, int64_t n, int64_t m, int order_f, const F *d,
                         const int32_t *rows, int64_t n_iter, const int32_t *cols, int n_cols,
                         const int32_t *pos, F *out, int64_t ldo, char *wsbase, size_t ws_off,
                         hipStream_t st, int64_t coff0 = 0, const F *center = nullptr) {
    if (n_cols <= 16)
        return launch_syrk<F, 1>(X, n, m, order_f, d, rows, n_iter, cols, n_cols, pos, out, ldo,
                                 wsbase, ws_off, st, coff0, coff0 + 128, center);
    if (n_cols <= 32)
        return launch_syrk<F, 2>(X, n, m, order_f, d, rows, n_iter, cols, n_cols, pos, out, ldo,
                                 wsbase, ws_off, st, coff0, coff0 + 128, center);
    if (n_cols <= 64)
        return launch_syrk<F, 4>(X, n, m, order_f, d, rows, n_iter, cols, n_cols, pos, out, ldo,
                                 wsbase, ws_off, st, coff0, coff0 + 128, center);
    if (n_cols <= 128)
        return launch_syrk<F, 8>(X, n, m, order_f, d, rows, n_iter, cols, n_cols, pos, out, ldo,
                                 wsbase, ws_off, st, coff0, coff0 + 128, center);
    if constexpr (sizeof(F) == 4) {
        return launch_syrk<F, 16>(X, n, m, order_f, d, rows, n_iter, cols, n_cols, pos, out, ldo,
                                  wsbase, ws_off, st, coff0, coff0 + 128, center);
    } else {
        set_error("internal: f64 syrk panel wider than 128 columns");
        return TM_EINVAL;
    }
}

__global__ void iota_or_copy_i32_kernel(int32_t *dst, const int32_t *src, int32_t start, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[start + i] : start + i;
}

template <typename F>
static int run_dense_sandwich(const F *X, int64_t n, int64_t m, int order_f, const F *d,
                              const int32_t *rows, int64_t n_rows, const int32_t *cols,
                              int64_t n_cols_in, F *out, hipStream_t st, const F *center = nullptr) {
    // center (may be NULL, length m, indexed by the column of X): the product of X - 1 center'
    const int64_t n_cols = cols ? n_cols_in : m;
    const int64_t n_iter = rows ? n_rows : n;
    if (n_cols == 0) return TM_OK;
    if (n_iter == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)(n_cols * n_cols), st));
        return TM_OK;
    }
    TM_REQUIRE(n_cols < (1 << 20), "too many columns");
    // a handful of columns: one lane per pair of columns, no tiles (syrk_narrow.hip)
    if (rows == nullptr && cols == nullptr && center == nullptr && syrk_narrow_ok(m) && tune("syrk_narrow", 1))
        return run_syrk_narrow<F>(X, n, m, order_f, d, out, st);
    if constexpr (sizeof(F) == 8) {
        // unrestricted C-ordered f64 block of <= 128 columns: the LDS-light kernel with fragment
        // prefetch and dynamic work items (syrk_co.hip; 3.2-3.4 ms against 3.6-3.7 ms at cfg4)
        if (!order_f && rows == nullptr && cols == nullptr && syrk_co_ok(X, m) && syrk_co_pays(m) &&
            tune("syrk_co", 1))
            return run_syrk_co(X, n, m, d, out, nullptr, st, center);
    }
    if constexpr (sizeof(F) == 4) {
        // unrestricted C-ordered f32 block of 129 .. 256 columns: three-piece bf16 split on the bf16
        // matrix cores (syrk_bf16.hip; 16x the rate of the f32-input MFMA, f32 accuracy)
        if (!order_f && rows == nullptr && cols == nullptr && center == nullptr && m > 128 &&
            syrk_bf16x3_ok(X, m) && tune("syrk_bf16", 1))
            return run_syrk_bf16x3(X, n, m, d, out, st);
    }
    void *wsv = nullptr;
    const int direct_max = sizeof(F) == 4 ? 256 : 128;
    if (n_cols <= direct_max) {
        const int W = n_cols <= 16 ? 16 : n_cols <= 32 ? 32 : n_cols <= 64 ? 64 : n_cols <= 128 ? 128 : 256;
        int rc = get_workspace(syrk_ws_bytes<F>(W), &wsv, st);
        if (rc) return rc;
        return syrk_dispatch<F>(X, n, m, order_f, d, rows, n_iter, cols, (int)n_cols, nullptr, out,
                                n_cols, reinterpret_cast<char *>(wsv), 0, st, 0, center);
    }
    // wide blocks: 128-column panels.  Diagonal panels are lower-triangular syrks; every panel
    // pair (a < b) is one rectangular pass over the 256 virtual columns [panel a | panel b].
    const int PW = 128;
    const int np = (int)ceil_div(n_cols, PW);
    const size_t idx_bytes = 4096;  // 2 x 256 int32 (virtual cols, positions) + slack
    int rc = get_workspace(idx_bytes + syrk_ws_bytes<F>(256), &wsv, st);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(wsv);
    int32_t *vcols = reinterpret_cast<int32_t *>(base);
    int32_t *vpos = vcols + 256;
    // all columns of a C-ordered, aligned block: the panels are column OFFSETS for the 16-byte-load
    // kernels (a column list would send them down the element-wise load path: 5.9 instead of
    // 3.4 ms for 2M x 256 f64)
    constexpr int VECW = 16 / (int)sizeof(F);
    const bool contiguous = cols == nullptr && !order_f && (m % VECW == 0 || sizeof(F) == 8) &&
                            (reinterpret_cast<uintptr_t>(X) & 15) == 0;
    for (int a = 0; a < np; ++a) {
        const int wa = (int)std::min<int64_t>(PW, n_cols - (int64_t)a * PW);
        hipLaunchKernelGGL(iota_or_copy_i32_kernel, dim3(1), dim3(128), 0, st, vcols, cols, a * PW,
                           wa);
        hipLaunchKernelGGL(iota_or_copy_i32_kernel, dim3(1), dim3(128), 0, st, vpos,
                           (const int32_t *)nullptr, a * PW, wa);
        TM_LAUNCH_CHECK();
        rc = syrk_dispatch<F>(X, n, m, order_f, d, rows, n_iter, contiguous ? nullptr : vcols, wa, vpos,
                              out, n_cols, base, idx_bytes, st, contiguous ? (int64_t)a * PW : 0, center);
        if (rc) return rc;
        for (int b = a + 1; b < np; ++b) {
            const int wb = (int)std::min<int64_t>(PW, n_cols - (int64_t)b * PW);
            hipLaunchKernelGGL(iota_or_copy_i32_kernel, dim3(1), dim3(128), 0, st, vcols + PW, cols,
                               b * PW, wb);
            hipLaunchKernelGGL(iota_or_copy_i32_kernel, dim3(1), dim3(128), 0, st, vpos + PW,
                               (const int32_t *)nullptr, b * PW, wb);
            TM_LAUNCH_CHECK();
            rc = launch_syrk<F, 16, true>(X, n, m, order_f, d, rows, n_iter, contiguous ? nullptr : vcols,
                                          PW + wb, vpos, out, n_cols, base, idx_bytes, st,
                                          contiguous ? (int64_t)a * PW : 0,
                                          contiguous ? (int64_t)b * PW : 128, center);
            if (rc) return rc;
        }
    }
    return TM_OK;
}

// Wide float64 blocks (130 .. 512 even columns, C order, unrestricted): 128-column panels.  The DIAGONAL
// panels run on the int8 matrix cores in place (K1e through its row strides; every panel has its own
// envelope check and f64 hand-over), the off-diagonal panel pairs on the rectangular f64 MFMA tile set.
// 2M x 256: two int8 panels + one rectangle instead of three f64 passes (profiles/r4_regimes.txt).
static int run_dense_sandwich_i8_wide(const double *X, int64_t n, int64_t m, const double *d,
                                      const double *colmax, double *out, hipStream_t st,
                                      const double *center = nullptr) {
    TM_REQUIRE(m > 128 && m <= 512 && m % 2 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0,
               "a 16-byte aligned C-ordered float64 block of 130 .. 512 (even) columns");
    if (n == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(double) * (size_t)(m * m), st));
        return TM_OK;
    }
    const int PW = 128;
    const int np = (int)ceil_div(m, PW);
    for (int a = 0; a < np; ++a) {
        const int wa = (int)std::min<int64_t>(PW, m - (int64_t)a * PW);
        int rc = run_syrk_i8_panel(X + (int64_t)a * PW, m, n, wa, d, colmax + (int64_t)a * PW,
                                   out + ((int64_t)a * PW) * m + (int64_t)a * PW, m, nullptr, nullptr, st,
                                   center ? center + (int64_t)a * PW : nullptr);
        if (rc) return rc;
    }
    void *wsv = nullptr;
    const size_t idx_bytes = 4096;
    int rc = get_workspace(idx_bytes + syrk_ws_bytes<double>(256), &wsv, st);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(wsv);
    int32_t *vpos = reinterpret_cast<int32_t *>(base) + 256;
    for (int a = 0; a < np; ++a) {
        hipLaunchKernelGGL(iota_or_copy_i32_kernel, dim3(1), dim3(128), 0, st, vpos, (const int32_t *)nullptr,
                           a * PW, PW);
        for (int b = a + 1; b < np; ++b) {
            const int wb = (int)std::min<int64_t>(PW, m - (int64_t)b * PW);
            hipLaunchKernelGGL(iota_or_copy_i32_kernel, dim3(1), dim3(128), 0, st, vpos + PW,
                               (const int32_t *)nullptr, b * PW, wb);
            TM_LAUNCH_CHECK();
            rc = launch_syrk<double, 16, true>(X, n, m, 0, d, nullptr, n, nullptr, PW + wb, vpos, out, m, base,
                                               idx_bytes, st, (int64_t)a * PW, (int64_t)b * PW, center);
            if (rc) return rc;
        }
    }
    return TM_OK;
}

// -------------------------------------------------------------------------------------------
// K5  matvec / rmatvec
// -------------------------------------------------------------------------------------------
// C-order matvec: one wave per output row, lanes stride over the selected columns.
template <typename F>
__global__ __launch_bounds__(256) void dense_matvec_c_kernel(
    const F *__restrict__ X, int64_t m, const F *__restrict__ v, const int32_t *__restrict__ rows,
    int64_t n_iter, const int32_t *__restrict__ cols, int n_cols, F *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t t = wave0; t < n_iter; t += nw) {
        const int64_t row = rows ? (int64_t)rows[t] : t;
        const F *xr = X + row * m;
        F acc = F(0);
        for (int c = lane; c < n_cols; c += 64) {
            const int64_t j = cols ? (int64_t)cols[c] : c;
            acc += xr[j] * v[j];
        }
        acc = wave_sum(acc);
        if (lane == 0) out[t] += acc;
    }
}

// C-order fast path (all rows, all columns, 16-byte aligned rows): a wave streams MV_R rows per
// step with one 16-byte load per lane and row segment, so MV_R KiB-sized loads are in flight per
// wave; the MV_R row sums are then reduced across the lanes with a halving butterfly (the number
// of live values halves with every exchange: 7 + 3 shuffles for 8 rows instead of 6 per row).
// LPR = lanes per row (m / VEC when that is a power of two <= 64; 64 with NL loads per row else).
constexpr int MV_R = 8;

template <typename F, int LPR, int NL>
__global__ __launch_bounds__(256) void dense_matvec_c_stream_kernel(const F *__restrict__ X,
                                                                   int64_t n, int64_t m,
                                                                   const F *__restrict__ v,
                                                                   F *__restrict__ out) {
    constexpr int VEC = 16 / (int)sizeof(F);
    constexpr int RPL = 64 / LPR;                      // rows covered by one wave-wide load
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    const int lane = threadIdx.x & 63;
    const int seg = lane / LPR;                        // row within a load
    const int sl = lane % LPR;
    vec_t vv[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) vv[q] = *reinterpret_cast<const vec_t *>(v + ((int64_t)q * 64 + sl) * VEC);
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    constexpr int ROWS = MV_R * RPL;                   // rows per wave step
    const int64_t nstep = (n + ROWS - 1) / ROWS;
    for (int64_t t = wave0; t < nstep; t += nw) {
        const int64_t r0 = t * ROWS;
        F acc[MV_R];
        vec_t x[MV_R][NL];
#pragma unroll
        for (int r = 0; r < MV_R; ++r) {
            const int64_t row = min(r0 + r * RPL + seg, n - 1);
#pragma unroll
            for (int q = 0; q < NL; ++q)
                x[r][q] = __builtin_nontemporal_load(
                    reinterpret_cast<const vec_t *>(X + row * m + ((int64_t)q * 64 + sl) * VEC));
        }
#pragma unroll
        for (int r = 0; r < MV_R; ++r) {
            F a = F(0);
#pragma unroll
            for (int q = 0; q < NL; ++q)
#pragma unroll
                for (int e = 0; e < VEC; ++e) a = fma(x[r][q][e], vv[q][e], a);
            acc[r] = a;
        }
        // Halving butterfly inside each LPR-lane segment.  The three halving exchanges (4 + 2 + 1
        // values) use the lane strides 1, 2, 4 and the remaining plain sums the strides 8 .. LPR/2,
        // because strides < 16 are DPP moves on the VALU (quad_perm, row_half_mirror, row_ror)
        // while larger ones go through the LDS crossbar (ds_bpermute): 2 crossbar exchanges per 8
        // rows instead of 10.  After the step with stride s the lanes with bit s set own the upper
        // half of the remaining row values.
        static_assert(MV_R == 8 && LPR >= 8, "three halving steps at strides 1, 2, 4");
        int sel = 0;
        {
            const bool up = (lane & 1) != 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const F keep = up ? acc[i + 4] : acc[i];
                const F send = up ? acc[i] : acc[i + 4];
                acc[i] = keep + dpp_xor<1>(send);
            }
            sel += up ? 4 : 0;
        }
        {
            const bool up = (lane & 2) != 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const F keep = up ? acc[i + 2] : acc[i];
                const F send = up ? acc[i] : acc[i + 2];
                acc[i] = keep + dpp_xor<2>(send);
            }
            sel += up ? 2 : 0;
        }
        {
            const bool up = (lane & 4) != 0;
            const F keep = up ? acc[1] : acc[0];
            const F send = up ? acc[0] : acc[1];
            acc[0] = keep + dpp_xor<4>(send);
            sel += up ? 1 : 0;
        }
        if (LPR >= 16) acc[0] += dpp_xor<8>(acc[0]);
#pragma unroll
        for (int s = 16; s <= LPR / 2; s <<= 1) acc[0] += __shfl_xor(acc[0], s, 64);
        // every lane of the segment with the same low 3 bits now holds the sum of row `sel`
        const int64_t row = r0 + (int64_t)sel * RPL + seg;
        if (sl < 8 && row < n) out[row] += acc[0];
    }
}

// C-order matvec for row lengths the streaming kernel has no lane split for (any m up to
// MV_TILE_MAX_M, all rows, all columns, 16-byte aligned X).  A row of 10 doubles is 80 bytes: one
// wave per row keeps 80 bytes per wave in flight and the launch runs at 0.38 TB/s.  Here a block
// copies TR whole rows (TR * m contiguous elements) from HBM to LDS with flat 16-byte loads, so
// the loads are coalesced whatever m is, then TPR lanes per row sum the products from LDS.  The
// LDS rows are padded to an odd number of elements: lanes of different rows hit different banks.
constexpr int MV_TILE_MAX_M = 1280;
constexpr int MV_TILE_BYTES = 40960;
constexpr int MV_TILE_V = MV_TILE_BYTES / 16 / 256;    // 16-byte vectors per lane and tile

template <typename F>
__global__ __launch_bounds__(256) void dense_matvec_c_tile_kernel(const F *__restrict__ X, int64_t n,
                                                                 int m, int tpr_log2, int tr,
                                                                 const F *__restrict__ v,
                                                                 F *__restrict__ out) {
    constexpr int VEC = 16 / (int)sizeof(F);
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    extern __shared__ __attribute__((aligned(16))) unsigned char mv_tile_smem[];
    const int ms = m | 1;                              // padded row length in LDS
    const int tpr = 1 << tpr_log2;
    const int rstep = 256 >> tpr_log2;                 // rows summed at a time; tr = rows per tile
    F *vl = reinterpret_cast<F *>(mv_tile_smem);
    F *tile = vl + ((m + 1) & ~1);
    for (int j = threadIdx.x; j < m; j += 256) vl[j] = v[j];
    // flat position of this lane's first vector inside a tile, and the step between its vectors;
    // both are the same for every tile because a tile starts at a row boundary
    const int e0 = threadIdx.x * VEC;
    const int r_first = e0 / m, c_first = e0 - r_first * m;
    const int dr = (256 * VEC) / m, dc = (256 * VEC) - dr * m;
    const int my_row = threadIdx.x >> tpr_log2, my_s = threadIdx.x & (tpr - 1);
    const int64_t ntiles = (n + tr - 1) / tr;
    vec_t x[MV_TILE_V];
    // the tile's vectors go to registers one tile ahead: they are in flight while the previous
    // tile is summed from LDS
    auto fetch = [&](int64_t t) {
        const int64_t row0 = t * tr;
        const int nvec = (int)(min((int64_t)tr, n - row0) * m) / VEC;
        const vec_t *src = reinterpret_cast<const vec_t *>(X + row0 * (int64_t)m);
#pragma unroll
        for (int u = 0; u < MV_TILE_V; ++u) {
            const int q = (int)threadIdx.x + u * 256;
            if (q < nvec) x[u] = __builtin_nontemporal_load(src + q);
        }
    };
    if ((int64_t)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t row0 = t * tr;
        const int rows_here = (int)min((int64_t)tr, n - row0);
        const int elems = rows_here * m;
        const int nvec = elems / VEC;
        __syncthreads();                               // previous tile fully read (and vl written)
        int r = r_first, c = c_first;
#pragma unroll
        for (int u = 0; u < MV_TILE_V; ++u) {
            if ((int)threadIdx.x + u * 256 < nvec) {
                int rr = r, cc = c;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    tile[rr * ms + cc] = x[u][e];
                    if (++cc == m) { cc = 0; ++rr; }
                }
            }
            r += dr; c += dc;
            if (c >= m) { c -= m; ++r; }
        }
        // the elems % VEC elements a whole vector does not cover (partial last tile only)
        if ((int)threadIdx.x < elems - nvec * VEC) {
            const int e = nvec * VEC + threadIdx.x;
            const int rr = e / m;
            tile[rr * ms + (e - rr * m)] = X[row0 * (int64_t)m + e];
        }
        __syncthreads();
        if (t + gridDim.x < ntiles) fetch(t + gridDim.x);
        for (int row = my_row; row < rows_here; row += rstep) {
            const F *lr = tile + row * ms;
            F acc = F(0);
            for (int j = my_s; j < m; j += tpr) acc = fma(lr[j], vl[j], acc);
            for (int s = 1; s < tpr; s <<= 1) acc += __shfl_xor(acc, s, 64);
            if (my_s == 0) out[row0 + row] += acc;
        }
    }
}

// C-order transpose_matvec for narrow or oddly sized rows (any m <= RMV_FLAT_MAX_M, all rows, all
// columns, 16-byte aligned X).  The streaming kernel gives every lane one 16-byte piece of a row,
// so a 10-column matrix uses 5 lanes of 64.  Here the lanes walk the slab's elements as one flat
// array of 16-byte vectors.  With P = m / gcd(m, VEC) vectors per column period and A = the
// largest multiple of P <= 256 active lanes, lane t reads the vectors t, t + A, t + 2A, ...: their
// elements always fall in the same VEC columns, so the lane keeps VEC running sums and only the
// row index (for v[row]) moves, by A * VEC / m whole rows per step.
constexpr int RMV_FLAT_MAX_M = 1024;
constexpr int RMV_FLAT_U = 4;
constexpr int64_t RMV_SLAB_BYTES = 128 * 1024;
constexpr int64_t RMV_FLAT_V_BYTES = 32 * 1024;

template <typename F>
__global__ __launch_bounds__(256) void dense_rmatvec_c_flat_kernel(
    const F *__restrict__ X, int64_t n, int m, int active, const F *__restrict__ v,
    int64_t rows_per_block, F *__restrict__ out) {
    constexpr int VEC = 16 / (int)sizeof(F);
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ F red[256 * VEC];
    extern __shared__ __attribute__((aligned(16))) unsigned char rmv_flat_smem[];
    F *vs = reinterpret_cast<F *>(rmv_flat_smem);       // the slab's part of v
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;     // a multiple of VEC
    const int64_t t1 = min(t0 + rows_per_block, n);
    const int tid = threadIdx.x;
    for (int i = tid; i < (int)(t1 - t0); i += 256) vs[i] = v[t0 + i];
    __syncthreads();
    const int drow = active * VEC / m;                 // whole rows per step
    vec_t acc;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = F(0);
    if (tid < active) {
        int ro[VEC];                                   // row (relative to the step's first row) of
        {                                              // each element of this lane's vector
            int rr = (tid * VEC) / m, cc = tid * VEC - rr * m;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                ro[e] = rr;
                if (++cc == m) { cc = 0; ++rr; }
            }
        }
        const int64_t total = (t1 - t0) * m;           // elements of the slab
        const F *src = X + t0 * (int64_t)m;
        const int64_t nrow = t1 - t0;
        const int64_t estep = (int64_t)active * VEC;
        int64_t e0 = (int64_t)tid * VEC;
        int rbase = 0;
        // whole steps: every lane's vector and rows are inside the slab
        for (; e0 + (RMV_FLAT_U - 1) * estep + VEC <= total &&
               rbase + (RMV_FLAT_U - 1) * drow + ro[VEC - 1] < nrow;
             e0 += RMV_FLAT_U * estep, rbase += RMV_FLAT_U * drow) {
            vec_t x[RMV_FLAT_U];
            F w[RMV_FLAT_U][VEC];
#pragma unroll
            for (int u = 0; u < RMV_FLAT_U; ++u) {
                x[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(src + e0 + u * estep));
#pragma unroll
                for (int e = 0; e < VEC; ++e) w[u][e] = vs[rbase + u * drow + ro[e]];
            }
#pragma unroll
            for (int u = 0; u < RMV_FLAT_U; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[e] = fma(x[u][e], w[u][e], acc[e]);
        }
        // the slab's last steps, element by element
        for (; e0 < total; e0 += estep, rbase += drow) {
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                if (e0 + e < total) acc[e] = fma(src[e0 + e], vs[rbase + ro[e]], acc[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[tid * VEC + e] = acc[e];
    __syncthreads();
    // flat element i of the active lanes belongs to column i % m: fold the active * VEC sums, a
    // whole number of rows, to one row (by 8 per round, at most 256 / m rows wide)
    int cnt = active * VEC;
    if (m <= 256) {
        const int maxg = 256 / m;
        while (cnt > m) {
            const int nw = min(maxg, (cnt / m + 7) / 8) * m;
            F s = F(0);
            if (tid < nw)
                for (int i = tid; i < cnt; i += nw) s += red[i];
            __syncthreads();
            if (tid < nw) red[tid] = s;
            __syncthreads();
            cnt = nw;
        }
        if (tid < m) atomic_add(&out[tid], red[tid]);
    } else {
        for (int j = tid; j < m; j += 256) {
            F s = F(0);
            for (int i = j; i < cnt; i += m) s += red[i];
            atomic_add(&out[j], s);
        }
    }
}

// F-order matvec: one thread per output row.
template <typename F>
__global__ __launch_bounds__(256) void dense_matvec_f_kernel(
    const F *__restrict__ X, int64_t n, const F *__restrict__ v, const int32_t *__restrict__ rows,
    int64_t n_iter, const int32_t *__restrict__ cols, int n_cols, F *__restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_iter; t += stride) {
        const int64_t row = rows ? (int64_t)rows[t] : t;
        F acc = F(0);
        for (int c = 0; c < n_cols; ++c) {
            const int64_t j = cols ? (int64_t)cols[c] : c;
            acc += X[j * n + row] * v[j];
        }
        out[t] += acc;
    }
}

// C-order rmatvec: block owns a slab of rows; lane <-> column (64-column tiles), the 4 waves
// interleave rows; LDS combine, one global atomic per (block, column).
template <typename F>
__global__ __launch_bounds__(256) void dense_rmatvec_c_kernel(
    const F *__restrict__ X, int64_t m, const F *__restrict__ v, const int32_t *__restrict__ rows,
    int64_t n_iter, int64_t rows_per_block, const int32_t *__restrict__ cols, int n_cols,
    F *__restrict__ out) {
    __shared__ F red[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);
    for (int c0 = 0; c0 < n_cols; c0 += 64) {
        const int c = c0 + lane;
        const int64_t j = c < n_cols ? (cols ? (int64_t)cols[c] : c) : 0;
        F acc = F(0);
        if (c < n_cols) {
#pragma unroll 4
            for (int64_t t = t0 + wave; t < t1; t += 4) {
                const int64_t row = rows ? (int64_t)rows[t] : t;
                acc += X[row * m + j] * v[row];
            }
        }
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && c < n_cols)
            atomic_add(&out[c], (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
        __syncthreads();
    }
}

// C-order fast path (all rows, all columns, 16-byte aligned rows, 64 * VEC columns per pass):
// a wave streams MV_R rows per step with one 16-byte load per lane and row (lane <-> VEC adjacent
// columns), the 4 waves of a block interleave row groups; v[row] is wave-uniform.  SQ: weighted
// squared deviations (K7) instead of the plain product.
template <typename F, bool SQ>
__global__ __launch_bounds__(256) void dense_rmatvec_c_stream_kernel(
    const F *__restrict__ X, int64_t n, int64_t m, const F *__restrict__ v,
    const F *__restrict__ shift, int64_t rows_per_block, F *__restrict__ out) {
    constexpr int VEC = 16 / (int)sizeof(F);
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ F red[4][64 * VEC];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n);
    for (int64_t c0 = 0; c0 < m; c0 += 64 * VEC) {
        const int64_t c = c0 + (int64_t)lane * VEC;
        const bool cok = c < m;                      // m % VEC == 0: a vector is all in or all out
        vec_t acc, sh;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { acc[e] = F(0); sh[e] = F(0); }
        if (SQ && cok) sh = *reinterpret_cast<const vec_t *>(shift + c);
        if (cok) {
            for (int64_t r0 = t0 + (int64_t)wave * MV_R; r0 < t1; r0 += 4 * MV_R) {
                vec_t x[MV_R];
                F w[MV_R];
#pragma unroll
                for (int r = 0; r < MV_R; ++r) {
                    const int64_t row = min(r0 + r, t1 - 1);
                    x[r] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(X + row * m + c));
                    w[r] = r0 + r < t1 ? v[row] : F(0);
                }
#pragma unroll
                for (int r = 0; r < MV_R; ++r)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const F xv = SQ ? x[r][e] - sh[e] : x[r][e];
                        acc[e] = fma(SQ ? xv * xv : xv, w[r], acc[e]);
                    }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[wave][lane * VEC + e] = acc[e];
        __syncthreads();
        if (wave == 0 && cok) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int q = lane * VEC + e;
                atomic_add(&out[c + e], (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]));
            }
        }
        __syncthreads();
    }
}

// F-order rmatvec: (column, row-slab) per wave, lanes stride down the column.
template <typename F>
__global__ __launch_bounds__(256) void dense_rmatvec_f_kernel(
    const F *__restrict__ X, int64_t n, const F *__restrict__ v, const int32_t *__restrict__ rows,
    int64_t n_iter, int64_t rows_per_block, const int32_t *__restrict__ cols, int n_cols,
    F *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);
    for (int c = wave; c < n_cols; c += 4) {
        const int64_t j = cols ? (int64_t)cols[c] : c;
        const F *xc = X + j * n;
        F acc = F(0);
        for (int64_t t = t0 + lane; t < t1; t += 64) {
            const int64_t row = rows ? (int64_t)rows[t] : t;
            acc += xc[row] * v[row];
        }
        acc = wave_sum(acc);
        if (lane == 0) atomic_add(&out[c], acc);
    }
}

// F-order fast path (all rows, all columns, 16-byte aligned columns): a block owns a slab of
// FRS rows whose v (and, for K7, nothing else) is staged in LDS once; its 4 waves then take the
// columns in turn and stream the slab part of each column with 16-byte nontemporal loads (lane
// <-> VEC consecutive rows, 4 loads in flight), multiply with v from LDS, reduce across the wave
// and issue one atomic per (block, column).  SQ: weighted squared deviations (K7).
constexpr int FRS = 8192;

template <typename F, bool SQ>
__global__ __launch_bounds__(256) void dense_rmatvec_f_stream_kernel(
    const F *__restrict__ X, int64_t n, int64_t m, const F *__restrict__ v,
    const F *__restrict__ shift, F *__restrict__ out) {
    constexpr int VEC = 16 / (int)sizeof(F);
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ __attribute__((aligned(16))) F vl[FRS];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * FRS;
    const int nr = (int)min((int64_t)FRS, n - t0);          // rows of this slab (n % VEC == 0)
    for (int i = threadIdx.x * VEC; i < FRS; i += 256 * VEC) {
        vec_t w;
#pragma unroll
        for (int e = 0; e < VEC; ++e) w[e] = F(0);
        if (i < nr) w = *reinterpret_cast<const vec_t *>(v + t0 + i);
        *reinterpret_cast<vec_t *>(vl + i) = w;
    }
    __syncthreads();
    constexpr int UN = 4;
    for (int64_t c = wave; c < m; c += 4) {
        const F *xc = X + c * n + t0;
        const F sh = SQ ? shift[c] : F(0);
        F acc = F(0);
        for (int r0 = lane * VEC; r0 < nr; r0 += 64 * VEC * UN) {
            vec_t x[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int r = min(r0 + u * 64 * VEC, nr - VEC);
                x[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(xc + r));
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int r = r0 + u * 64 * VEC;
                if (r < nr) {
                    const vec_t w = *reinterpret_cast<const vec_t *>(vl + r);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const F xv = SQ ? x[u][e] - sh : x[u][e];
                        acc = fma(SQ ? xv * xv : xv, w[e], acc);
                    }
                }
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) atomic_add(&out[c], acc);
    }
}

// -------------------------------------------------------------------------------------------
// K7  out[j] += sum_i w[i] * (X[i, j] - shift[j])^2      (ext/dense.pyx:103-122,
// transpose_square_dot_weights; used once per standardize()).  Same decomposition as rmatvec.
// -------------------------------------------------------------------------------------------
template <typename F, bool ORDER_F>
__global__ __launch_bounds__(256) void dense_col_sq_dev_kernel(
    const F *__restrict__ X, int64_t n, int64_t m, const F *__restrict__ w,
    const F *__restrict__ shift, int64_t rows_per_block, F *__restrict__ out) {
    __shared__ F red[4][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n);
    if (ORDER_F) {
        for (int64_t c = wave; c < m; c += 4) {
            const F sh = shift[c];
            const F *xc = X + c * n;
            F acc = F(0);
            for (int64_t t = t0 + lane; t < t1; t += 64) {
                const F x = xc[t] - sh;
                acc += w[t] * x * x;
            }
            acc = wave_sum(acc);
            if (lane == 0) atomic_add(&out[c], acc);
        }
        return;
    }
    for (int64_t c0 = 0; c0 < m; c0 += 64) {
        const int64_t c = c0 + lane;
        F acc = F(0);
        if (c < m) {
            const F sh = shift[c];
#pragma unroll 4
            for (int64_t t = t0 + wave; t < t1; t += 4) {
                const F x = X[t * m + c] - sh;
                acc += w[t] * x * x;
            }
        }
        red[wave][lane] = acc;
        __syncthreads();
        if (wave == 0 && c < m)
            atomic_add(&out[c], (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
        __syncthreads();
    }
}

template <typename F>
static int run_dense_col_sq_dev(const F *X, int64_t n, int64_t m, int order_f, const F *w,
                                const F *shift, F *out, hipStream_t st) {
    if (n == 0 || m == 0) return TM_OK;
    int64_t nblk = std::min<int64_t>(std::max<int64_t>(1, ceil_div(n, 512)), NUM_CU * 4);
    const int64_t rpb = ceil_div(n, nblk);
    nblk = ceil_div(n, rpb);
    constexpr int VEC = 16 / (int)sizeof(F);
    if (!order_f && m % VEC == 0 &&
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(shift)) & 15) == 0) {
        hipLaunchKernelGGL((dense_rmatvec_c_stream_kernel<F, true>), dim3((unsigned)nblk), dim3(256), 0,
                           st, X, n, m, w, shift, rpb, out);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    if (order_f && n % VEC == 0 && n >= VEC &&
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(w)) & 15) == 0) {
        hipLaunchKernelGGL((dense_rmatvec_f_stream_kernel<F, true>), dim3((unsigned)ceil_div(n, FRS)),
                           dim3(256), 0, st, X, n, m, w, shift, out);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    if (order_f)
        hipLaunchKernelGGL((dense_col_sq_dev_kernel<F, true>), dim3((unsigned)nblk), dim3(256), 0, st,
                           X, n, m, w, shift, rpb, out);
    else
        hipLaunchKernelGGL((dense_col_sq_dev_kernel<F, false>), dim3((unsigned)nblk), dim3(256), 0,
                           st, X, n, m, w, shift, rpb, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_dense_matvec(const F *X, int64_t n, int64_t m, int order_f, const F *v,
                            const int32_t *rows, int64_t n_rows, const int32_t *cols,
                            int64_t n_cols_in, F *out, hipStream_t st) {
    const int64_t n_cols = cols ? n_cols_in : m;
    const int64_t n_iter = rows ? n_rows : n;
    if (n_iter == 0 || n_cols == 0) return TM_OK;
    if (order_f) {
        const int64_t nblk = std::min<int64_t>(ceil_div(n_iter, 256), NUM_CU * 8);
        prof_begin(st);
        hipLaunchKernelGGL((dense_matvec_f_kernel<F>), dim3((unsigned)nblk), dim3(256), 0, st, X, n,
                           v, rows, n_iter, cols, (int)n_cols, out);
        prof_end(st);
    } else {
        constexpr int VEC = 16 / (int)sizeof(F);
        const bool aligned = ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
        const int64_t lpr = m / VEC;
        if (!rows && !cols && aligned && m % VEC == 0 &&
            (lpr == 64 || lpr == 128 || lpr == 32 || lpr == 16 || lpr == 8)) {
            const int64_t rows_per_step = MV_R * (lpr >= 64 ? 1 : 64 / lpr);
            const int64_t nblk = std::min<int64_t>(ceil_div(ceil_div(n, rows_per_step), 4), NUM_CU * 8);
            prof_begin(st);
#define TM_MV_LAUNCH(LPR_, NL_)                                                               \
    hipLaunchKernelGGL((dense_matvec_c_stream_kernel<F, LPR_, NL_>), dim3((unsigned)nblk),     \
                       dim3(256), 0, st, X, n, m, v, out)
            if (lpr == 128) TM_MV_LAUNCH(64, 2);
            else if (lpr == 64) TM_MV_LAUNCH(64, 1);
            else if (lpr == 32) TM_MV_LAUNCH(32, 1);
            else if (lpr == 16) TM_MV_LAUNCH(16, 1);
            else TM_MV_LAUNCH(8, 1);
#undef TM_MV_LAUNCH
            prof_end(st);
            TM_LAUNCH_CHECK();
            return TM_OK;
        }
        if (!rows && !cols && aligned && m <= MV_TILE_MAX_M) {
            // rows per tile: as many as MV_TILE_BYTES hold, a power of two between 4 and 256
            const int64_t ms = m | 1;
            int tpr_log2 = 0;
            while ((256 >> tpr_log2) * ms * (int64_t)sizeof(F) > MV_TILE_BYTES && tpr_log2 < 6) ++tpr_log2;
            // (narrow rows: several rows per lane)
            const int tr = (256 >> tpr_log2) * (int)std::max<int64_t>(1, MV_TILE_BYTES / (256 * ms * (int64_t)sizeof(F)));
            const size_t lds = (size_t)(tr * ms + ((m + 1) & ~(int64_t)1)) * sizeof(F);
            const int64_t per_cu = std::min<int64_t>(8, (144 * 1024) / (int64_t)lds);
            const int64_t nblk = std::min<int64_t>(ceil_div(n, tr), NUM_CU * per_cu);
            prof_begin(st);
            hipLaunchKernelGGL((dense_matvec_c_tile_kernel<F>), dim3((unsigned)nblk), dim3(256), lds, st,
                               X, n, (int)m, tpr_log2, tr, v, out);
            prof_end(st);
            TM_LAUNCH_CHECK();
            return TM_OK;
        }
        const int64_t nblk = std::min<int64_t>(ceil_div(n_iter, 4), NUM_CU * 8);
        prof_begin(st);
        hipLaunchKernelGGL((dense_matvec_c_kernel<F>), dim3((unsigned)nblk), dim3(256), 0, st, X, m,
                           v, rows, n_iter, cols, (int)n_cols, out);
        prof_end(st);
    }
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_dense_rmatvec(const F *X, int64_t n, int64_t m, int order_f, const F *v,
                             const int32_t *rows, int64_t n_rows, const int32_t *cols,
                             int64_t n_cols_in, F *out, hipStream_t st) {
    const int64_t n_cols = cols ? n_cols_in : m;
    const int64_t n_iter = rows ? n_rows : n;
    if (n_iter == 0 || n_cols == 0) return TM_OK;
    int64_t nblk = std::min<int64_t>(std::max<int64_t>(1, ceil_div(n_iter, 512)), NUM_CU * 4);
    const int64_t rpb = ceil_div(n_iter, nblk);
    nblk = ceil_div(n_iter, rpb);
    constexpr int VEC = 16 / (int)sizeof(F);
    // rows that do not fill the streaming kernel's 64 lanes with whole vectors: the flat kernel
    // keeps all lanes loading
    const int64_t flat_max = tune("rmv_flat_max", RMV_FLAT_MAX_M);
    int g = VEC;                                       // gcd(m, VEC), VEC a power of two
    while (m % g) g >>= 1;
    const int period = (int)std::min<int64_t>(m / g, 1 << 20);   // vectors per column period
    // (measured at 4 GB, profiles/r5_dense_matvec_widths.txt: the streaming kernel is the faster
    // one exactly when one pass of a wave is 32 .. 64 whole vectors)
    const bool one_pass = m % VEC == 0 && m / VEC >= 32 && m / VEC <= 64;
    if (!order_f && !rows && !cols && m <= flat_max && period <= 256 && !one_pass &&
        (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        const int active = (256 / period) * period;
        // a slab of about RMV_SLAB_BYTES per block, a multiple of 32 rows (so that it starts on
        // a 16-byte boundary) and at most RMV_FLAT_ROWS rows (its part of v is staged in LDS)
        constexpr int64_t RMV_FLAT_ROWS = RMV_FLAT_V_BYTES / (int64_t)sizeof(F);
        const int64_t want = std::min<int64_t>(RMV_FLAT_ROWS, std::max<int64_t>(32, RMV_SLAB_BYTES / (m * (int64_t)sizeof(F))));
        int64_t nb = std::min<int64_t>(std::max<int64_t>(1, ceil_div(n, want)), NUM_CU * 8);
        int64_t rp = std::min<int64_t>(RMV_FLAT_ROWS, ceil_div(ceil_div(n, nb), 32) * 32);
        nb = ceil_div(n, rp);
        prof_begin(st);
        hipLaunchKernelGGL((dense_rmatvec_c_flat_kernel<F>), dim3((unsigned)nb), dim3(256),
                           (size_t)rp * sizeof(F), st, X, n, (int)m, active, v, rp, out);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    if (!order_f && !rows && !cols && m % VEC == 0 &&
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        prof_begin(st);
        hipLaunchKernelGGL((dense_rmatvec_c_stream_kernel<F, false>), dim3((unsigned)nblk), dim3(256),
                           0, st, X, n, m, v, (const F *)nullptr, rpb, out);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    if (order_f && !rows && !cols && n % VEC == 0 && n >= VEC &&
        ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        prof_begin(st);
        hipLaunchKernelGGL((dense_rmatvec_f_stream_kernel<F, false>), dim3((unsigned)ceil_div(n, FRS)),
                           dim3(256), 0, st, X, n, m, v, (const F *)nullptr, out);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    prof_begin(st);
    if (order_f)
        hipLaunchKernelGGL((dense_rmatvec_f_kernel<F>), dim3((unsigned)nblk), dim3(256), 0, st, X,
                           n, v, rows, n_iter, rpb, cols, (int)n_cols, out);
    else
        hipLaunchKernelGGL((dense_rmatvec_c_kernel<F>), dim3((unsigned)nblk), dim3(256), 0, st, X,
                           m, v, rows, n_iter, rpb, cols, (int)n_cols, out);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

// -------------------------------------------------------------------------------------------
// SplitMatrix assembly scatter (split_matrix.py:341-354)
// -------------------------------------------------------------------------------------------
template <typename F>
__global__ void scatter_block_kernel(const F *__restrict__ src, int64_t nr, int64_t nc,
                                     const int64_t *__restrict__ ri,
                                     const int64_t *__restrict__ ci, double *__restrict__ out,
                                     int64_t p, int mirror, int diag) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (diag) {
        if (e < nr) out[ri[e] * p + ri[e]] += (double)src[e];
        return;
    }
    if (e >= nr * nc) return;
    const int64_t a = e / nc, b = e % nc;
    const double v = (double)src[e];
    out[ri[a] * p + ci[b]] = v;
    if (mirror) out[ci[b] * p + ri[a]] = v;
}

template <typename F>
static int run_scatter(const F *src, int64_t nr, int64_t nc, const int64_t *ri, const int64_t *ci,
                       double *out, int64_t p, int mirror, int diag, hipStream_t st) {
    const int64_t total = diag ? nr : nr * nc;
    if (total == 0) return TM_OK;
    hipLaunchKernelGGL((scatter_block_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256),
                       0, st, src, nr, nc, ri, ci, out, p, mirror, diag);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

using namespace tmh;

extern "C" {

int tm_dense_sandwich_i8_wide_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                  double *out, void *stream) {
    return tmh::run_dense_sandwich_i8_wide(X, n, m, d, colmax, out, tmh::as_stream(stream));
}

int tm_dense_sandwich_i8_wide_centered_f64(const double *X, int64_t n, int64_t m, const double *d,
                                           const double *colmax, const double *center, double *out, void *stream) {
    return tmh::run_dense_sandwich_i8_wide(X, n, m, d, colmax, out, tmh::as_stream(stream), center);
}

// (X - 1 center')[rows, cols]' diag(d[rows]) (X - 1 center')[rows, cols]; center: length m, indexed by the column of X
int tm_dense_sandwich_centered_f32(const float *X, int64_t n, int64_t m, int order_f, const float *dv,
                                   const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                                   const float *center, float *out, void *stream) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    return run_dense_sandwich<float>(X, n, m, order_f, dv, rows, n_rows, cols, n_cols, out, as_stream(stream), center);
}
int tm_dense_sandwich_centered_f64(const double *X, int64_t n, int64_t m, int order_f, const double *dv,
                                   const int32_t *rows, int64_t n_rows, const int32_t *cols, int64_t n_cols,
                                   const double *center, double *out, void *stream) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    return run_dense_sandwich<double>(X, n, m, order_f, dv, rows, n_rows, cols, n_cols, out, as_stream(stream), center);
}

#define TM_DENSE_ENTRY(NAME, F, RUN)                                                              \
    int NAME(const F *X, int64_t n, int64_t m, int order_f, const F *dv, const int32_t *rows,     \
             int64_t n_rows, const int32_t *cols, int64_t n_cols, F *out, void *stream) {         \
        TM_REQUIRE(n >= 0 && m >= 0, "negative shape");                                           \
        return RUN<F>(X, n, m, order_f, dv, rows, n_rows, cols, n_cols, out, as_stream(stream));  \
    }

TM_DENSE_ENTRY(tm_dense_sandwich_f32, float, run_dense_sandwich)
TM_DENSE_ENTRY(tm_dense_sandwich_f64, double, run_dense_sandwich)
TM_DENSE_ENTRY(tm_dense_matvec_f32, float, run_dense_matvec)
TM_DENSE_ENTRY(tm_dense_matvec_f64, double, run_dense_matvec)
TM_DENSE_ENTRY(tm_dense_rmatvec_f32, float, run_dense_rmatvec)
TM_DENSE_ENTRY(tm_dense_rmatvec_f64, double, run_dense_rmatvec)

int tm_dense_col_sq_dev_f32(const float *X, int64_t n, int64_t m, int order_f, const float *w,
                            const float *shift, float *out, void *stream) {
    return run_dense_col_sq_dev<float>(X, n, m, order_f, w, shift, out, as_stream(stream));
}
int tm_dense_col_sq_dev_f64(const double *X, int64_t n, int64_t m, int order_f, const double *w,
                            const double *shift, double *out, void *stream) {
    return run_dense_col_sq_dev<double>(X, n, m, order_f, w, shift, out, as_stream(stream));
}

int tm_scatter_block_f32(const float *src, int64_t nr, int64_t nc, const int64_t *ri,
                         const int64_t *ci, double *out, int64_t p, int mirror, int diag,
                         void *stream) {
    return run_scatter<float>(src, nr, nc, ri, ci, out, p, mirror, diag, as_stream(stream));
}
int tm_scatter_block_f64(const double *src, int64_t nr, int64_t nc, const int64_t *ri,
                         const int64_t *ci, double *out, int64_t p, int mirror, int diag,
                         void *stream) {
    return run_scatter<double>(src, nr, nc, ri, ci, out, p, mirror, diag, as_stream(stream));
}

}  // extern "C"
