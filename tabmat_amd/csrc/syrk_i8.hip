// K1e  Dense self sandwich  out = X' diag(d) X  for a C-ordered FLOAT64 block of <= 128 columns on the
// INT8 matrix cores (reference: ext/dense_helpers-tmpl.cpp:266-311; the dense term of BASELINE
// configs[3], split_matrix.py:337-354).
//
// v_mfma_f64_16x16x4_f64 runs at the vector f64 rate on gfx950 and sustains 0.6 of it: the f64 syrk
// (syrk_co.hip) needs 3.3 ms for 10M x 128 while the block streams in 1.3 ms and the int8 matrix
// cores (3.9 POPS measured) sit idle.  Ozaki-style slicing moves the product there:
//   * Y = diag(sqrt d) X is written in 40-bit fixed point per column,
//         F[r][i] = round(Y[r][i] * 2^k_i),   |F| <= 2^38,   2^k_i = 2^38 / 2^ceil(log2(colmax_i sqrt(dmax)))
//     by ONE v_fma_f64 with a magic constant (1.5 * 2^52 + 0x8080808080): the low 40 mantissa bits
//     of the result are F + bias, whose five bytes g_0 .. g_4 are the balanced base-256 digits
//     b_s = g_s - 128 in [-128, 127] (two's complement byte: g_s ^ 0x80),  F = sum_s b_s 256^s;
//   * S_ij = 2^-(k_i + k_j) sum_r F_ri F_rj = 2^-(k_i + k_j) sum_{s,t} 256^(s + t) sum_r b_s[r][i] b_t[r][j]:
//     every digit pair is an int8 GEMM (v_mfma_i32_16x16x64_i8, exact in int32 over <= 2048 rows); the
//     pairs of one weight class s + t share an accumulator; the classes s + t >= 2 are kept (22 of the
//     25 pairs, 7 accumulators per tile);
//   * after every work item of 2048 rows the seven int32 classes of a tile are folded into the
//     workgroup's f64 partial (sum_c 256^c C_c, exact in f64 up to the final roundings).
// Error, relative to an entry's natural scale ||y_i|| ||y_j|| (y = sqrt(d) x; M_i = colmax_i sqrt(dmax)
// is what the fixed point is scaled to, R_i = M_i / ||y_i||):
//   * rounding to fixed point: |F - Y 2^k| <= 1/2 = 2^-39 M  ->  2^-39 R_i / sqrt 3; R <= 1 unless the
//     weights are tiny exactly where the column is large, typically 1e-2 .. 1e-3: ~1e-14;
//   * the three dropped pairs (0,0) (0,1) (1,0): on the diagonal b_0^2 does not average out -- a bias
//     of 2^12.4 per row against (2^38 / (M / rms))^2: 2^-63.6 (M_i / rms y_i)^2 = 2^-63.6 n R_i^2.
//     (The first cut of this kernel kept six digits and the classes s + t >= 5: 21 pairs, and a
//     diagonal bias of 2^-45.6 (M / rms)^2 -- 2.4e-12 measured on standard normal columns, and an
//     envelope of M / rms < ~30.  One digit less and the lowest classes kept costs one MFMA more per
//     tile and chunk and is 2^18 times better where it matters.)
// The envelope R_i^2 <= min(64, 2^27 / n) (both terms < 1e-11) is checked on the DEVICE after the
// product -- ||y_i||^2 is the diagonal just computed -- together with the screening of d (negative
// or non-finite weights): a call outside it is handed to the f64 kernel without a host
// synchronisation; both kernels are launched and the one the flag does not select returns at once.
// Non-finite X is excluded once per block on the host (dense_matrix.py).  Measured against the oracle
// in tests/test_gpu_syrk_i8.py.
//
// Layout (4 waves, ONE per SIMD: 256 VGPRs of digit fragments + 252 AGPRs of int32 accumulators):
// raw f64 rows come in by LDS-DMA in half chunks of 32 rows into a ring of THREE buffers (every half is
// requested one whole chunk iteration before it is converted); a chunk is 64 rows (K of the MFMA);
// lane (16 columns x 4 row quads per wave instruction) converts 4 rows of a column at a time,
// transposes the 4 x 5 digit bytes with v_perm_b32 and writes five 4-byte words into the digit planes
// [digit][column][64 rows]; the 25 fragments a wave needs (5 column blocks x 5 digits) are read into
// registers between two barriers, then the planes are rewritten for the next chunk while the 198
// MFMAs of the wave's 9 tiles run from registers.
#include <algorithm>
#include <cmath>

#include "common.hpp"

#if defined(I8_NO_SB)
#define I8_SB
#else
#define I8_SB __builtin_amdgcn_sched_barrier(0)
#endif
#if defined(I8_TRACE)   // ablation build: workgroup 0 stamps the phases of chunk iterations 8 .. 11 (scripts/dev/trace_i8.py)
#define I8_STAMP(k)                                                                                     \
    do {                                                                                                \
        if (blockIdx.x == 0 && lane == 0 && trace_it >= 8 && trace_it < 12)                             \
            trace_buf[((trace_it - 8) * 4 + wave) * 8 + (k)] = __builtin_readcyclecounter();            \
    } while (0)
#else
#define I8_STAMP(k)
#endif
namespace tmh {

constexpr int I8_W = 128;                       // padded columns
constexpr int I8_T = 36;                        // lower-triangular 16 x 16 tiles
constexpr int I8_ND = 5;                        // digits (40-bit fixed point)
constexpr int I8_WLO = 2;                       // lowest weight class s + t kept
constexpr int I8_NC = 2 * (I8_ND - 1) - I8_WLO + 1;   // classes kept: s + t = 2 .. 8 (7)
constexpr int I8_FBITS = 38;                    // |F| <= 2^38 < 0.498 * 2^40
constexpr int I8_RS = 64;                       // rows per chunk = K of the MFMA
constexpr int I8_HS = 32;                       // rows per half chunk (one DMA ring slot)
constexpr int I8_WAVES = 4;
constexpr int I8_THREADS = I8_WAVES * 64;
constexpr int I8_PSTR = 64;                     // bytes per column of a digit plane (64 rows, 16-byte groups swizzled)
constexpr int I8_PLANE = I8_W * I8_PSTR;
constexpr int I8_PLANES = I8_ND * I8_PLANE;     // 40 960 B
constexpr int I8_RAWSTR = 1024 + 32;            // bytes per raw f64 row in LDS
constexpr int I8_RAWBUF = I8_HS * I8_RAWSTR;    // one half chunk: 33 792 B
#ifndef I8_DMA_AUX
#define I8_DMA_AUX 0                 // cache policy bits of the LDS-DMA copies (1 / 2 / 3 = sc0 / nt / both: within the noise, -1 %)
#endif
#ifndef I8_CPI_N
#define I8_CPI_N 32
#endif
constexpr int I8_CPI = I8_CPI_N;                // chunks per work item (2048 rows: int32 stays exact)
constexpr int I8_ITEM_ROWS = I8_CPI * I8_RS;
constexpr size_t I8_LDS = (size_t)I8_PLANES + 3 * (size_t)I8_RAWBUF + 3 * I8_HS * sizeof(double) + 16;
static_assert(I8_LDS <= 160 * 1024, "planes + the ring of 3 half chunks fill the CU's LDS");

typedef int i8_v4 __attribute__((ext_vector_type(4)));

struct I8Info {                 // written by the prep kernels, read by the main / finish kernels
    unsigned dmax_bits_hi;      // max |d| as the high word of its double (monotone for d >= 0)
    unsigned dmax_bits_lo;
    unsigned flag;              // != 0: take the f64 kernel (negative / non-finite weight)
    unsigned pad;
};

// the digit pairs (s, t) with s + t >= I8_WLO, class by class (22 of the 25)
constexpr int i8_pair_find(int idx, bool want_s) {
    int k = 0;
    for (int w = I8_WLO; w <= 2 * (I8_ND - 1); ++w)
        for (int s = 0; s < I8_ND; ++s) {
            const int t = w - s;
            if (t < 0 || t >= I8_ND) continue;
            if (k == idx) return want_s ? s : t;
            ++k;
        }
    return -1;
}
constexpr int i8_pair_s(int idx) { return i8_pair_find(idx, true); }
constexpr int i8_pair_t(int idx) { return i8_pair_find(idx, false); }
constexpr int i8_count_pairs() {
    int k = 0;
    while (i8_pair_find(k, true) >= 0) ++k;
    return k;
}
constexpr int I8_NP = i8_count_pairs();
static_assert(I8_NP == 22, "the pair groups of the chunk loop are laid out for 22 pairs");
// (a class holds <= 5 pairs: 5 * 128 * 128 * 2048 rows < 2^31 -- the int32 sums are exact)

// The 36 tiles (bi, bj <= bi) of the 8 column blocks, 9 per wave, dealt so that a wave needs the digit
// fragments of only FIVE blocks (100 registers; all 8 blocks x 5 digits = 160 do not fit next to the
// 252 accumulators and the conversion -- and a single spill reload inside the chunk loop is a
// scratch load, whose s_waitcnt vmcnt(0) also waits for every LDS-DMA copy in flight):
//   wave 0, blocks {0,1,2,3,4}: (1,0) (2,0) (3,0) (4,0) (2,1) (3,1) (4,1) (3,2) (4,2)
//   wave 1, blocks {0,1,5,6,7}: (5,0) (6,0) (7,0) (5,1) (6,1) (7,1) (6,5) (0,0) (1,1)
//   wave 2, blocks {2,3,5,6,7}: (5,2) (6,2) (7,2) (5,3) (6,3) (7,3) (7,5) (2,2) (3,3)
//   wave 3, blocks {3,4,5,6,7}: (5,4) (6,4) (7,4) (4,3) (7,6) (4,4) (5,5) (6,6) (7,7)
template <int WID>
struct I8Tiles {
    static constexpr int NB = 5;
    static constexpr int block(int l) {
        constexpr int B[4][5] = {{0, 1, 2, 3, 4}, {0, 1, 5, 6, 7}, {2, 3, 5, 6, 7}, {3, 4, 5, 6, 7}};
        return B[WID][l];
    }
    static constexpr int local(int b) {
        for (int l = 0; l < NB; ++l)
            if (block(l) == b) return l;
        return 0;
    }
    static constexpr int bi(int q) {
        constexpr int T[4][9] = {{1, 2, 3, 4, 2, 3, 4, 3, 4}, {5, 6, 7, 5, 6, 7, 6, 0, 1},
                                 {5, 6, 7, 5, 6, 7, 7, 2, 3}, {5, 6, 7, 4, 7, 4, 5, 6, 7}};
        return T[WID][q];
    }
    static constexpr int bj(int q) {
        constexpr int T[4][9] = {{0, 0, 0, 0, 1, 1, 1, 2, 2}, {0, 0, 0, 1, 1, 1, 5, 0, 1},
                                 {2, 2, 2, 3, 3, 3, 5, 2, 3}, {4, 4, 4, 3, 6, 4, 5, 6, 7}};
        return T[WID][q];
    }
    static constexpr int tile(int q) { return bi(q) * (bi(q) + 1) / 2 + bj(q); }
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t i8_rsrc_t;
__device__ __forceinline__ i8_rsrc_t i8_rsrc(const void *base, int64_t bytes) {
    const unsigned nb = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)nb, 0x00020000);
}
__device__ __forceinline__ void i8_dma16(i8_rsrc_t rs, void *lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff, 0, 0, I8_DMA_AUX);
}
__device__ __forceinline__ void i8_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ unsigned i8_perm(unsigned hi, unsigned lo, unsigned sel) {
    return __builtin_amdgcn_perm(hi, lo, sel);
}
#else
struct i8_rsrc_t {};
inline i8_rsrc_t i8_rsrc(const void *, int64_t) { return {}; }
inline void i8_dma16(i8_rsrc_t, void *, int) {}
inline void i8_lds_barrier() {}
inline unsigned i8_perm(unsigned, unsigned, unsigned) { return 0; }
#endif

// ---- screening of the weights: max |d| and "any weight negative or non-finite"
__global__ __launch_bounds__(256) void i8_screen_kernel(const double *__restrict__ d, int64_t n, I8Info *info) {
    double mx = 0.0;
    bool bad = false;
    // (four independent loads per turn: one load per turn ran at 1.7 TB/s, 0.05 ms for the 80 MB of cfg4)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double v0 = d[i], v1 = d[i + stride], v2 = d[i + 2 * stride], v3 = d[i + 3 * stride];
        bad |= !(v0 >= 0.0) || !(v0 <= 1.7e308) || !(v1 >= 0.0) || !(v1 <= 1.7e308) ||
               !(v2 >= 0.0) || !(v2 <= 1.7e308) || !(v3 >= 0.0) || !(v3 <= 1.7e308);
        mx = fmax(fmax(mx, fmax(v0, v1)), fmax(v2, v3));
    }
    for (; i < n; i += stride) {
        const double v = d[i];
        bad |= !(v >= 0.0) || !(v <= 1.7e308);          // negative, NaN or inf
        mx = fmax(mx, v);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmax(mx, __shfl_down(mx, off, 64));
        bad |= (bool)__shfl_down((int)bad, off, 64);
    }
    // one pair of atomics per WORKGROUP (thousands of waves on the one address cost more than the scan)
    __shared__ double wmx[4];
    __shared__ int wbad[4];
    if ((threadIdx.x & 63) == 0) {
        wmx[threadIdx.x >> 6] = mx;
        wbad[threadIdx.x >> 6] = (int)bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(wmx[0], wmx[1]), fmax(wmx[2], wmx[3]));
        if (wbad[0] | wbad[1] | wbad[2] | wbad[3]) atomicOr(&info->flag, 1u);
        // max of non-negative doubles = max of their bit patterns
        atomicMax(reinterpret_cast<unsigned long long *>(info), (unsigned long long)__double_as_longlong(mx > 0.0 ? mx : 0.0));
    }
}

// sigma[i] = 2^k_i (the fixed-point scale of column i), rscale[i] = 2^-k_i
// history (may be NULL; per matrix, device memory, int32[tm_dense_sandwich_i8_history_words()] zeroed once):
// [0] = consecutive calls that left the envelope after the product, [1] = calls, [4 ..] = 128 doubles: the
// diagonal of the previous call's result.  After three misses in a row the int8 kernel is skipped (flag bit 2: the f64
// kernel alone runs, instead of both) and tried again every 32nd call.
__global__ void i8_scale_kernel(const double *__restrict__ colmax, int m, int64_t n_rows, I8Info *info,
                                double *__restrict__ sigma, double *__restrict__ rscale, int *history) {
    const int i = threadIdx.x;
    if (history != nullptr && i == 0) {
        const int calls = history[1]++;
        if (history[0] >= 3 && (calls & 31) != 0) atomicOr(&info->flag, 4u);
    }
    if (i >= I8_W) return;
    const double dmax = __longlong_as_double(*reinterpret_cast<const long long *>(info));
    // The envelope test of the column against the diagonal of the PREVIOUS call (history + 4: 128 doubles, 0 =
    // none yet): the weights of an IRLS solver move slowly, so a call that is going to miss is recognised
    // BEFORE the product and costs the f64 kernel alone, not the int8 attempt as well.
    // A wrong guess costs one call: every call records its diagonal afterwards (i8_record_diag_kernel).
    if (history != nullptr && i < m && n_rows > 0) {
        const double prev = reinterpret_cast<const double *>(history + 4)[i];
        const double bound = fmin(64.0, 134217728.0 / (double)n_rows);
        if (prev > 0.0 && !(colmax[i] * colmax[i] * dmax <= bound * prev)) atomicOr(&info->flag, 4u);
    }
    __syncthreads();
    double s = 1.0, r = 1.0;
    if (i < m) {
        const double big = colmax[i] * sqrt(dmax);
        if (big > 0.0 && info->flag == 0) {
            int e;
            frexp(big, &e);                      // big = f * 2^e, f in [0.5, 1): big <= 2^e
            s = ldexp(1.0, I8_FBITS - e);
            r = ldexp(1.0, e - I8_FBITS);
        }
    }
    sigma[i] = s;
    rscale[i] = r;
}

// ---- the main kernel
// CSUM: X' d (length m) comes out of the same pass -- the conversion forms x sqrt(d) for every entry anyway,
// one more v_fma_f64 per entry accumulates (x sqrt d) sqrt d per lane in f64 (exact arithmetic on the raw
// values, independent of the fixed-point envelope); StandardizedMatrix.sandwich then needs no second pass
// (reference: standardized_mat.py:149-150 calls transpose_matvec).
// CEN: the columns are CENTRED on the way in, y = sqrt(d) (x - center): the product is (X - 1 c')' diag(d) (X - 1 c')
// and the column sums are (X - 1 c')' d.  StandardizedMatrix.sandwich (standardized_mat.py:123-172) subtracts
// mean-sized rank-one terms from the raw product; slicing the centred columns instead keeps the fixed point
// (and the f64 sums) at the scale of the RESULT -- an uncentred "year" column (2000 +- 5) would otherwise lose
// (mean / std)^2 = 1.6e5 of the 2e-14.  One more v_add_f64 per entry in the conversion's shadow.
template <bool CSUM, bool CEN>
__global__ __launch_bounds__(I8_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
void syrk_i8_kernel(const double *__restrict__ X, int64_t ldx, int64_t n, int64_t m, const double *__restrict__ d,
                    const double *__restrict__ sigma, const I8Info *__restrict__ info, int n_items,
                    unsigned *__restrict__ counter, double *__restrict__ part, double *__restrict__ colsum,
                    const double *__restrict__ center) {
    if (info->flag != 0) return;                                          // the f64 kernel takes this call
    // SEPARATE static LDS objects: the compiler tracks LDS-DMA copies per LDS variable (alias scopes of
    // the module-LDS lowering) and makes every LDS access that may alias a copy in flight wait for it
    // (s_waitcnt vmcnt(0)) -- carved out of one dynamic buffer, the fragment reads waited for the copies
    // issued a moment earlier and nothing was asynchronous.  The raw ring is read back with inline-asm
    // ds_read (invisible to that tracking: the slot being read is never the one being filled; the
    // hand-over is the explicit vmcnt wait + barrier below).
    __shared__ __attribute__((aligned(16))) unsigned char planes[I8_PLANES];      // [6][128][I8_PSTR]: ONE chunk
    __shared__ __attribute__((aligned(16))) unsigned char raw[3 * I8_RAWBUF];     // ring of 3 half chunks: [32 rows][I8_RAWSTR] f64
    __shared__ double dl[3 * I8_HS];                                              // sqrt(d) of the ring slots
    __shared__ unsigned slot_mem[4];
    unsigned *slot = slot_mem;
#if defined(I8_TRACE)
    unsigned long long *trace_buf = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(counter) - 256 + 2560);
    int trace_it = 0;
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // conversion role: wave w owns columns 32 w .. 32 w + 31; lane = (column in block of 16, row quad)
    const int cl = lane & 15, rq = lane >> 4;

    // ---- the workgroup's stream of HALF chunks: items of I8_CPI chunks from the atomic counter
    unsigned idL = blockIdx.x;            // item of the next half to request
    int hL = 0;                           // its half offset inside the item (0 .. 2 CPI - 1)
    unsigned idNext;
    if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
    __syncthreads();
    idNext = __builtin_amdgcn_readfirstlane(*slot);
    bool pending = false;

    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    double dreg = 0.0;
    const int dma_voff = lane * 2 < m ? lane * 16 : 0x7ffffff0;           // column pairs beyond the block read as 0
    // request the next half chunk into ring slot rb: wave w copies rows 8 w .. 8 w + 7
    auto issue_half = [&](int rb) -> unsigned {
        const unsigned id = idL;
        const int64_t tb = (int64_t)id * I8_ITEM_ROWS + (int64_t)hL * I8_HS;
        // (ldx = row stride of X in elements: a 128-column panel of a wider block is read in place)
        const int64_t left = min((n - tb) * ldx * 8, (int64_t)I8_HS * ldx * 8);
        // (the 64-bit products above are computed on the vector unit: without the readfirstlane the
        // descriptor counts as divergent and every copy sits in a waterfall loop)
        const uint64_t xb = (uint64_t)(uintptr_t)(X + tb * ldx);
        const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xb);
        const unsigned xhi = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
        const unsigned nbytes = __builtin_amdgcn_readfirstlane(
            left <= 0 ? 0u : (left > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)left));
        const i8_rsrc_t rs = i8_rsrc(reinterpret_cast<const void *>((uintptr_t)(((uint64_t)xhi << 32) | xlo)), nbytes);
        // (d first: the wave that loads it waits for it with the younger copies still in flight)
        if (wave == 0 && lane < I8_HS) dreg = tb + lane < n ? d[tb + lane] : 0.0;
#if !defined(I8_ABLATE_NO_DMA)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            i8_dma16(rs, raw + rb * I8_RAWBUF + (8 * wave + j) * I8_RAWSTR, (int)((8 * wave + j) * ldx * 8) + dma_voff);
#endif
        if (++hL == 2 * I8_CPI) {
            hL = 0;
            idL = idNext;
            if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
            pending = true;
        }
        return id;
    };
    auto publish_d = [&](int rb) {          // sqrt(d) of the half just requested (wave 0, after its load landed)
        if (wave == 0 && lane < I8_HS) dl[rb * I8_HS + lane] = sqrt(dreg);
    };

    // this lane's two columns and their scales
    const double sg0 = sigma[32 * wave + cl], sg1 = sigma[32 * wave + 16 + cl];
    double cen0 = 0.0, cen1 = 0.0;
    if constexpr (CEN) {
        cen0 = 32 * wave + cl < m ? center[32 * wave + cl] : 0.0;
        cen1 = 32 * wave + 16 + cl < m ? center[32 * wave + 16 + cl] : 0.0;
    }
    double cs0 = 0.0, cs1 = 0.0;                                           // CSUM: this lane's share of X' d, columns cb = 0 / 1
    const double MAGIC = 6755399441055744.0 + 551911719040.0;             // 1.5 * 2^52 + 0x8080808080
    // one half chunk (ring slot rb) -> rows 32 hh .. 32 hh + 31 of the digit planes.  Per call: the
    // lane's 4 rows (quad qq of 8) of column block cb (0 / 1) -- 16 calls cover the half.
    const unsigned raw_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)raw;
    const unsigned planes_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)planes;
    struct Quad { double x[4]; };
    // the lane's 4 raw rows (quad qq of 8) of column block cb from ring slot rb: issued here, complete
    // after quad_wait (inline asm: see the note on the LDS objects above)
    auto quad_issue = [&](int rb, auto cb_c, auto qq_c) -> Quad {
        constexpr int cb = decltype(cb_c)::value, qq = decltype(qq_c)::value;   // qq: 0 / 1 (quads rq and rq + 4)
        Quad q;
#if defined(I8_ABLATE_NO_CONVERT) || defined(I8_ABLATE_DMA_ONLY)
        q.x[0] = q.x[1] = q.x[2] = q.x[3] = 0.0;
        return q;
#endif
        const int col = 32 * wave + 16 * cb + cl;
        const int row0 = 4 * (rq + 4 * qq);                                     // row inside the half
        const unsigned ra = raw_lds + (unsigned)(rb * I8_RAWBUF + row0 * I8_RAWSTR + col * 8);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1056\n\tds_read_b64 %2, %4 offset:2112\n\t"
                     "ds_read_b64 %3, %4 offset:3168"
                     : "=&v"(q.x[0]), "=&v"(q.x[1]), "=&v"(q.x[2]), "=&v"(q.x[3])
                     : "v"(ra)
                     : "memory");
#else
        (void)ra;
        q.x[0] = q.x[1] = q.x[2] = q.x[3] = 0.0;
#endif
        static_assert(I8_RAWSTR == 1056, "the ds_read offsets above are multiples of the raw row stride");
        return q;
    };
    auto quad_wait = [&](Quad &q) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(I8_ABLATE_NO_CONVERT) && !defined(I8_ABLATE_DMA_ONLY)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q.x[0]), "+v"(q.x[1]), "+v"(q.x[2]), "+v"(q.x[3])::"memory");
#endif
    };
    // 4 rows of one column -> fixed point -> 5 digit words of the planes (rows 32 hh + row0 ..), cut into
    // I8_CSTEPS steps of two or three instructions: the chunk loop issues one step behind each MFMA
    // (fenced with sched_barrier -- left to the scheduler, the conversion ended up in one lump behind
    // the MFMAs, or the MFMAs of one accumulator back to back).
    struct Conv {
        double dv[4];
        unsigned lo[4], hi[4];
        unsigned t[4], h[2], g[5];
    };
    auto conv_step = [&](auto k_c, const Quad &q, Conv &c, int rb, int hh, auto cb_c, auto qq_c) {
        constexpr int k = decltype(k_c)::value;
        constexpr int cb = decltype(cb_c)::value, qq = decltype(qq_c)::value;
#if defined(I8_ABLATE_NO_CONVERT) || defined(I8_ABLATE_DMA_ONLY)
        return;
#endif
        const int row0 = 4 * (rq + 4 * qq);
        if constexpr (k == 0) {                     // sqrt(d) of the 4 rows (LDS; used from step 1 + I8_CLAT on)
#pragma unroll
            for (int j = 0; j < 4; ++j) c.dv[j] = dl[rb * I8_HS + row0 + j];
        } else if constexpr (k >= 1 && k <= 4) {
            const double sg = cb ? sg1 : sg0;
            double xv = q.x[k - 1];
            if constexpr (CEN) xv -= cb ? cen1 : cen0;
            const double u = xv * c.dv[k - 1];
            const double t = __builtin_fma(u, sg, MAGIC);
            if constexpr (CSUM) {
                if constexpr (cb) cs1 = __builtin_fma(u, c.dv[k - 1], cs1);
                else cs0 = __builtin_fma(u, c.dv[k - 1], cs0);
            }
            c.lo[k - 1] = (unsigned)__double2loint(t);
            c.hi[k - 1] = (unsigned)__double2hiint(t);
        } else if constexpr (k == 5) {              // 4 rows x 4 digit bytes -> 4 digits x 4 row bytes
            c.t[0] = i8_perm(c.lo[1], c.lo[0], 0x05010400u);     // r0.0 r1.0 r0.1 r1.1
            c.t[1] = i8_perm(c.lo[1], c.lo[0], 0x07030602u);     // r0.2 r1.2 r0.3 r1.3
        } else if constexpr (k == 6) {
            c.t[2] = i8_perm(c.lo[3], c.lo[2], 0x05010400u);
            c.t[3] = i8_perm(c.lo[3], c.lo[2], 0x07030602u);
        } else if constexpr (k == 7) {              // g ^ 0x80: biased byte -> two's complement digit
            c.g[0] = i8_perm(c.t[2], c.t[0], 0x05040100u) ^ 0x80808080u;   // r0.0 r1.0 r2.0 r3.0
        } else if constexpr (k == 8) {
            c.g[1] = i8_perm(c.t[2], c.t[0], 0x07060302u) ^ 0x80808080u;
        } else if constexpr (k == 9) {
            c.g[2] = i8_perm(c.t[3], c.t[1], 0x05040100u) ^ 0x80808080u;
        } else if constexpr (k == 10) {
            c.g[3] = i8_perm(c.t[3], c.t[1], 0x07060302u) ^ 0x80808080u;
        } else if constexpr (k == 11) {
            c.h[0] = i8_perm(c.hi[1], c.hi[0], 0x05010400u);               // r0.4 r1.4 . .
            c.h[1] = i8_perm(c.hi[3], c.hi[2], 0x05010400u);
        } else if constexpr (k == 12) {
            c.g[4] = i8_perm(c.h[1], c.h[0], 0x05040100u) ^ 0x80808080u;
        } else if constexpr (k == 13) {
            // rows 32 hh + row0 ..: 16-byte group 2 hh + qq, swizzled by the column (see frag below).
            // Written with inline asm: the compiler pairs these stores into ds_write2st64_b32, whose merged
            // memory operand loses the LDS variable -- and an LDS access the wait-count pass cannot
            // attribute waits for EVERY LDS-DMA copy in flight (vmcnt(0) right behind the request of a half).
            const int col = 32 * wave + 16 * cb + cl;
            const unsigned pa =
                planes_lds + (unsigned)(col * I8_PSTR + (((2 * hh + qq) ^ ((cl >> 1) & 3)) << 4) + 4 * rq);
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:8192\n\tds_write_b32 %0, %3 offset:16384\n\t"
                         "ds_write_b32 %0, %4 offset:24576\n\tds_write_b32 %0, %5 offset:32768"
                         :
                         : "v"(pa), "v"(c.g[0]), "v"(c.g[1]), "v"(c.g[2]), "v"(c.g[3]), "v"(c.g[4])
                         : "memory");
#else
            (void)pa;
#endif
            static_assert(I8_PLANE == 8192 && I8_ND == 5, "the ds_write offsets above are the plane strides");
        }
    };
    constexpr int I8_CSTEPS = 14;
    constexpr int I8_CLAT = 3;                      // MFMAs between the sqrt(d) read and its first use
    auto quad_convert = [&](const Quad &q, int rb, int hh, auto cb_c, auto qq_c) {
        Conv c;
        static_for<I8_CSTEPS>([&](auto kc) { conv_step(kc, q, c, rb, hh, cb_c, qq_c); });
    };
    auto convert_quad = [&](int rb, int hh, auto cb_c, auto qq_c) {
        Quad q = quad_issue(rb, cb_c, qq_c);
        quad_wait(q);
        quad_convert(q, rb, hh, cb_c, qq_c);
    };
    auto convert_half = [&](int rb, int hh) {
        convert_quad(rb, hh, C0{}, C0{});
        convert_quad(rb, hh, C0{}, C1{});
        convert_quad(rb, hh, C1{}, C0{});
        convert_quad(rb, hh, C1{}, C1{});
    };

    // fragment of column block b, digit s: lane (i = lane & 15, kg = lane >> 4) -> rows 16 kg .. 16 kg + 15.
    // A column is 64 bytes, its four 16-byte row groups stored at (kg ^ (column >> 1)) & 3: both the
    // ds_read_b128 of the fragments and the ds_write_b32 of the conversion are then free of bank conflicts
    // (with a padded stride of 80 bytes every fragment read took 8 LDS cycles instead of 4).
    const int foff = (lane & 15) * I8_PSTR + (((lane >> 4) ^ ((lane >> 1) & 3)) << 4);
    auto frag = [&](int b, int s) {
        return *reinterpret_cast<const i8_v4 *>(planes + s * I8_PLANE + b * 16 * I8_PSTR + foff);
    };

    auto run = [&](auto wid) {
        constexpr int WID = decltype(wid)::value;
        i8_v4 acc[9][I8_NC];
        // the workgroup's f64 partial [tile][16][16]: this wave's 9 tiles, folded after every item
        double *dst = part + (int64_t)blockIdx.x * (I8_T * 256);
        auto flush = [&]() {
            // (the MFMAs are inline asm: the hazard recognizer does not see them -- their results are read
            // below with v_accvgpr_read, 18 wait states behind the last one at the most)
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
            // (the partial's 36 addresses must not be hoisted out of the chunk loop: 72 registers)
            unsigned lane_off = (unsigned)(((4 * (lane >> 4)) * 16 + (lane & 15)) * 8);
            asm volatile("" : "+v"(lane_off));
            char *const fb = reinterpret_cast<char *>(dst) + lane_off;
            static_for<9>([&](auto sc) {
                constexpr int s9 = decltype(sc)::value;
                constexpr int t = I8Tiles<WID>::tile(s9);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // sum_c 256^c C_c over the kept classes (Horner from the top; 256^I8_WLO is applied at the end)
                    double v = (double)acc[s9][I8_NC - 1][r];
#pragma unroll
                    for (int c = I8_NC - 2; c >= 0; --c) v = v * 256.0 + (double)acc[s9][c][r];
                    // (no-return atomic: 36 read-modify-writes in flight instead of 36 round trips; this wave
                    // is the only writer of the address, so the sum order stays the item order)
                    atomicAdd(reinterpret_cast<double *>(fb + (t * 256 + r * 16) * 8), v);
                }
            });
        };

        // ---- prologue: halves 0, 1 -> planes (chunk 0); halves 2, 3 requested.  Half k lives in ring slot
        // k % 3; a half is requested one whole iteration before it is converted.
        unsigned id_c = issue_half(0);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // d of half 0 (older than the 8 copies)
        publish_d(0);
        (void)issue_half(1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        publish_d(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        i8_lds_barrier();
        if (pending) { idNext = __builtin_amdgcn_readfirstlane(*slot); pending = false; }
        convert_half(0, 0);
        convert_half(1, 1);
        i8_lds_barrier();                                            // slots 0, 1 free, planes = chunk 0
        unsigned id_c1 = issue_half(2);                              // half 2 (first half of chunk 1)
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        publish_d(2);
        (void)issue_half(0);                                         // half 3
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        publish_d(0);
        int sa = 2;                                                  // ring slot of the first half of chunk c + 1

        // Iteration c: planes hold chunk c; the halves of chunk c + 1 sit in slots sa, sa + 1 (landed or
        // landing); the halves of chunk c + 2 are requested behind barriers B and C into the two slots
        // that are free by then.
        // (an item = I8_CPI whole chunks, rows beyond n read as 0: the inner trip count is fixed and the
        // fold sits at ONE place behind it -- a conditional fold inside the chunk loop made the allocator
        // keep every accumulator twice across the join)
        while (id_c < (unsigned)n_items) {
          // (zeroed HERE, not in the fold: the accumulators are then not live across the item loop)
#pragma unroll
          for (int t = 0; t < 9; ++t)
#pragma unroll
              for (int c = 0; c < I8_NC; ++c) {
                  acc[t][c] = i8_v4{0, 0, 0, 0};
                  // pinned to the accumulation registers: left to itself the allocator carried part of
                  // the accumulators through the chunk loop in VGPRs and spilled ~340 registers
                  asm volatile("" : "+a"(acc[t][c]));
              }
          for (int oc = 0; oc < I8_CPI; ++oc) {
            I8_STAMP(0);
            const int sb = sa == 2 ? 0 : sa + 1;     // slot of the second half of chunk c + 1
            const int sc = sb == 2 ? 0 : sb + 1;     // the third slot: second half of chunk c, consumed
            i8_lds_barrier();                        // A: planes complete, slot sc consumed, dl published; this
                                                     // wave waited for its rows of slot sa before it
            I8_STAMP(1);
            if (pending) { idNext = __builtin_amdgcn_readfirstlane(*slot); pending = false; }
            Quad qa = quad_issue(sa, C0{}, C0{});    // (slot sa: every wave's rows landed before A)
            // The 198 MFMAs (22 digit pairs x the wave's 9 tiles, pair-major).  The first 27 (class 2: the
            // digits 0, 2, then 1) start while the fragments of the digits 3 and 4 are still being read;
            // behind barrier B every group runs interleaved with the conversion of one quad of the NEXT chunk
            // (its raw rows were read into registers during the group before): one MFMA keeps the matrix
            // pipe for 16 cycles, two conversion instructions issue in its shadow.
            using TL = I8Tiles<WID>;
            i8_v4 F[TL::NB][I8_ND];
            auto mf = [&](auto lo_c, auto hi_c) {
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
                static_for<HI - LO>([&](auto ic) {
                    constexpr int k = LO + decltype(ic)::value;
                    constexpr int idx = k / 9, q = k % 9;
                    constexpr int ps = i8_pair_s(idx), pt = i8_pair_t(idx);
#if defined(I8_ABLATE_DMA_ONLY)
                    return;
#endif
                    // (inline asm with the accumulator tied in place: through the builtin the allocator gave
                    // many MFMAs a destination other than their accumulator and, with 252 of the 256
                    // accumulation registers taken, parked accumulators in VGPRs around them -- ~1000
                    // v_accvgpr_read / _write and the stalls for the MFMA results they copy)
                    i8_v4 &cacc = acc[q][ps + pt - I8_WLO];
                    const i8_v4 fa = F[TL::local(TL::bi(q))][ps], fb = F[TL::local(TL::bj(q))][pt];
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(cacc) : "v"(fa), "v"(fb));
                });
            };
#define I8_MF(lo, hi) mf(std::integral_constant<int, lo>{}, std::integral_constant<int, hi>{})
            // the digit fragments of the wave's five blocks (25 of the chunk's 40), in the order of first use
            static_for<I8_ND>([&](auto oc_) {
                constexpr int order[I8_ND] = {0, 2, 1, 3, 4};
                constexpr int sd = order[decltype(oc_)::value];
#pragma unroll
                for (int b = 0; b < TL::NB; ++b) {
#if defined(I8_ABLATE_DMA_ONLY)
                    F[b][sd] = i8_v4{0, 0, 0, 0};
#else
                    F[b][sd] = frag(TL::block(b), sd);
#endif
                }
            });
            static_assert(i8_pair_s(0) == 0 && i8_pair_t(0) == 2 && i8_pair_s(2) == 2 && i8_pair_s(3) + i8_pair_t(3) == 3,
                          "the first three pairs are class 2");
            I8_MF(0, 27);
#if !defined(I8_NO_WEAVE)
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);          // digits 0, 2
            static_for<15>([&](auto) {                                   // digits 1, 3, 4 behind the first MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            });
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
#endif
            I8_SB;
            I8_STAMP(2);
            // B: every wave holds its fragments: the planes may be rewritten
            i8_lds_barrier();
            I8_STAMP(3);
            quad_wait(qa);
            const unsigned id_a = issue_half(sc);    // first half of chunk c + 2 -> the free slot
            // one group: MFMAs [lo, hi) of the list, the conversion steps of quad q behind them one by one
            auto group = [&](auto lo_c, auto hi_c, const Quad &q, int rb, int hh, auto cb_c, auto qq_c) {
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
                static_assert(HI - LO >= I8_CSTEPS + I8_CLAT, "a group holds all steps of one quad");
                Conv c;
                static_for<HI - LO>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    mf(std::integral_constant<int, LO + i>{}, std::integral_constant<int, LO + i + 1>{});
                    if constexpr (i == 0) conv_step(std::integral_constant<int, 0>{}, q, c, rb, hh, cb_c, qq_c);
                    if constexpr (i > I8_CLAT && i - I8_CLAT < I8_CSTEPS)
                        conv_step(std::integral_constant<int, i - I8_CLAT>{}, q, c, rb, hh, cb_c, qq_c);
                    I8_SB;
                });
            };
#define I8_GROUP(lo, hi, q, rb, hh, cb, qq) \
    group(std::integral_constant<int, lo>{}, std::integral_constant<int, hi>{}, q, rb, hh, cb, qq)
            Quad qb = quad_issue(sa, C0{}, C1{});
            I8_GROUP(27, 49, qa, sa, 0, C0{}, C0{});
            quad_wait(qb);
            I8_SB;
            qa = quad_issue(sa, C1{}, C0{});
            I8_GROUP(49, 71, qb, sa, 0, C0{}, C1{});
            quad_wait(qa);
            I8_SB;
            qb = quad_issue(sa, C1{}, C1{});
            I8_GROUP(71, 93, qa, sa, 0, C1{}, C0{});
            quad_wait(qb);
            I8_SB;
            I8_GROUP(93, 114, qb, sa, 0, C1{}, C1{});
            I8_SB;
            // sqrt(d) of the half just requested into slot sc (its old values were last read before barrier A)
            I8_STAMP(4);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            publish_d(sc);
            I8_STAMP(5);
            // C: slot sa is consumed by everybody; this wave's rows of slot sb have landed (the 8 copies
            // into slot sc stay in flight)
            i8_lds_barrier();
            I8_STAMP(6);
            (void)issue_half(sa);                    // second half of chunk c + 2 -> the slot just consumed
            qa = quad_issue(sb, C0{}, C0{});
            I8_MF(114, 120);                         // (covers the latency of the first raw read behind C)
            quad_wait(qa);
            I8_SB;
            qb = quad_issue(sb, C0{}, C1{});
            I8_GROUP(120, 140, qa, sb, 1, C0{}, C0{});
            quad_wait(qb);
            I8_SB;
            qa = quad_issue(sb, C1{}, C0{});
            I8_GROUP(140, 160, qb, sb, 1, C0{}, C1{});
            quad_wait(qa);
            I8_SB;
            qb = quad_issue(sb, C1{}, C1{});
            I8_GROUP(160, 180, qa, sb, 1, C1{}, C0{});
            quad_wait(qb);
            I8_SB;
            I8_GROUP(180, 198, qb, sb, 1, C1{}, C1{});
            I8_SB;
#undef I8_MF
#undef I8_GROUP
            static_assert(I8_NP * 9 == 198, "the group bounds above");
            // sqrt(d) of the half just requested into slot sa (its old values were read before barrier C);
            // the wait also covers this wave's rows of the next chunk's first half (slot sc)
            I8_STAMP(7);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            publish_d(sa);
#if defined(I8_TRACE)
            ++trace_it;
#endif
            sa = sc;                                 // chunk c + 2 starts in the slot requested at B
            id_c = id_c1;
            id_c1 = id_a;
          }
          flush();                                    // the item is complete: fold the int32 classes
        }
    };
    if (wave == 0) run(std::integral_constant<int, 0>{});
    else if (wave == 1) run(std::integral_constant<int, 1>{});
    else if (wave == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 3>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (copies of halves beyond the end are still in flight)
    if constexpr (CSUM) {
        // lane (cl, rq) holds the rows 4 (rq + 4 qq) .. of its two columns: sum over the 4 row quads
        cs0 += __shfl_xor(cs0, 16, 64);
        cs0 += __shfl_xor(cs0, 32, 64);
        cs1 += __shfl_xor(cs1, 16, 64);
        cs1 += __shfl_xor(cs1, 32, 64);
        if (lane < 16) {
            if (32 * wave + cl < m) atomicAdd(colsum + 32 * wave + cl, cs0);
            if (32 * wave + 16 + cl < m) atomicAdd(colsum + 32 * wave + 16 + cl, cs1);
        }
    }
}

// out[i][j] = 2^-(k_i + k_j) * sum over the workgroups' partials, mirrored; a quarter tile per block
__global__ __launch_bounds__(1024) void syrk_i8_finish_kernel(const double *__restrict__ part, int nblk,
                                                              int n_cols, const double *__restrict__ rscale,
                                                              const I8Info *__restrict__ info,
                                                              double *__restrict__ out, int64_t ldo) {
    if (info->flag != 0) return;
#if defined(I8_TRACE)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.y == 0) {
        const unsigned long long *tb = reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(info) + 2560);
        for (int k = threadIdx.x; k < 128; k += 64) out[k] = (double)(tb[k] - tb[0]);
    }
    return;
#endif
    __shared__ double red[16][64];
    const int e = blockIdx.y * 64 + threadIdx.x, s = threadIdx.y, t = blockIdx.x;
    double a = 0.0;
    for (int b = s; b < nblk; b += 16) a += part[((int64_t)b * I8_T + t) * 256 + e];
    red[s][threadIdx.x] = a;
    __syncthreads();
    if (s == 0) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < 16; w += 4)
            v += (red[w][threadIdx.x] + red[w + 1][threadIdx.x]) + (red[w + 2][threadIdx.x] + red[w + 3][threadIdx.x]);
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        const int bj = t - bi * (bi + 1) / 2;
        const int ci = 16 * bi + (e >> 4), cj = 16 * bj + (e & 15);
        if (ci < n_cols && cj < n_cols && (bi != bj || (e >> 4) >= (e & 15))) {
            // 256^I8_WLO of the lowest class kept, 2^-k of both columns (all powers of two: exact)
            v = v * 65536.0 * rscale[ci] * rscale[cj];
            static_assert(I8_WLO == 2, "65536 = 256^I8_WLO");
            out[(int64_t)ci * ldo + cj] = v;
            if (ci != cj) out[(int64_t)cj * ldo + ci] = v;
        }
    }
}

// The a-posteriori envelope check (header: R_j^2 = colmax_j^2 dmax / out[j][j] <= min(64, 2^27 / n)); a
// column that fails it hands the call to the f64 kernel.  (A column whose weighted norm is 0 has only
// zero digits: exact, passes as 0 <= 0.)
__global__ void i8_envelope_kernel(const double *__restrict__ out, int64_t ldo, const double *__restrict__ colmax,
                                   int m, int64_t n, I8Info *info, int *history) {
    const int j = threadIdx.x;
    __shared__ int miss;
    if (j == 0) miss = 0;
    __syncthreads();
    const bool live = info->flag == 0;
#if defined(I8_TRACE)
    return;
#endif
    if (j < m && live) {
        const double dmax = __longlong_as_double(*reinterpret_cast<const long long *>(info));
        const double big2 = colmax[j] * colmax[j] * dmax;
        const double bound = fmin(64.0, 134217728.0 / (double)n);
        if (!(big2 <= bound * out[(int64_t)j * ldo + j])) {
            atomicOr(&info->flag, 2u);
            miss = 1;
        }
    }
    __syncthreads();
    // (only calls in which the int8 kernel ran count: a negative weight or a skipped call leaves the history)
    if (history != nullptr && j == 0 && live) history[0] = miss ? history[0] + 1 : 0;
}

// the diagonal of this call's result (whichever kernel produced it) for the next call's prediction
__global__ void i8_record_diag_kernel(const double *__restrict__ out, int64_t ldo, int m, int *history) {
    const int j = threadIdx.x;
    if (j < I8_W) reinterpret_cast<double *>(history + 4)[j] = j < m ? out[(int64_t)j * ldo + j] : 0.0;
}

// the f64 kernel's side of the hand-over: run_syrk_co_if(flag != 0)
int run_syrk_co_flagged(const double *X, int64_t ldx, int64_t n, int64_t m, const double *d, double *out,
                        int64_t ldo, double *colsum, const unsigned *flag, void *ws, hipStream_t st,
                        const double *center);
size_t syrk_co_ws_bytes();

// X: first element of the block (or of a 128-column panel of a wider one), ldx / ldo: row strides of X / out
// center (may be NULL): per-column centres c -- the product of X - 1 c' (colmax is then max |x - c| per column)
int run_syrk_i8_panel(const double *X, int64_t ldx, int64_t n, int64_t m, const double *d, const double *colmax,
                      double *out, int64_t ldo, double *colsum, int *history, hipStream_t st, const double *center) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    // (any row stride / width parity: with an odd stride every other row starts at an 8-byte aligned address, which
    // the 16-byte LDS-DMA copies take -- measured 10M x 127: 2.31 ms against 6.36 ms on the element-load f64 syrk.
    // The pair of columns that straddles the end of an odd-width row brings the next row's first entry into
    // the padded column m: its digits land in tiles the finish kernel drops; behind the last row the buffer
    // descriptor returns 0.)
    TM_REQUIRE(ldx >= m && ldo >= m, "row strides");
    TM_REQUIRE(m == 0 || syrk_co_ok(X, m), "the int8 syrk takes a 16-byte aligned C-ordered f64 block of <= 128 columns");
    if (m == 0) return TM_OK;
    if (n == 0) {
        TM_HIP(hipMemset2DAsync(out, sizeof(double) * (size_t)ldo, 0, sizeof(double) * (size_t)m, (size_t)m, st));
        if (colsum) TM_HIP(hipMemsetAsync(colsum, 0, sizeof(double) * (size_t)m, st));
        return TM_OK;
    }
    if (colsum) TM_HIP(hipMemsetAsync(colsum, 0, sizeof(double) * (size_t)m, st));
    const int64_t n_items64 = ceil_div(n, I8_ITEM_ROWS);
    TM_REQUIRE(n_items64 < (1ll << 31), "too many rows");
    const int n_items = (int)n_items64;
    const int grid = (int)std::min<int64_t>(n_items, tune("i8_grid", NUM_CU));
    const size_t part_bytes = sizeof(double) * (size_t)grid * I8_T * 256;
    // one workspace for both kernels: [this kernel's region | the f64 kernel's region]
    const size_t own_bytes = (4096 + part_bytes + 255) / 256 * 256;
    void *wsv = nullptr;
    int rc = get_workspace(own_bytes + syrk_co_ws_bytes(), &wsv, st);
    if (rc) return rc;
    char *wb = reinterpret_cast<char *>(wsv);
    I8Info *info = reinterpret_cast<I8Info *>(wb);                      // 16 B
    unsigned *counter = reinterpret_cast<unsigned *>(wb + 256);
    double *sigma = reinterpret_cast<double *>(wb + 512);               // 128 doubles
    double *rscale = reinterpret_cast<double *>(wb + 512 + 1024);       // 128 doubles
    double *part = reinterpret_cast<double *>(wb + 4096);
    TM_HIP(hipMemsetAsync(wb, 0, 4096 + part_bytes, st));
    const int sgrid = (int)std::min<int64_t>(4 * NUM_CU, ceil_div(n, 1024));
    hipLaunchKernelGGL(i8_screen_kernel, dim3((unsigned)sgrid), dim3(256), 0, st, d, n, info);
    hipLaunchKernelGGL(i8_scale_kernel, dim3(1), dim3(I8_W), 0, st, colmax, (int)m, n, info, sigma, rscale, history);
    TM_LAUNCH_CHECK();
    prof_begin(st);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(I8_THREADS), 0, st, X, ldx, n, m, d, sigma, info, n_items,
                           counter, part, colsum, center);
    };
    if (center) {
        if (colsum) go(syrk_i8_kernel<true, true>);
        else go(syrk_i8_kernel<false, true>);
    } else {
        if (colsum) go(syrk_i8_kernel<true, false>);
        else go(syrk_i8_kernel<false, false>);
    }
    prof_end(st);
    TM_LAUNCH_CHECK();
    hipLaunchKernelGGL(syrk_i8_finish_kernel, dim3(I8_T, 4), dim3(64, 16), 0, st, part, grid, (int)m, rscale, info,
                       out, ldo);
    hipLaunchKernelGGL(i8_envelope_kernel, dim3(1), dim3(I8_W), 0, st, out, ldo, colmax, (int)m, n, info, history);
    TM_LAUNCH_CHECK();
    // weights outside the envelope: the f64 kernel (its launches return at once when the flag is clear)
    prof_hold(true);               // (the event pair stays on the int8 kernel)
    rc = run_syrk_co_flagged(X, ldx, n, m, d, out, ldo, colsum, &info->flag, wb + own_bytes, st, center);
    prof_hold(false);
    if (rc == TM_OK && history != nullptr) {
        hipLaunchKernelGGL(i8_record_diag_kernel, dim3(1), dim3(I8_W), 0, st, out, ldo, (int)m, history);
        TM_LAUNCH_CHECK();
    }
    return rc;
}

int run_syrk_i8(const double *X, int64_t n, int64_t m, const double *d, const double *colmax, double *out,
                double *colsum, int *history, hipStream_t st, const double *center = nullptr) {
    return run_syrk_i8_panel(X, m, n, m, d, colmax, out, m, colsum, history, st, center);
}

}  // namespace tmh

extern "C" {

int tm_dense_sandwich_i8_history_words(void) { return 4 + 2 * tmh::I8_W; }

int tm_dense_sandwich_i8_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                             double *out, void *stream) {
    return tmh::run_syrk_i8(X, n, m, d, colmax, out, nullptr, nullptr, tmh::as_stream(stream));
}

int tm_dense_sandwich_i8_xtd_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                 double *out, double *colsum, void *stream) {
    return tmh::run_syrk_i8(X, n, m, d, colmax, out, colsum, nullptr, tmh::as_stream(stream));
}

int tm_dense_sandwich_i8_hist_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                  double *out, double *colsum, int32_t *history, void *stream) {
    return tmh::run_syrk_i8(X, n, m, d, colmax, out, colsum, history, tmh::as_stream(stream));
}

int tm_dense_sandwich_i8_centered_f64(const double *X, int64_t n, int64_t m, const double *d, const double *colmax,
                                      const double *center, double *out, double *colsum, int32_t *history,
                                      void *stream) {
    return tmh::run_syrk_i8(X, n, m, d, colmax, out, colsum, history, tmh::as_stream(stream), center);
}

}  // extern "C"
