// K3 (lane-group form): sparse x dense sandwich  out = A^T diag(d) B  for a C-ordered dense
// operand with more than 64 columns (reference: ext/sparse.pyx:211-260 csr_dense_sandwich ->
// ext/sparse_helpers-tmpl.cpp:23-146 _csr_denseC_sandwich).
//
// Why another form of the gather kernel: round 1's csr_dense_ellw_kernel spends 13 wave
// instructions per nonzero (v_readlane + s_nop + v_add per row offset, a uniform LDS read per
// pair of values, three scalar instructions per pair of slots for the padding test) and runs at
// exactly that rate -- one wave instruction per cycle and CU.  Here a nonzero costs
//     v_add_u32_dpp (row offset, broadcast inside the lane's row of 16)      1
//     ds_read_b128 x 2 (f64: 2 x 2 dense columns per lane) / x 1 (f32)       2 / 1
//     v_fmac_f64_dpp x 4 / v_fmac_f32_dpp x 4 (value broadcast by DPP)       4
// per TWO nonzeros: the wave is cut into two halves of 32 lanes that work on different sparse
// columns at the same time (a half covers the 128 dense columns of its nonzero with 4 (f64) or
// 4 (f32) columns per lane).  `row_newbcast:s` hands every lane the {value, row offset} held by
// lane s of its own row of 16 lanes -- no v_readlane, no SGPR hazard, no LDS ring.
//
// Stream ("lane-group twin", built once per block by SlabLg.from_csr): rows in slabs of
// LG_R = 64, columns (sorted by density) in groups of LG_C = 16 = one wave.  Column w of a group
// belongs to half h = w / 8 and is the half's column j = w % 8.  A ROUND of a (slab, group) block
// is 4 chunks of 32 slots; chunk c holds columns j = 2c, 2c + 1 of both halves, 8 positions each:
//     slot  h * 16 + (j & 1) * 8 + it   =  the (8 * round + it)-th nonzero of column 8h + j
// as {value F, koff u32}; koff = (1 + row in slab) * row bytes, 0 = padding (value 0).  Row 0 of
// an LDS slab buffer is all zero and d[0] of its d-vector is 0, so padding needs no select.
// Round 0 of every block sits at a fixed stride (no pointer chase in the common case); slot 0 of
// chunk 0 carries the number of further rounds in koff bits 24..31, those live in a second pair
// of arrays addressed through xptr (columns with more than 8 nonzeros in a slab: 0.4 % at 5 %).
// Positions of a column are compacted (nonzeros first), so "position `it` of column j is used by
// either half" is monotone in `it`; the first LG_UNC positions run unconditionally, the rest in
// pairs behind one scalar bit test.
#include "common.hpp"
#include "reduce.hpp"
#include <stdlib.h>

namespace tmh {

constexpr int LG_R = 64;            // rows per slab
constexpr int LG_C = 16;            // sparse columns per wave
constexpr int LG_NW = 16;           // waves per workgroup (256 sparse columns)
constexpr int LG_THREADS = LG_NW * 64;
constexpr int LG_W = 128;           // dense columns per part
constexpr int LG_CHUNKS = 4;        // chunks per round
constexpr int LG_SLOTS = 32;        // distinct slots per chunk
constexpr unsigned LG_KMASK = 0xFFFFFu;

template <typename F>
struct LgLds {
    static constexpr int ROWB = LG_W * (int)sizeof(F);
    static constexpr int BUFB = (LG_R + 1) * ROWB;          // [zero row][LG_R rows]
    static constexpr int DL_OFF = 2 * BUFB;
    static constexpr int DLN = LG_R + 2;                       // [0][d of LG_R rows][pad]
    static constexpr int TOTAL = DL_OFF + 2 * DLN * (int)sizeof(F);
};

template <int SEL>
__device__ __forceinline__ unsigned lg_bcast_add(unsigned k, unsigned off) {
    unsigned r;
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "=v"(r)
        : "v"(k), "v"(off), "n"(SEL));
    return r;
}
#ifndef LG_ABL
#define LG_ABL 0
#endif
#ifndef LG_SPREAD
#define LG_SPREAD 1
#endif
template <int SEL>
__device__ __forceinline__ void lg_fmac(double &acc, double a, double x) {
#if LG_ABL == 1
    asm("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(x));
#elif LG_ABL == 2
    asm volatile("" ::"v"(x));
#else
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(a), "v"(x), "n"(SEL));
#endif
}
template <int SEL>
__device__ __forceinline__ void lg_fmac(float &acc, float a, float x) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(a), "v"(x), "n"(SEL));
}

#ifdef LG_PROF
__device__ long long lg_prof[256 * 16 * 8];
#define LG_T(x) const long long x = clock64()
#define LG_ACC(i, v) pacc[i] += (v)
#else
#define LG_T(x)
#define LG_ACC(i, v)
#endif

template <typename F, int UNC>
__global__ __launch_bounds__(LG_THREADS) void csr_dense_lg_kernel(
    const F *__restrict__ vals, const unsigned *__restrict__ koff, const int64_t *__restrict__ xptr,
    const F *__restrict__ xvals, const unsigned *__restrict__ xkoff, int n_groups, int64_t n_slabs,
    int64_t slabs_per_block, const F *__restrict__ B, int64_t n, int64_t r, int nB,
    const F *__restrict__ d, F *__restrict__ ws, int dbg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = LgLds<F>;
    constexpr int VEC = 16 / (int)sizeof(F);             // dense columns per 16-byte read
    constexpr int NRD = L::ROWB / 512;                    // reads per nonzero and lane (2 / 1)
    constexpr int ROWB = L::ROWB;
    constexpr int SLABB = LG_R * ROWB;
    constexpr int NV = SLABB / 16 / LG_THREADS;           // 16-byte pieces staged per thread
    constexpr int RPP = 1024 / ROWB;                       // slab rows per 1 KiB wave piece
    constexpr int LOGROW = ROWB == 1024 ? 10 : 9;
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.z * LG_NW + wave;
    const bool active = group < n_groups;
    const int j0 = blockIdx.y * LG_W;
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    const unsigned lane_off = (unsigned)(lane & 31) * 16u;
    const int lane32 = (lane >> 5) * 16 + (lane & 15);     // the slot this lane loads
    F *dl_all = reinterpret_cast<F *>(smem_raw + L::DL_OFF);
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_byte *)smem_raw;

    F acc[8][NRD][VEC];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int u = 0; u < NRD; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[j][u][e] = F(0);
    // zero rows of both buffers, d = 0 for them
    for (int i = tid; i < 2 * ROWB / (int)sizeof(F); i += LG_THREADS) {
        const int b = i / (ROWB / (int)sizeof(F)), c = i % (ROWB / (int)sizeof(F));
        reinterpret_cast<F *>(smem_raw + b * L::BUFB)[c] = F(0);
    }
    if (tid < 2) dl_all[tid * L::DLN] = F(0);

    F dsc = F(0);
    // async copy of B[slab rows, j0 .. j0 + 128): piece i of this wave's NV 1 KiB pieces
    auto issue_piece = [&](int64_t s, int buf, int i) {
        if (dbg & 1) return;
        const int piece = wave * NV + i;
        const int row = piece * RPP + (lane * 16) / ROWB;
        const int c = ((lane * 16) % ROWB) / (int)sizeof(F);
        const int64_t k = min(s * LG_R + row, n - 1);
        const int cc = min(j0 + c, nB - VEC);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(B + k * r + cc),
            (__attribute__((address_space(3))) void *)(smem_raw + buf * L::BUFB + ROWB + piece * 1024),
            16, 0, 0);
    };
    auto issue_slab = [&](int64_t s, int buf) {
#pragma unroll
        for (int i = 0; i < NV; ++i) issue_piece(s, buf, i);
        if (tid < LG_R) dsc = d[min(s * LG_R + tid, n - 1)];
    };
    auto finish_slab = [&](int64_t s, int buf) {
        if (tid < LG_R) dl_all[buf * L::DLN + 1 + tid] = (s * LG_R + tid < n) ? dsc : F(0);
    };

    // the four chunks of round 0 of the NEXT slab, requested while the current one is worked on
    F pv[LG_CHUNKS];
    unsigned pk[LG_CHUNKS];
#pragma unroll
    for (int c = 0; c < LG_CHUNKS; ++c) { pv[c] = F(0); pk[c] = 0u; }
    auto load_chunk = [&](int64_t s, int c) {
        if (!active || s >= s1) return;
        if ((dbg & 2) && s > s0) return;
        const int64_t q = ((s * n_groups + group) * LG_CHUNKS + c) * LG_SLOTS + lane32;
        pv[c] = vals[q];
        pk[c] = koff[q];
    };
    auto load_round0 = [&](int64_t s) {
#pragma unroll
        for (int c = 0; c < LG_CHUNKS; ++c) load_chunk(s, c);
    };

    if (s0 < s1) {
        load_round0(s0);
        issue_slab(s0, 0);
        finish_slab(s0, 0);
    }
    __syncthreads();
#ifdef LG_PROF
    long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int64_t s = s0; s < s1; ++s) {
        LG_T(t_top);
        const int buf = (int)((s - s0) & 1);
        const bool more = s + 1 < s1;
#if LG_SPREAD == 0
        F cv[LG_CHUNKS];
        unsigned ck[LG_CHUNKS];
#pragma unroll
        for (int c = 0; c < LG_CHUNKS; ++c) { cv[c] = pv[c]; ck[c] = pk[c]; }
        if (more) {
            issue_slab(s + 1, buf ^ 1);
            load_round0(s + 1);
        }
#else
        if (more && tid < LG_R) dsc = d[min((s + 1) * LG_R + tid, n - 1)];
        if (more && !(active && !(dbg & 4))) {      // a wave without columns still copies its share
#pragma unroll
            for (int i = 0; i < NV; ++i) issue_piece(s + 1, buf ^ 1, i);
        }
#endif
        LG_T(t_issue);
        LG_ACC(0, t_issue - t_top);
        if (active && !(dbg & 4)) {
            const F *dl = dl_all + buf * L::DLN;
            const unsigned slab_base = lds_base + (unsigned)(buf * L::BUFB);
            // one chunk: fold d into the values, redirect d == 0 rows to the zero row, then the
            // positions of columns j = 2c, 2c + 1 of both halves
            auto do_chunk = [&](auto cc, F v, unsigned kraw) {
                constexpr int c = decltype(cc)::value;
                const unsigned kk = kraw & LG_KMASK;
                const unsigned long long real = __builtin_amdgcn_ballot_w64(kk != 0u);
                const unsigned comb = ((unsigned)real | (unsigned)(real >> 32)) & 0xFFFFu;
                if (comb == 0u) return;
#if LG_ABL == 4
                F dk = F(1);
                asm volatile("" : "+v"(dk));
#else
                const F dk = dl[kk >> LOGROW];
#endif
                F a = v * dk;
                unsigned kq = slab_base + (dk != F(0) ? kk : 0u);      // absolute LDS address
                // DPP reads of a VGPR need two wait states after the VALU write
                asm volatile("s_nop 1" : "+v"(a), "+v"(kq));
                auto positions = [&](auto jlc, auto itc, auto cntc) {
                    constexpr int jl = decltype(jlc)::value;
                    constexpr int it = decltype(itc)::value;
                    constexpr int cnt = decltype(cntc)::value;
                    vec_t x[cnt][NRD];
                    static_for<cnt>([&](auto e) {
                        constexpr int SEL = jl * 8 + it + decltype(e)::value;
                        const unsigned addr = lg_bcast_add<SEL>(kq, lane_off);
#pragma unroll
                        for (int u = 0; u < NRD; ++u)
#if LG_ABL == 3
                        {
                            vec_t t;
                            asm volatile("" : "=v"(t) : "v"(addr));
                            x[decltype(e)::value][u] = t;
                        }
#elif LG_ABL == 5
                        {
                            if (u == 0) x[decltype(e)::value][u] = *reinterpret_cast<
                                const __attribute__((address_space(3))) vec_t *>(
                                (lds_byte *)(uintptr_t)(addr + u * 512));
                            else { vec_t t; asm volatile("" : "=v"(t) : "v"(addr)); x[decltype(e)::value][u] = t; }
                        }
#else
                            x[decltype(e)::value][u] = *reinterpret_cast<
                                const __attribute__((address_space(3))) vec_t *>(
                                (lds_byte *)(uintptr_t)(addr + u * 512));
#endif
                    });
                    static_for<cnt>([&](auto e) {
                        constexpr int SEL = jl * 8 + it + decltype(e)::value;
#pragma unroll
                        for (int u = 0; u < NRD; ++u)
#pragma unroll
                            for (int q = 0; q < VEC; ++q)
                                lg_fmac<SEL>(acc[2 * c + jl][u][q], a, x[decltype(e)::value][u][q]);
                    });
                };
                static_for<2>([&](auto jlc) {
                    constexpr int jl = decltype(jlc)::value;
                    positions(jlc, std::integral_constant<int, 0>{}, std::integral_constant<int, UNC>{});
                    if constexpr (UNC <= 2) {
                        if (comb & (1u << (jl * 8 + 2))) {
                            positions(jlc, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
                            if (comb & (1u << (jl * 8 + 4))) {
                                positions(jlc, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
                                if (comb & (1u << (jl * 8 + 6)))
                                    positions(jlc, std::integral_constant<int, 6>{}, std::integral_constant<int, 2>{});
                            }
                        }
                    } else {
                        if (comb & (1u << (jl * 8 + 4))) {
                            positions(jlc, std::integral_constant<int, 4>{}, std::integral_constant<int, 2>{});
                            if (comb & (1u << (jl * 8 + 6)))
                                positions(jlc, std::integral_constant<int, 6>{}, std::integral_constant<int, 2>{});
                        }
                    }
                });
            };
#if LG_SPREAD == 0
            const int extra = __builtin_amdgcn_readfirstlane((int)(ck[0] >> 24));
            static_for<LG_CHUNKS>([&](auto cc) { do_chunk(cc, cv[decltype(cc)::value], ck[decltype(cc)::value]); });
#else
            // the copy of the next slab and the loads of its chunks are issued BETWEEN the chunks
            // of this one: issued in one burst at the top of the iteration the 16 waves queue up
            // behind the vector-memory pipe for ~2000 cycles before any of them computes
            const int extra = __builtin_amdgcn_readfirstlane((int)(pk[0] >> 24));
            static_for<LG_CHUNKS>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                if (more) {
#pragma unroll
                    for (int i = c * NV / LG_CHUNKS; i < (c + 1) * NV / LG_CHUNKS; ++i)
                        issue_piece(s + 1, buf ^ 1, i);
                }
                const F v = pv[c];
                const unsigned k = pk[c];
                if (more) load_chunk(s + 1, c);
                do_chunk(cc, v, k);
            });
#endif
            if (extra > 0) {        // columns with more than 8 nonzeros in this slab: rare
                const int64_t xb = xptr[s * n_groups + group];
                for (int m = 0; m < extra; ++m) {
                    const int64_t q = ((xb + m) * LG_CHUNKS) * LG_SLOTS + lane32;
                    static_for<LG_CHUNKS>([&](auto cc) {
                        constexpr int c = decltype(cc)::value;
                        do_chunk(cc, xvals[q + c * LG_SLOTS], xkoff[q + c * LG_SLOTS]);
                    });
                }
            }
        }
        LG_T(t_comp);
        LG_ACC(1, t_comp - t_issue);
        if (s + 1 < s1) finish_slab(s + 1, buf ^ 1);
#ifdef LG_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
#endif
        LG_T(t_mem);
        LG_ACC(2, t_mem - t_comp);
        __syncthreads();
        LG_T(t_bar);
        LG_ACC(3, t_bar - t_mem);
        LG_ACC(4, 1);
    }
#ifdef LG_PROF
    if (lane == 0 && blockIdx.x < 128 && blockIdx.z < 2)
        for (int i = 0; i < 8; ++i) lg_prof[((blockIdx.z * 128 + blockIdx.x) * 16 + wave) * 8 + i] = pacc[i];
#endif
    if (active) {
        // ws layout: [part][block][n_groups * LG_C kernel columns][128]
        const int h = lane >> 5;
        F *dst = ws + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_groups + group) *
                          (LG_C * LG_W);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int u = 0; u < NRD; ++u) {
                vec_t o;
#pragma unroll
                for (int q = 0; q < VEC; ++q) o[q] = acc[j][u][q];
                *reinterpret_cast<vec_t *>(dst + (8 * h + j) * LG_W + u * (512 / (int)sizeof(F)) +
                                           (lane & 31) * VEC) = o;
            }
    }
}

// tmp [part][m][128] -> out[m][nB]
template <typename F>
__global__ void lg_untile_kernel(const F *__restrict__ tmp, int64_t m, int64_t nB,
                                 F *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * nB) return;
    const int64_t i = e / nB, j = e % nB;
    out[e] = tmp[((j / LG_W) * m + i) * LG_W + (j % LG_W)];
}

template <typename F>
static int run_csr_dense_lg(const F *vals, const unsigned *koff, const int64_t *xptr, const F *xvals,
                            const unsigned *xkoff, int64_t n, int64_t m, const F *B, int64_t r,
                            const F *d, int unc, F *out, hipStream_t st) {
    const int64_t nB = r;
    const int64_t total = m * nB;
    if (total == 0) return TM_OK;
    constexpr int VEC = 16 / (int)sizeof(F);
    if ((reinterpret_cast<uintptr_t>(B) & 15) != 0 || r % VEC != 0 || nB < VEC) {
        set_error("tm_csr_dense_sandwich_lg: B must be C-ordered with 16-byte aligned rows");
        return TM_EUNSUPPORTED;
    }
    if (m % LG_C != 0) {
        set_error("tm_csr_dense_sandwich_lg: m must be a multiple of tm_lg_group_cols()");
        return TM_EINVAL;
    }
    const int64_t n_slabs = ceil_div(n, LG_R);
    const int n_groups = (int)(m / LG_C);
    const int n_parts = (int)ceil_div(nB, LG_W);
    const int nz = (int)ceil_div(n_groups, LG_NW);
    if (n_slabs == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    int64_t nblk = std::max<int64_t>(1, NUM_CU / ((int64_t)n_parts * nz));
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const int64_t stride = m * LG_W;  // per (part, block)
    const size_t tmp_bytes = (sizeof(F) * (size_t)(n_parts * stride) + 255) / 256 * 256;
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_parts * nblk * stride) + 256,
                           &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    const size_t lds = (size_t)LgLds<F>::TOTAL;
    auto kern = unc >= 4 ? &csr_dense_lg_kernel<F, 4> : &csr_dense_lg_kernel<F, 2>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts, (unsigned)nz), dim3(LG_THREADS),
                       lds, st, vals, koff, xptr, xvals, xkoff, n_groups, n_slabs, spb, B, n, r,
                       (int)nB, d, ws, getenv("TM_LG_DBG") ? atoi(getenv("TM_LG_DBG")) : 0);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_parts, tmp, n_parts * stride, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((lg_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                       tmp, m, nB, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {
#ifdef LG_PROF
int tm_lg_prof_fetch(long long *host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(tmh::lg_prof), sizeof(long long) * 256 * 16 * 8);
}
#endif

int tm_lg_rows(void) { return tmh::LG_R; }
int tm_lg_group_cols(void) { return tmh::LG_C; }

int tm_csr_dense_sandwich_lg_f32(const float *vals, const uint32_t *koff, const int64_t *xptr,
                                 const float *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                 const float *B, int64_t r, const float *d, int unconditional,
                                 float *out, void *stream) {
    return tmh::run_csr_dense_lg<float>(vals, koff, xptr, xvals, xkoff, n, m, B, r, d, unconditional,
                                        out, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_lg_f64(const double *vals, const uint32_t *koff, const int64_t *xptr,
                                 const double *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                 const double *B, int64_t r, const double *d, int unconditional,
                                 double *out, void *stream) {
    return tmh::run_csr_dense_lg<double>(vals, koff, xptr, xvals, xkoff, n, m, B, r, d,
                                         unconditional, out, tmh::as_stream(stream));
}

}  // extern "C"
