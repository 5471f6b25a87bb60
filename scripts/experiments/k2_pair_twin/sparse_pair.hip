// K2 (v4)  unrestricted sparse self sandwich  out = A^T diag(d) A  on the PAIR TWIN
// (reference: ext/sparse.pyx:17-77 sparse_sandwich).
//
// Why another form: the chunk-major kernel (sparse.hip, v3) is bound by VALU issue -- ~200 vector
// instructions per 8-row group and tile, of which only ~60 form pairs; the rest chases per-row
// chunk pointers, clamps addresses, builds keys and walks the slots 8..15 of every row whether
// they exist or not.  Here the stream is STATIC:
//   base   bv F[nch][G][64], bk int32[nch][G][64]   G = ceil(n / 8) groups of 8 rows; slot
//          (r, t) = r * 8 + t of group g in chunk c = the t-th nonzero (t < 8) of row 8 g + r inside
//          the 128-column chunk c, as {value, column inside the chunk}; -1 = padding.  Fixed
//          stride: no pointers, one coalesced 64-lane load per list and group.
//   overflow  a row's 9th, 10th ... nonzero of a chunk (5 % of the nonzeros at 5 % density), compact:
//          ov F[], ok int32[] = (row in group << 7) | column, ordered by (chunk, group, row);
//          optr int32[nch][G + 1] = first entry of every group.  An entry is broadcast from its lane
//          with v_readlane and paired with the 8 slots of its row (one ds_add with 8 live lanes).
// Pair generation inside a group is the v3 scheme: 8 lanes per row, lane t holds slot t of the I list
// and of the J list, step s = 0..7 brings the J entry of lane t ^ s with DPP moves, 8 ds_add_f64
// cover the 8 x 8 pairs of 8 rows; one signed compare decides validity and, on diagonal tiles, b <= a.
#include <stdlib.h>

#include "common.hpp"
#include "k2_common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr int PT_WAVES = 16;

template <typename F, int TS>
__global__ __launch_bounds__(PT_WAVES * 64) void sparse_sandwich_pair_kernel(
    const F *__restrict__ bv, const int32_t *__restrict__ bk, const F *__restrict__ ov,
    const int32_t *__restrict__ ok, const int32_t *__restrict__ optr, int64_t n_ov, int nch,
    const F *__restrict__ d, int64_t n, int64_t G, int nb_diag, int nb_off, int max_nb,
    F *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    F *tile = reinterpret_cast<F *>(smem_raw);  // [TS][TS], column-swizzled
    // 1-D grid as in v3: the nch diagonal tiles first (nb_diag workgroups each), then the
    // off-diagonal ones (nb_off each)
    int part, blk, nblk_part;
    {
        const int b = blockIdx.x;
        const int ndiag_blocks = nch * nb_diag;
        if (b < ndiag_blocks) {
            const int Id = b / nb_diag;
            blk = b % nb_diag;
            nblk_part = nb_diag;
            part = Id * (Id + 1) / 2 + Id;
        } else {
            const int o = (b - ndiag_blocks) / nb_off;
            blk = (b - ndiag_blocks) % nb_off;
            nblk_part = nb_off;
            int Io = (int)((sqrtf(8.0f * (float)o + 1.0f) - 1.0f) * 0.5f);
            while ((Io + 1) * (Io + 2) / 2 <= o) ++Io;
            while (Io * (Io + 1) / 2 > o) --Io;
            const int Jo = o - Io * (Io + 1) / 2;
            part = (Io + 1) * (Io + 2) / 2 + Jo;
        }
    }
    int I = (int)((sqrtf(8.0f * (float)part + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= part) ++I;
    while (I * (I + 1) / 2 > part) --I;
    const int J = part - I * (I + 1) / 2;
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) tile[b] = F(0);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane >> 3;
    const int64_t gpb = (G + nblk_part - 1) / nblk_part;      // groups per workgroup (< 2^31 / 64)
    const int64_t gA = (int64_t)blk * gpb;
    const int ng = (int)(min(gA + gpb, G) - gA);              // groups of this workgroup
    // uniform bases of this workgroup's range; everything below is a 32-bit offset from them
    const F *bvA = bv + ((int64_t)I * G + gA) * 64, *bvB = bv + ((int64_t)J * G + gA) * 64;
    const int32_t *bkA = bk + ((int64_t)I * G + gA) * 64, *bkB = bk + ((int64_t)J * G + gA) * 64;
    const int32_t *opA = optr + (int64_t)I * (G + 1) + gA, *opB = optr + (int64_t)J * (G + 1) + gA;
    const F *dW = d + gA * 8;
    const int rows_left = (int)min((int64_t)ng * 8, n - gA * 8);      // rows of the range that exist
    const unsigned ov_last = (unsigned)max((int64_t)0, min(n_ov - 1, (int64_t)0x7fffffff));

    auto run_tile = [&](auto diag_c) {
    constexpr bool DIAG = decltype(diag_c)::value;
    struct Ptr { int a0, a1, b0, b1; };
    struct Grp { F va, vb, dv, ova, ovb; int ka, kb, oka, okb, cA, cB; };
    auto load_ptrs = [&](int g) {
        Ptr q;
        const int gc = min(g, max(ng - 1, 0));
        q.a0 = opA[gc];
        q.a1 = opA[gc + 1];
        if constexpr (!DIAG) {
            q.b0 = opB[gc];
            q.b1 = opB[gc + 1];
        } else {
            q.b0 = q.b1 = 0;
        }
        return q;
    };
    auto load_group = [&](const Ptr &q, int g) {
        Grp e;
        const bool valid = g < ng;
        const int gc = min(g, max(ng - 1, 0));
        const unsigned off = (unsigned)gc * 64u + (unsigned)lane;
        e.va = bvA[off];
        e.ka = bkA[off];
        const int row = gc * 8 + lr;
        const F dv = dW[min(row, max(rows_left - 1, 0))];
        e.dv = (valid && row < rows_left) ? dv : F(0);
        e.cA = valid ? q.a1 - q.a0 : 0;
        const unsigned ia = min((unsigned)q.a0 + (unsigned)min(lane, max(e.cA - 1, 0)), ov_last);
        e.ova = ov[ia];
        e.oka = ok[ia];
        if constexpr (!DIAG) {
            e.vb = bvB[off];
            e.kb = bkB[off];
            e.cB = valid ? q.b1 - q.b0 : 0;
            const unsigned ib = min((unsigned)q.b0 + (unsigned)min(lane, max(e.cB - 1, 0)), ov_last);
            e.ovb = ov[ib];
            e.okb = ok[ib];
        } else {
            e.vb = e.ovb = F(0);
            e.kb = e.okb = 0;
            e.cB = 0;
        }
        return e;
    };
    // Pair keys as in v3, pre-scaled to byte offsets of the tile (<< SH):
    //   B entry: kb = col << SH, or a key above every limit for padding;
    //   A entry: limit la = (col << SH) | offmask (negative for padding), base ba = row base |
    //            column swizzle; target of (a, b) = tile + (kb ^ ba), wanted iff kb <= la.
    constexpr int SH = sizeof(F) == 8 ? 3 : 2;
    constexpr int offmask = DIAG ? 0 : 0x70000000;
    char *const tile_bytes = reinterpret_cast<char *>(tile);
    auto a_lim = [&](int col) { return (col << SH) | offmask; };          // col = -1: negative
    auto a_base = [&](int col) { return (int)(((unsigned)col << (7 + SH)) | ((col & 15) << (3 + SH))); };
    auto b_key = [&](int col) { return (col << SH) & 0x7fffffff; };       // col = -1: 0x7ffffff8
    auto add_pair = [&](int kb, int la, int ba, F prod) {
        if (kb <= la) atomic_add(reinterpret_cast<F *>(tile_bytes + (unsigned)(kb ^ ba)), prod);
    };
    auto process = [&](const Grp &cur) {
        const F dk = cur.dv;
        const bool on = dk != F(0);                 // rows with d == 0 (or beyond n) contribute nothing
        const int colA = on ? cur.ka : -1;
        const int colB = DIAG ? colA : (on ? cur.kb : -1);
        const int la = a_lim(colA), ba = a_base(colA), kb = b_key(colB);
        const F av = cur.va * dk, vb = DIAG ? cur.va : cur.vb;
        // base x base: all 8 x 8 slot pairs of the 8 rows
        add_pair(kb, la, ba, av * vb);
        add_pair(dpp_xor_i32<1>(kb), la, ba, av * dpp_xor<1>(vb));
        add_pair(dpp_xor_i32<2>(kb), la, ba, av * dpp_xor<2>(vb));
        add_pair(dpp_xor_i32<3>(kb), la, ba, av * dpp_xor<3>(vb));
        const int kb4 = dpp_xor_i32<4>(kb);
        const F vb4 = dpp_xor<4>(vb);
        add_pair(kb4, la, ba, av * vb4);
        add_pair(dpp_xor_i32<1>(kb4), la, ba, av * dpp_xor<1>(vb4));
        add_pair(dpp_xor_i32<2>(kb4), la, ba, av * dpp_xor<2>(vb4));
        add_pair(dpp_xor_i32<3>(kb4), la, ba, av * dpp_xor<3>(vb4));
        // overflow entries of the I list: each against the 8 J slots of its row, and against the J
        // overflow entries of the same row (diagonal tiles: the I overflow itself, columns <= its own)
        // (uniform by construction; say so, or the loops below become divergent waterfall loops)
        const int cA = __builtin_amdgcn_readfirstlane(cur.cA);
        const int cB = DIAG ? cA : __builtin_amdgcn_readfirstlane(cur.cB);
        const int okB = DIAG ? cur.oka : cur.okb;
        const F ovB = DIAG ? cur.ova : cur.ovb;
#ifndef PT_NO_OVA
        for (int e = 0; e < cA; ++e) {
            const int key = __builtin_amdgcn_readlane(cur.oka, e);
            const F val = readlane_f<F>(cur.ova, e);
            const int re = key >> 7, col = key & 127;
            const int la_e = a_lim(col), ba_e = a_base(col);
#ifdef PT_NO_DRE
            const F dre = F(1);
#else
            const F dre = readlane_f<F>(dk, re * 8);
            if (dre == F(0)) continue;
#endif
            const F ave = val * dre;
            add_pair(lr == re ? kb : 0x7ffffff8, la_e, ba_e, ave * vb);
#ifndef PT_NO_OVOV
            const int colo = okB & 127;
            const bool m2 = lane < cB && (okB >> 7) == re && (!DIAG || colo <= col);
            if (__any(m2)) {
                if (m2)
                    atomic_add(reinterpret_cast<F *>(tile_bytes + (unsigned)((colo << SH) ^ ba_e)),
                               ave * ovB);
            }
#endif
        }
#endif
#ifndef PT_NO_OVB
        if constexpr (!DIAG) {
            // overflow entries of the J list against the 8 I slots of their row
            for (int e = 0; e < cB; ++e) {
                const int key = __builtin_amdgcn_readlane(cur.okb, e);
                const F val = readlane_f<F>(cur.ovb, e);
                const int re = key >> 7, col = key & 127;
                add_pair(col << SH, lr == re ? la : -8, ba, av * val);
            }
        }
#endif
    };
    // software pipeline as in v3: two groups per turn, two turns per iteration with alternating
    // registers; the overflow pointers of a group are fetched two turns before its entries
    const int gstep = PT_WAVES;
    const int gw = wave;
    Grp ea[2], eb[2];
    Ptr ps[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) ea[i] = load_group(load_ptrs(gw + i * gstep), gw + i * gstep);
#pragma unroll
    for (int i = 0; i < 2; ++i) ps[i] = load_ptrs(gw + (2 + i) * gstep);
    for (int g = gw; g < ng; g += 4 * gstep) {
        eb[0] = load_group(ps[0], g + 2 * gstep);
        eb[1] = load_group(ps[1], g + 3 * gstep);
        ps[0] = load_ptrs(g + 4 * gstep);
        ps[1] = load_ptrs(g + 5 * gstep);
        process(ea[0]);
        if (g + gstep < ng) process(ea[1]);
        if (g + 2 * gstep >= ng) break;
        ea[0] = load_group(ps[0], g + 4 * gstep);
        ea[1] = load_group(ps[1], g + 5 * gstep);
        ps[0] = load_ptrs(g + 6 * gstep);
        ps[1] = load_ptrs(g + 7 * gstep);
        process(eb[0]);
        if (g + 3 * gstep < ng) process(eb[1]);
    }
    };
    if (I == J) run_tile(std::true_type{});
    else run_tile(std::false_type{});
    __syncthreads();
    F *dst = ws + ((int64_t)part * max_nb + blk) * (TS * TS);
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) {
        const int r = b / TS, c = b % TS;
        dst[b] = tile[r * TS + (c ^ ((r & 15) << 3))];
    }
}

template <typename F>
static int run_sparse_sandwich_pair(const F *bv, const int32_t *bk, const F *ov, const int32_t *ok,
                                    const int32_t *optr, int64_t n, int64_t m, int64_t n_ov, const F *d,
                                    F *out, hipStream_t st) {
    if (m == 0) return TM_OK;
    if (n == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)(m * m), st));
        return TM_OK;
    }
    constexpr int TS = 128;
    const int nchunk = (int)ceil_div(m, TS);
    const int n_parts = nchunk * (nchunk + 1) / 2;
    TM_REQUIRE(n_parts <= 65535, "too many sparse columns for the tiled sandwich");
    const int64_t G = ceil_div(n, 8);
    const size_t lds = sizeof(F) * (size_t)(TS * TS);
    const int n_off = n_parts - nchunk;
    // a diagonal tile issues the same 8 base atomics with half the live lanes and has no J-overflow
    // phase: ~0.7 of an off-diagonal tile per row
    static const double wdiag = getenv("TABMAT_AMD_PT_W") ? atof(getenv("TABMAT_AMD_PT_W")) : 0.7;
    int nb_off = n_off > 0 ? std::max(1, (int)(NUM_CU / (n_off + wdiag * nchunk))) : 0;
    int nb_diag = std::max(1, (NUM_CU - n_off * nb_off) / nchunk);
    if (n_off == 0) nb_off = nb_diag;
    const int cap = (int)std::max<int64_t>(1, ceil_div(n, 1024));
    nb_diag = std::min(nb_diag, cap);
    nb_off = std::min(nb_off, cap);
    {   // 32-bit slot offsets inside a workgroup's range: groups per workgroup * 64 < 2^31
        const int min_nb = (int)ceil_div(G, (int64_t)1 << 24);
        nb_diag = std::max(nb_diag, min_nb);
        nb_off = std::max(nb_off, min_nb);
    }
    TM_REQUIRE(n_ov < (1ll << 31), "sparse block too large for the pair twin (overflow >= 2^31)");
    const int64_t nblk = std::max(nb_diag, nb_off);
    const size_t tmp_bytes = (sizeof(F) * (size_t)n_parts * TS * TS + 255) / 256 * 256;
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)n_parts * (size_t)nblk * TS * TS + 256,
                           &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    auto kern = &sparse_sandwich_pair_kernel<F, TS>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (nb_diag != nb_off)
        TM_HIP(hipMemsetAsync(ws, 0, sizeof(F) * (size_t)n_parts * (size_t)nblk * TS * TS, st));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nchunk * nb_diag + n_off * nb_off)), dim3(PT_WAVES * 64),
                       lds, st, bv, bk, ov, ok, optr, n_ov, nchunk, d, n, G, nb_diag, nb_off, (int)nblk,
                       ws);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, (int64_t)TS * TS, (int)nblk, n_parts, tmp,
                                   (int64_t)n_parts * TS * TS, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((sparse_sandwich_assemble_kernel<F, TS>),
                       dim3((unsigned)ceil_div(m, 64), (unsigned)m), dim3(64), 0, st, tmp, (int)m,
                       nchunk, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {
int tm_sparse_sandwich_pair_f32(const float *bv, const int32_t *bk, const float *ov, const int32_t *ok,
                                const int32_t *optr, int64_t n, int64_t m, int64_t n_ov, const float *d,
                                float *out, void *stream) {
    return tmh::run_sparse_sandwich_pair<float>(bv, bk, ov, ok, optr, n, m, n_ov, d, out,
                                                tmh::as_stream(stream));
}
int tm_sparse_sandwich_pair_f64(const double *bv, const int32_t *bk, const double *ov,
                                const int32_t *ok, const int32_t *optr, int64_t n, int64_t m,
                                int64_t n_ov, const double *d, double *out, void *stream) {
    return tmh::run_sparse_sandwich_pair<double>(bv, bk, ov, ok, optr, n, m, n_ov, d, out,
                                                 tmh::as_stream(stream));
}
}  // extern "C"
