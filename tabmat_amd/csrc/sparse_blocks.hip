// K2b  Sparse self sandwich  out = A' diag(d) A  (reference: ext/sparse.pyx:17-77), unrestricted, on a
// static BLOCK LIST.
//
// The chunked kernel (sparse.hip, sparse_sandwich_chunked_kernel) gives 8 lanes to every row of a
// 128 x 128 tile and forms the 8 x 8 pairs of the first 8 entries of the row's two lists with DPP
// moves; a 9th .. 16th entry (19 % of the rows at 5 % density) is broadcast one slot at a time over
// the 8 rows of the wave step -- 40 % of its LDS atomics are those overhang steps, issued with a
// handful of live lanes (profiles/r2_k2_k3_sq_counters.txt: ~20 of 64 lanes live per ds_add_f64,
// 12.5 VALU instructions per atomic).  The pattern of the matrix does not change between calls, so
// the overhang is resolved ONCE, at ingest: every (row, tile) is cut into BLOCKS of at most 8 x 8
// entries -- block (a, b) pairs entries 8a .. 8a+7 of the row's list in chunk I with entries
// 8b .. 8b+7 of its list in chunk J (b <= a on diagonal tiles) -- and the kernel walks a flat list
// of 16-byte block descriptors {first A entry, first B entry, row, counts}.  Every wave step is
// 8 blocks x 8 slots x 8 DPP steps of the same straight-line code: no ballots, no overhang
// phases, no long-list fallback, and the workgroups of a tile split its BLOCKS evenly.
// 1.42 blocks per (row, tile) at 5 % density; +16 B per block of HBM (2.3 GB at cfg4).
#include <algorithm>

#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr int KB_WAVES = 16;
constexpr int KB_TS = 128;

// out[i][j] (n_out x n_out) from the reduced tile buffer [part][TS * TS]; mirror included
template <typename F>
__global__ void blocks_assemble_kernel(const F *__restrict__ tiles, int n_out, F *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= n_out || j >= n_out) return;
    const int hi = max(i, j), lo = min(i, j);
    const int I = hi / KB_TS, J = lo / KB_TS;
    const int part = I * (I + 1) / 2 + J;
    out[(int64_t)i * n_out + j] = tiles[(int64_t)part * KB_TS * KB_TS + (hi % KB_TS) * KB_TS + (lo % KB_TS)];
}

// Block classes.  A block whose two sides both hold more than 4 entries is FULL: 8 lanes x 8 DPP
// steps (lane t meets the B entry of lane t ^ s, s = 0 .. 7).  Every other block is HALF and needs
// only the 4 quad-local steps s = 0 .. 3, with all 8 lanes live when one side is long:
//   A side <= 4 entries, B side > 4:  lane t loads A[t & 3] and B[t]   -> pairs (t & 3, t ^ s)
//   A side > 4, B side <= 4:          lane t loads A[t] and B[t & 3]   -> pairs (t, (t & 3) ^ s)
//   both <= 4:                        lanes 0 .. 3 only
// (each pair exactly once).  5.7 steps per block on average at 5 % density instead of 8.  The host
// sorts the blocks of a workgroup by class and splits its 16 waves between the two lists in
// proportion to their cost, so that both lists are walked in row order AT THE SAME PACE: the
// entries a FULL block and a HALF block of neighbouring rows share a cache line of are then read
// while the line is still in the L2 (one list after the other re-streams every line: 6.2 ms).
constexpr int KB_FLAG_REP_A = 1 << 16;     // lane t takes A entry t & 3
constexpr int KB_FLAG_REP_B = 1 << 17;     // lane t takes B entry t & 3

// U8: `ind` points at BYTES, the column of an entry inside its 128-column chunk (a quarter of the index bytes the
// entry gathers move).
// D12 (round 6): 12-byte descriptors {first A entry, first B entry, row | nA - 1 << 24 | nB - 1 << 27 | flags << 30}
// for blocks of fewer than 2^24 rows (0.56 GB less at BASELINE configs[3]; the descriptors are read coalesced and
// once: same time as the 16-byte form, profiles/r5_k2b.txt).
template <typename F, bool U8, bool D12 = false>
__global__ __launch_bounds__(KB_WAVES * 64) void sparse_sandwich_blocks_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int32_t *__restrict__ cptr,
    int64_t n, int64_t nnz1, const int32_t *__restrict__ blocks, const int4 *__restrict__ wg_tab,
    const F *__restrict__ d, int max_nb, F *__restrict__ ws, WgLogBuf *__restrict__ wglog) {
    constexpr int TS = KB_TS;
    const unsigned long long t_begin = wg_log_begin(wglog);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);   // [TS][TS] doubles, column-swizzled
    // wg_tab: 2 int4 per workgroup: {part, slot, first block, end}, {end of the FULL list, waves on
    // the FULL list, first row, last row}
    const int4 w = wg_tab[2 * blockIdx.x], x = wg_tab[2 * blockIdx.x + 1];
    const int part = __builtin_amdgcn_readfirstlane(w.x), slot = __builtin_amdgcn_readfirstlane(w.y);
    const int b0 = __builtin_amdgcn_readfirstlane(w.z), b1 = __builtin_amdgcn_readfirstlane(w.w);
    const int bfull = __builtin_amdgcn_readfirstlane(x.x);
    // waves on the FULL list: the table is built for KB_WAVES waves; a launch with fewer (tm_tune_set
    // "k2b_waves") rescales the split -- at least one wave per non-empty list, or blocks would be skipped
    int wfull = __builtin_amdgcn_readfirstlane(x.y);
    {
        const int nwv = (int)(blockDim.x >> 6);
        if (nwv != KB_WAVES) {
            wfull = (wfull * nwv + KB_WAVES / 2) / KB_WAVES;
            if (bfull > b0) wfull = max(wfull, 1);
            if (b1 > bfull) wfull = min(wfull, nwv - 1);
            if (bfull == b0) wfull = 0;
            if (b1 == bfull) wfull = nwv;
        }
    }
    int I = (int)((sqrtf(8.0f * (float)part + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= part) ++I;
    while (I * (I + 1) / 2 > part) --I;
    const int J = part - I * (I + 1) / 2;
    const int i0 = I * TS, j0 = J * TS;
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane >> 3, lt = lane & 7;          // block of the wave step, entry slot
    if (b1 > b0) {
        // 32-bit offsets relative to the first entry of the workgroup's row range in chunk I / J
        // (uniform bases in SGPRs, as in the chunked kernel)
        const int row_first = __builtin_amdgcn_readfirstlane(x.z), row_last = __builtin_amdgcn_readfirstlane(x.w);
        const int32_t *cpA = cptr + (int64_t)I * (n + 1), *cpB = cptr + (int64_t)J * (n + 1);
        const int baseA = __builtin_amdgcn_readfirstlane(cpA[row_first]);
        const int baseB = __builtin_amdgcn_readfirstlane(cpB[row_first]);
        const unsigned spanA1 = (unsigned)max(__builtin_amdgcn_readfirstlane(cpA[row_last + 1]) - baseA - 1, 0);
        const unsigned spanB1 = (unsigned)max(__builtin_amdgcn_readfirstlane(cpB[row_last + 1]) - baseB - 1, 0);
        const F *dataA = data + min((int64_t)baseA, nnz1), *dataB = data + min((int64_t)baseB, nnz1);
        const int32_t *indA = ind + min((int64_t)baseA, nnz1), *indB = ind + min((int64_t)baseB, nnz1);
        const unsigned char *ind8 = reinterpret_cast<const unsigned char *>(ind);
        const unsigned char *ind8A = ind8 + min((int64_t)baseA, nnz1), *ind8B = ind8 + min((int64_t)baseB, nnz1);
        // (byte columns are chunk-relative: the tile origin is already taken off)
        const int i0c = U8 ? 0 : i0, j0c = U8 ? 0 : j0;
        const unsigned nrow1 = (unsigned)max(n - 1, (int64_t)0);
        auto ldi = [](const int32_t *base, unsigned i) {
            return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(base) + (i << 2));
        };
        auto ldf = [](const F *base, unsigned i) {
            return *reinterpret_cast<const F *>(reinterpret_cast<const char *>(base) + (i * (unsigned)sizeof(F)));
        };
        // pair keys as in the chunked kernel: target of (a, b) = tile + (kb ^ ba), wanted iff kb <= la
        constexpr int SH = 3;
        constexpr int BIGKEY = 0x7ffffff0;
        char *const tile_bytes = reinterpret_cast<char *>(tile);
        auto add_pair = [&](int kb, int la, int ba, F prod) {
            if (kb <= la)
                atomic_add(reinterpret_cast<lds_acc_t *>(tile_bytes + (unsigned)(kb ^ ba)), (lds_acc_t)prod);
        };

        // one class list [s0, s1) walked by the waves w0 .. w0 + nw - 1: descriptors two turns ahead,
        // entries one turn ahead, two wave steps per turn (as in the chunked kernel)
        auto run_list = [&](auto diag_c, auto full_c, int s0, int s1, int w0, int nw) {
            constexpr bool DIAG = decltype(diag_c)::value;
            constexpr bool FULL = decltype(full_c)::value;
            constexpr int offmask = DIAG ? 0 : 0x70000000;
            const int nseg = s1 - s0;
            if (nseg <= 0) return;
            constexpr int DW = D12 ? 3 : 4;                 // words per descriptor
            const int32_t *blk = blocks + (int64_t)s0 * DW;
            struct Dsc { int4 q; bool valid; };
            struct Grp { int nA, nB, ta, tb; F d, va, vb; int ca, cb; };
            auto load_desc = [&](int g) {
                Dsc r;
                const int k = g + lr;
                r.valid = k < nseg;
                const unsigned kc = (unsigned)min(k, nseg - 1);
                if constexpr (D12) {
                    const int32_t *b = reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(blk) + kc * 12u);
                    const int x = b[0], y = b[1];
                    const unsigned w = (unsigned)b[2];
                    r.q = int4{x, y, (int)(w & 0xffffffu),
                               (int)((((w >> 24) & 7u) + 1u) | ((((w >> 27) & 7u) + 1u) << 8) | (((w >> 30) & 3u) << 16))};
                } else {
                    r.q = *reinterpret_cast<const int4 *>(reinterpret_cast<const char *>(blk) + (kc << 4));
                }
                return r;
            };
            auto load_entries = [&](const Dsc &sd) {
                Grp e;
                const unsigned row = min((unsigned)sd.q.z, nrow1);
                e.d = ldf(d, row);
                const bool on = sd.valid && e.d != F(0);      // rows with d == 0 contribute nothing
                e.nA = on ? (sd.q.w & 0xff) : 0;
                e.nB = on ? ((sd.q.w >> 8) & 0xff) : 0;
                // entry index of this lane on either side (a HALF block with one long side
                // replicates its short side: t & 3)
                e.ta = (!FULL && (sd.q.w & KB_FLAG_REP_A)) ? (lt & 3) : lt;
                e.tb = (!FULL && (sd.q.w & KB_FLAG_REP_B)) ? (lt & 3) : lt;
                const unsigned iA = min((unsigned)(sd.q.x - baseA) + min((unsigned)e.ta, (unsigned)max(e.nA - 1, 0)), spanA1);
                const unsigned iB = min((unsigned)(sd.q.y - baseB) + min((unsigned)e.tb, (unsigned)max(e.nB - 1, 0)), spanB1);
#ifdef KB_ABL_WRAP   // ablation (timing only, wrong results): every entry gather lands in a window the L2 holds
                const unsigned iAw = iA & (unsigned)(KB_ABL_WRAP), iBw = iB & (unsigned)(KB_ABL_WRAP);
                e.ca = ldi(indA, min(iAw, spanA1));
                e.va = ldf(dataA, min(iAw, spanA1));
                e.cb = ldi(indB, min(iBw, spanB1));
                e.vb = ldf(dataB, min(iBw, spanB1));
                return e;
#endif
                e.ca = U8 ? (int)ind8A[iA] : ldi(indA, iA);
                e.va = ldf(dataA, iA);
                e.cb = U8 ? (int)ind8B[iB] : ldi(indB, iB);
                e.vb = ldf(dataB, iB);
                return e;
            };
            auto process = [&](const Grp &cur) {
                // the pair steps run at a raised wave priority (ahead of the other waves' address arithmetic and
                // gathers, which wait for memory anyway): 4.08 -> 4.02 ms (profiles/r4_k2b.txt)
                __builtin_amdgcn_s_setprio(1);
                const int colA = cur.ta < cur.nA ? cur.ca - i0c : -1;
                const int colB = cur.tb < cur.nB ? cur.cb - j0c : -1;
                const int la = (colA < 0 ? -8 : colA << SH) | offmask;
                const int ba = (int)(((unsigned)colA << (7 + SH)) | ((colA & 15) << (3 + SH)));
                const int kb = colB < 0 ? BIGKEY : colB << SH;
                const F av = cur.va * cur.d, vb = cur.vb;
                add_pair(kb, la, ba, av * vb);
                add_pair(dpp_xor_i32<1>(kb), la, ba, av * dpp_xor<1>(vb));
                add_pair(dpp_xor_i32<2>(kb), la, ba, av * dpp_xor<2>(vb));
                add_pair(dpp_xor_i32<3>(kb), la, ba, av * dpp_xor<3>(vb));
#ifndef KB_ABL_HALFSTEPS   // ablation (timing only, wrong results): FULL blocks stop after 4 of their 8 steps
                if constexpr (FULL) {
                    const int kb4 = dpp_xor_i32<4>(kb);
                    const F vb4 = dpp_xor<4>(vb);
                    add_pair(kb4, la, ba, av * vb4);
                    add_pair(dpp_xor_i32<1>(kb4), la, ba, av * dpp_xor<1>(vb4));
                    add_pair(dpp_xor_i32<2>(kb4), la, ba, av * dpp_xor<2>(vb4));
                    add_pair(dpp_xor_i32<3>(kb4), la, ba, av * dpp_xor<3>(vb4));
                }
#endif
                __builtin_amdgcn_s_setprio(0);
            };
            // software pipeline as in the chunked kernel: descriptors two turns ahead, entries one turn
            // ahead, two wave steps per turn, two turns per iteration with alternating registers (a ring
            // of 4 entry sets with 3 steps of lead was measured slower: 1.93 vs 1.64 ms at 4M rows)
            const int gstep = nw * 8;
            const int gw = (wave - w0) * 8;
            Grp ea[2], eb[2];
            Dsc ps[2];
            ea[0] = load_entries(load_desc(gw));
            ea[1] = load_entries(load_desc(gw + gstep));
            ps[0] = load_desc(gw + 2 * gstep);
            ps[1] = load_desc(gw + 3 * gstep);
            for (int g = gw; g < nseg; g += 4 * gstep) {
                eb[0] = load_entries(ps[0]);
                eb[1] = load_entries(ps[1]);
                ps[0] = load_desc(g + 4 * gstep);
                ps[1] = load_desc(g + 5 * gstep);
                process(ea[0]);
                if (g + gstep < nseg) process(ea[1]);
                if (g + 2 * gstep >= nseg) break;
                ea[0] = load_entries(ps[0]);
                ea[1] = load_entries(ps[1]);
                ps[0] = load_desc(g + 6 * gstep);
                ps[1] = load_desc(g + 7 * gstep);
                process(eb[0]);
                if (g + 3 * gstep < nseg) process(eb[1]);
            }
        };
        auto run_tile = [&](auto diag_c) {
            if (wave < wfull) run_list(diag_c, std::true_type{}, b0, bfull, 0, wfull);
            else run_list(diag_c, std::false_type{}, bfull, b1, wfull, (int)(blockDim.x >> 6) - wfull);
        };
        if (I == J) run_tile(std::true_type{});
        else run_tile(std::false_type{});
    }
    __syncthreads();
    if (threadIdx.x == 0) wg_log_end(wglog, t_begin, WG_K2);
    F *dst = ws + ((int64_t)part * max_nb + slot) * (TS * TS);
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) {
        const int r = b / TS, c = b % TS;
        dst[b] = (F)tile[r * TS + (c ^ ((r & 15) << 3))];
    }
}

template <typename F, bool U8 = false, bool D12 = false>
static int run_sparse_sandwich_blocks(const F *data, const int32_t *ind, const int32_t *cptr, int64_t n,
                                      int64_t m, int64_t nnz, const int32_t *blocks, const int32_t *wg_tab,
                                      int n_wg, int max_nb, const F *d, F *out, hipStream_t st) {
    if (m == 0) return TM_OK;
    if (n == 0 || nnz == 0 || n_wg == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)(m * m), st));
        return TM_OK;
    }
    constexpr int TS = KB_TS;
    const int nchunk = (int)ceil_div(m, TS);
    const int n_parts = nchunk * (nchunk + 1) / 2;
    TM_REQUIRE(nnz < (1ll << 31) && n < (1ll << 29), "sparse block too large for the block-list sandwich");
    TM_REQUIRE(!D12 || n < (1ll << 24), "12-byte block descriptors hold 24 bits of row");
    TM_REQUIRE(max_nb >= 1 && n_wg >= 1, "empty workgroup table");
    const size_t lds = sizeof(lds_acc_t) * (size_t)(TS * TS);
    const size_t tmp_bytes = (sizeof(F) * (size_t)n_parts * TS * TS + 255) / 256 * 256;
    const size_t ws_bytes = sizeof(F) * (size_t)n_parts * (size_t)max_nb * TS * TS;
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + ws_bytes + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    // tiles have different numbers of workgroups: unused partial slots must read as 0
    TM_HIP(hipMemsetAsync(ws, 0, ws_bytes, st));
    auto kern = &sparse_sandwich_blocks_kernel<F, U8, D12>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    // (the wave split of the table -- wg_tab[5] = waves on the FULL list -- is built for this many waves)
    const int kb_waves = (int)std::min<int64_t>(KB_WAVES, std::max<int64_t>(2, tune("k2b_waves", KB_WAVES)));
    hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(kb_waves * 64), lds, st, data, ind, cptr, n, nnz - 1,
                       blocks, reinterpret_cast<const int4 *>(wg_tab), d,
                       max_nb, ws, wg_log_ptr());
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, (int64_t)TS * TS, max_nb, n_parts, tmp, (int64_t)n_parts * TS * TS,
                                   false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((blocks_assemble_kernel<F>), dim3((unsigned)ceil_div(m, 64), (unsigned)m), dim3(64),
                       0, st, tmp, (int)m, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_sparse_sandwich_blocks_f32(const float *cm_data, const int32_t *cm_indices, const int32_t *cptr,
                                  int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                  const int32_t *wg_tab, int n_wg, int max_nb, const float *d, float *out,
                                  void *stream) {
    return tmh::run_sparse_sandwich_blocks<float>(cm_data, cm_indices, cptr, n, m, nnz, blocks, wg_tab, n_wg,
                                                  max_nb, d, out, tmh::as_stream(stream));
}
int tm_sparse_sandwich_blocks_f64(const double *cm_data, const int32_t *cm_indices, const int32_t *cptr,
                                  int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                  const int32_t *wg_tab, int n_wg, int max_nb, const double *d, double *out,
                                  void *stream) {
    return tmh::run_sparse_sandwich_blocks<double>(cm_data, cm_indices, cptr, n, m, nnz, blocks, wg_tab, n_wg,
                                                   max_nb, d, out, tmh::as_stream(stream));
}

// the same with the columns as one byte per entry: column inside the entry's 128-column chunk
int tm_sparse_sandwich_blocks_u8_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                     int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                     const int32_t *wg_tab, int n_wg, int max_nb, const float *d, float *out,
                                     void *stream) {
    return tmh::run_sparse_sandwich_blocks<float, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), cptr, n, m,
                                                        nnz, blocks, wg_tab, n_wg, max_nb, d, out,
                                                        tmh::as_stream(stream));
}
int tm_sparse_sandwich_blocks_u8_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                     int64_t n, int64_t m, int64_t nnz, const int32_t *blocks,
                                     const int32_t *wg_tab, int n_wg, int max_nb, const double *d, double *out,
                                     void *stream) {
    return tmh::run_sparse_sandwich_blocks<double, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), cptr, n, m,
                                                         nnz, blocks, wg_tab, n_wg, max_nb, d, out,
                                                         tmh::as_stream(stream));
}

// byte columns AND 12-byte descriptors (blocks: int32 [B][3], see the kernel; n < 2^24)
int tm_sparse_sandwich_blocks_p12_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                      int64_t n, int64_t m, int64_t nnz, const int32_t *blocks12,
                                      const int32_t *wg_tab, int n_wg, int max_nb, const float *d, float *out,
                                      void *stream) {
    return tmh::run_sparse_sandwich_blocks<float, true, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), cptr,
                                                              n, m, nnz, blocks12, wg_tab, n_wg, max_nb, d, out,
                                                              tmh::as_stream(stream));
}
int tm_sparse_sandwich_blocks_p12_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *cptr,
                                      int64_t n, int64_t m, int64_t nnz, const int32_t *blocks12,
                                      const int32_t *wg_tab, int n_wg, int max_nb, const double *d, double *out,
                                      void *stream) {
    return tmh::run_sparse_sandwich_blocks<double, true, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), cptr,
                                                               n, m, nnz, blocks12, wg_tab, n_wg, max_nb, d, out,
                                                               tmh::as_stream(stream));
}

}  // extern "C"
