// K2e  Sparse self sandwich  out = A' diag(d) A  (reference: ext/sparse.pyx:17-77), unrestricted, for WIDE
// blocks: cost in proportion to the entries and PAIRS of nonzeros of a row, as the reference's loop has it
// (ext/sparse.pyx:55-74: for every entry of a row, every later entry of the row -- whatever the column count).
//
// The tiled kernels (sparse.hip chunked, sparse_blocks.hip block list) give a fixed group of lanes to every
// (row, 128 x 128 tile) and pay for it whether the row has six entries in each of the tile's two chunks
// (BASELINE configs[3]: 512 columns at 5 %) or one and a half (2048 columns at 1.25 %: the same 25.6 nonzeros
// and 340 pairs per row cost 11x more, profiles/r4_regimes.txt; one-hot ingest and from_csc produce such blocks).
// The direct kernel (sparse_direct.hip) is pair-proportional but pays one L2 atomic per pair (~20 G/s).
//
// Here the output still lives in LDS tile by tile (part = (I, J), J <= I, 128 x 128 doubles), but the work of a
// tile is a STREAM OF ENTRIES, lane <-> entry a of column chunk I (chunk-major RECORD twin: one coalesced 16-byte
// load of {value, column, row}); the lane looks up its row's list in chunk J (its two chunk pointers as one
// 8-byte load; neighbouring lanes share or neighbour rows) and walks it: one ds_add_f64 per pair.  Nothing is
// spent on rows without entries in chunk I, nothing on empty J lists beyond the pointer pair.  Diagonal tiles: the
// J list of entry a is its own row's list up to a itself (the lists are in column order), so every unordered pair
// is produced once and no column compare is needed.
//
// What bounds it is the vector memory pipe: ~25 CU cycles per scattered load instruction (the same figure the
// block-list kernel shows), hence one 16-byte record per entry instead of three arrays (14 -> 7 loads per step),
// the first KP_T0 list entries of a step requested one step ahead, and three pipeline stages (entry -> row weight
// and pointers -> list heads).  Versions, ablations and the two forms that lost (strip x block tiles with long
// lists; more steps in flight per wave): profiles/r5_k2_pairs.txt.
//
// The rows are cut into as many contiguous segments as a tile has workgroups and the grid is segment-major: the
// workgroups resident at one time sweep the SAME segment for all the tiles, so the entries of a row range are read
// from HBM once and by the other tiles from the L2 / Infinity Cache.
#include <algorithm>

#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr int KP_TS = 128;
constexpr int KP_WAVES = 16;
constexpr int KP_RANGE = 2048;          // granule of the row segments
#ifndef KP_T0_N
#define KP_T0_N 4
#endif
constexpr int KP_T0 = KP_T0_N;          // list entries requested one step ahead per lane (longer lists: in a loop)
typedef int32_t kp_rec_t __attribute__((ext_vector_type(4)));

// out[i][j] (n_out x n_out) from the reduced tile buffer [part][TS * TS] (lower triangle of tiles); mirror included
template <typename F>
__global__ void pairs_assemble_kernel(const F *__restrict__ tiles, int n_out, F *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= n_out || j >= n_out) return;
    const int hi = max(i, j), lo = min(i, j);
    const int I = hi / KP_TS, J = lo / KP_TS;
    const int part = I * (I + 1) / 2 + J;
    out[(int64_t)i * n_out + j] = tiles[(int64_t)part * KP_TS * KP_TS + (hi % KP_TS) * KP_TS + (lo % KP_TS)];
}

// PK: packed records -- {value, row << 7 | column inside the chunk}: 12 bytes for f64, 8 for f32 (blocks of fewer than
// 2^25 rows) -- instead of the 16-byte ones.
template <typename F, bool PK>
__global__ __launch_bounds__(KP_WAVES * 64) void sparse_sandwich_pairs_kernel(
    const void *__restrict__ rec_v, const int32_t *__restrict__ cptr, int64_t n, const F *__restrict__ d,
    int n_slots, F *__restrict__ ws) {
    const kp_rec_t *__restrict__ rec = reinterpret_cast<const kp_rec_t *>(rec_v);
    constexpr int RB = PK ? (int)sizeof(F) + 4 : 16;                 // bytes per record
    // a record as four words {value lo, value hi (f32: unused), column, row}, whatever the stored form
    auto fetch = [&](int p) -> kp_rec_t {
        if constexpr (!PK) {
            return rec[p];
        } else if constexpr (sizeof(F) == 8) {
            typedef int32_t i3 __attribute__((ext_vector_type(3), aligned(4)));
            const i3 t = *reinterpret_cast<const i3 *>(reinterpret_cast<const char *>(rec_v) + (int64_t)p * RB);
            return kp_rec_t{t[0], t[1], (int)((unsigned)t[2] & 127u), (int)((unsigned)t[2] >> 7)};
        } else {
            typedef int32_t i2 __attribute__((ext_vector_type(2)));
            const i2 t = *reinterpret_cast<const i2 *>(reinterpret_cast<const char *>(rec_v) + (int64_t)p * RB);
            return kp_rec_t{t[0], (int)((unsigned)t[1] & 127u), (int)((unsigned)t[1] >> 7), 0};
        }
    };
    // rec[p] = {value (f64: low word, high word; f32: bits), column, row} of chunk-major entry p (f32: {value,
    // column, row, 0})
    constexpr int TS = KP_TS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);   // [TS][TS] doubles
    // segment-major grid
    const int n_parts = (int)(gridDim.x / (unsigned)n_slots);
    const int part = blockIdx.x % n_parts, slot = blockIdx.x / n_parts;
    int I = (int)((sqrtf(8.0f * (float)part + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= part) ++I;
    while (I * (I + 1) / 2 > part) --I;
    const int J = part - I * (I + 1) / 2;
    const bool diag = I == J;
    const int i0 = PK ? 0 : I * TS, j0 = PK ? 0 : J * TS;     // (packed records carry chunk-relative columns)
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwave = (int)(blockDim.x >> 6);
    const int32_t *cpI = cptr + (int64_t)I * (n + 1);
    const int32_t *cpJ = cptr + (int64_t)J * (n + 1);
    const int64_t n_ranges = (n + KP_RANGE - 1) / KP_RANGE;

    // One wave step = 64 consecutive entries of chunk I, lane <-> entry.  Three dependent memory hops feed a step
    // -- the entry, then its row's weight and J-list pointers, then the list's first T0 entries -- so the steps of
    // a wave are software-pipelined three deep: hop h of step s + 3 - h is requested while step s adds its pairs.
    constexpr int T0 = KP_T0;
    struct Ent { int col, row; F val; bool on; };
    struct Lst { int lo, k; F w; int ca; };
    struct Pre { kp_rec_t r[T0]; };
    auto rec_val = [](const kp_rec_t &r) -> F {
        if constexpr (sizeof(F) == 8) return (F)__hiloint2double(r[1], r[0]);
        else return (F)__int_as_float(r[0]);
    };
    auto rec_col = [](const kp_rec_t &r) { return sizeof(F) == 8 ? r[2] : r[1]; };
    auto rec_row = [](const kp_rec_t &r) { return sizeof(F) == 8 ? r[3] : r[2]; };
    auto load_ent = [&](int p, int pend) {
        Ent e;
        e.on = p < pend;
        const int pp = e.on ? p : max(pend - 1, 0);
        kp_rec_t r;
        if constexpr (PK) r = fetch(pp);
        else r = __builtin_nontemporal_load(rec + pp);
        e.col = rec_col(r);
        e.val = rec_val(r);
        e.row = rec_row(r);
        return e;
    };
    auto load_lst = [&](const Ent &e, int p) {
        Lst l;
#if defined(KP_ABL_NOD)              // timing only: no gather of d
        const F dv = F(1);
#else
        const F dv = d[e.row];
#endif
        // (the two chunk pointers of the row as one 8-byte load: 4-byte aligned, which global loads accept)
        typedef int32_t i2 __attribute__((ext_vector_type(2), aligned(4)));
#if defined(KP_ABL_NOCP)             // timing only (wrong results): no gather of the chunk pointers
        const i2 lh = i2{p, p + ((e.row & 3) == 0 ? 2 : 1)};
#else
        const i2 lh = *reinterpret_cast<const i2 *>(cpJ + e.row);
#endif
        // diagonal tile: the row's list up to the entry itself; else its whole list in chunk J
        const int hi = diag ? p + 1 : lh[1];
        l.lo = lh[0];
        // (rows with d == 0 contribute nothing: an excluded row may hold inf / nan)
        l.k = (e.on && dv != F(0)) ? hi - l.lo : 0;
        l.w = e.val * dv;
        l.ca = (e.col - i0) * TS - j0;
        return l;
    };
    auto load_pre = [&](const Lst &l) {
        Pre q;
        // (clamped to the list's last entry: unconditional loads, no wait inside a lane branch)
        const int last = l.lo + max(l.k - 1, 0);
#pragma unroll
        for (int t = 0; t < T0; ++t) q.r[t] = fetch(min(l.lo + t, last));
        return q;
    };
    auto run = [&](const Lst &l, const Pre &q) {
#pragma unroll
        for (int t = 0; t < T0; ++t)
            if (t < l.k) atomic_add(&tile[l.ca + rec_col(q.r[t])], (lds_acc_t)(l.w * rec_val(q.r[t])));
        // longer lists (2.4 % of the lanes at 1.6 entries per row and chunk): trip by trip
        int t = T0;
        while (__builtin_amdgcn_ballot_w64(t < l.k) != 0) {
            if (t < l.k) {
                const kp_rec_t r = fetch(l.lo + t);
                atomic_add(&tile[l.ca + rec_col(r)], (lds_acc_t)(l.w * rec_val(r)));
            }
            ++t;
        }
    };

    {
        // this slot's rows: one contiguous segment (whole ranges of KP_RANGE rows)
        const int64_t rg0 = n_ranges * slot / n_slots, rg1 = n_ranges * (slot + 1) / n_slots;
        const int64_t r0 = rg0 * KP_RANGE, r1 = min(rg1 * KP_RANGE, n);
        const int pa0 = __builtin_amdgcn_readfirstlane(cpI[r0]);
        const int pa1 = __builtin_amdgcn_readfirstlane(cpI[r1]);
        const int nstep = (pa1 - pa0 + 63) / 64;
        const int stride = nwave * 64;
        if (wave < nstep) {
            // prologue: step s0 = wave fully fetched, s0 + nwave up to its lists, s0 + 2 nwave its entries
            int p = pa0 + wave * 64 + lane;            // position of this lane's entry in step `wave`
            Ent e = load_ent(p, pa1);
            Lst l0 = load_lst(e, p);
            Pre q0 = load_pre(l0);
            e = load_ent(p + stride, pa1);
            Lst l1 = load_lst(e, p + stride);
            e = load_ent(p + 2 * stride, pa1);
            for (int s = wave; s < nstep; s += nwave) {
                const Pre q1 = load_pre(l1);                        // list heads of step s + nwave
                const Lst l2 = load_lst(e, p + 2 * stride);         // lists of step s + 2 nwave
                e = load_ent(p + 3 * stride, pa1);                  // entries of step s + 3 nwave
                run(l0, q0);
                l0 = l1;
                q0 = q1;
                l1 = l2;
                p += stride;
            }
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)part * n_slots + slot) * (TS * TS);
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) dst[b] = (F)tile[b];
}

template <typename F, bool PK = false>
static int run_sparse_sandwich_pairs(const int32_t *rec, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                     const F *d, F *out, hipStream_t st) {
    if (m == 0) return TM_OK;
    if (n == 0 || nnz == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)(m * m), st));
        return TM_OK;
    }
    constexpr int TS = KP_TS;
    // entry positions are 32-bit ints and the stream walk looks 3 strides of KP_WAVES * 64 entries ahead of the
    // segment's end (load_ent: p + 3 * stride): the margin keeps that sum below 2^31
    TM_REQUIRE(nnz < (1ll << 31) - 4ll * KP_WAVES * 64 && n < (1ll << 31) - 1,
               "sparse block too large for 32-bit entry positions");
    const int nchunk = (int)ceil_div(m, TS);
    TM_REQUIRE(nchunk <= 128, "at most 16384 columns (the tile partials are kept per workgroup)");
    const int n_parts = nchunk * (nchunk + 1) / 2;
    const int64_t n_ranges = ceil_div(n, KP_RANGE);
    // ~3 rounds of workgroups over the chip, the same number of row segments for every tile
    int n_slots = (int)std::max<int64_t>(1, std::min<int64_t>(n_ranges, tune("k2p_rounds", 3) * NUM_CU / n_parts));
    n_slots = std::min(n_slots, 64);
    const size_t lds = sizeof(lds_acc_t) * (size_t)(TS * TS);
    const size_t tmp_bytes = (sizeof(F) * (size_t)n_parts * TS * TS + 255) / 256 * 256;
    // (one row segment: the workgroups write the assembled-tile buffer themselves)
    const size_t ws_bytes = n_slots > 1 ? sizeof(F) * (size_t)n_parts * (size_t)n_slots * TS * TS : 0;
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + ws_bytes + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = n_slots > 1 ? reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes) : tmp;
    auto kern = &sparse_sandwich_pairs_kernel<F, PK>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int waves = (int)std::min<int64_t>(KP_WAVES, std::max<int64_t>(1, tune("k2p_waves", KP_WAVES)));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)(n_parts * n_slots)), dim3(waves * 64), lds, st,
                       reinterpret_cast<const void *>(rec), cptr, n, d, n_slots, ws);
    prof_end(st);
    TM_LAUNCH_CHECK();
    if (n_slots > 1) {
        rc = launch_reduce_partials<F>(ws, (int64_t)TS * TS, n_slots, n_parts, tmp, (int64_t)n_parts * TS * TS, false, st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL((pairs_assemble_kernel<F>), dim3((unsigned)ceil_div(m, 64), (unsigned)m), dim3(64), 0, st,
                       tmp, (int)m, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_sparse_sandwich_pairs_f32(const int32_t *cm_rec, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                 const float *d, float *out, void *stream) {
    return tmh::run_sparse_sandwich_pairs<float>(cm_rec, cptr, n, m, nnz, d, out, tmh::as_stream(stream));
}
int tm_sparse_sandwich_pairs_f64(const int32_t *cm_rec, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                 const double *d, double *out, void *stream) {
    return tmh::run_sparse_sandwich_pairs<double>(cm_rec, cptr, n, m, nnz, d, out, tmh::as_stream(stream));
}

// packed records: {value, row << 7 | column inside the chunk} -- 12 bytes (f64) / 8 bytes (f32) per entry, n < 2^25;
// the record array must be readable 4 bytes beyond its last record
int tm_sparse_sandwich_pairs_pk_f32(const int32_t *cm_rec_pk, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                    const float *d, float *out, void *stream) {
    if (n >= (1ll << 25)) { tmh::set_error("packed pair records need fewer than 2^25 rows"); return TM_EUNSUPPORTED; }
    return tmh::run_sparse_sandwich_pairs<float, true>(cm_rec_pk, cptr, n, m, nnz, d, out, tmh::as_stream(stream));
}
int tm_sparse_sandwich_pairs_pk_f64(const int32_t *cm_rec_pk, const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                    const double *d, double *out, void *stream) {
    if (n >= (1ll << 25)) { tmh::set_error("packed pair records need fewer than 2^25 rows"); return TM_EUNSUPPORTED; }
    return tmh::run_sparse_sandwich_pairs<double, true>(cm_rec_pk, cptr, n, m, nnz, d, out, tmh::as_stream(stream));
}

}  // extern "C"
