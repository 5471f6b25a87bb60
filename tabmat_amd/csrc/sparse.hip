// Sparse-block kernels for gfx950.  All of them stream the CSR twin of the block
// (sparse_matrix.py:133-143) row by row -- the natural order for row-sharding -- and
// accumulate into an LDS-privatised slice of the output with wavefront atomics (ds_add),
// followed by the deterministic reduce_partials pass.
//
//   K2  sparse self sandwich   (reference: ext/sparse.pyx:17-77)
//   K3  sparse x dense cross   (reference: ext/sparse_helpers-tmpl.cpp:23-146)
//   K6  CSR matvec / transpose-matvec (reference: ext/sparse.pyx:79-199)
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr size_t SP_LDS_MAX = 128 * 1024;

__device__ __forceinline__ int64_t ceil_div_dev(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------
// K6a  out[Ci] += sum_j X[rows[Ci], j] v[j]   -- G lanes per row
// ---------------------------------------------------------------------------------------
template <typename F, int G>
__global__ __launch_bounds__(256) void csr_matvec_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ v, const int32_t *__restrict__ rows, int64_t n_iter,
    const int32_t *__restrict__ col_map, F *__restrict__ out) {
    const int sl = threadIdx.x % G;
    const int64_t g0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int64_t ng = ((int64_t)gridDim.x * blockDim.x) / G;
    for (int64_t t = g0; t < n_iter; t += ng) {
        const int64_t row = rows ? (int64_t)rows[t] : t;
        const int64_t p1 = ptr[row + 1];
        F acc = F(0);
        for (int64_t p = ptr[row] + sl; p < p1; p += G) {
            const int j = ind[p];
            if (!col_map || col_map[j] >= 0) acc += data[p] * v[j];
        }
#pragma unroll
        for (int off = G / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, G);
        if (sl == 0) out[t] += acc;
    }
}

// ---------------------------------------------------------------------------------------
// K6b  out[col_map[j]] += sum_{i in rows} X[i, j] v[i]  -- LDS bins over the output columns
// ---------------------------------------------------------------------------------------
template <typename F, int G, bool USE_LDS>
__global__ __launch_bounds__(256) void csr_rmatvec_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ v, const int32_t *__restrict__ rows, int64_t n_iter,
    int64_t rows_per_block, const int32_t *__restrict__ col_map, int n_out, F *__restrict__ ws,
    F *__restrict__ out, int square) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *bins = reinterpret_cast<lds_acc_t *>(smem_raw);          // doubles (see common.hpp)
    if (USE_LDS) {
        for (int b = threadIdx.x; b < n_out; b += blockDim.x) bins[b] = 0.0;
        __syncthreads();
    }
    const int sl = threadIdx.x % G;
    const int grp = threadIdx.x / G;
    constexpr int NG = 256 / G;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);
    for (int64_t t = t0 + grp; t < t1; t += NG) {
        const int64_t row = rows ? (int64_t)rows[t] : t;
        const F vi = v[row];
        const int64_t p1 = ptr[row + 1];
        for (int64_t p = ptr[row] + sl; p < p1; p += G) {
            const int j = ind[p];
            const int oc = col_map ? col_map[j] : j;
            if (oc >= 0) {
                const F x = square ? data[p] * data[p] : data[p];
                if (USE_LDS) atomic_add(&bins[oc], (lds_acc_t)(x * vi));
                else atomic_add(&out[oc], x * vi);
            }
        }
    }
    if (USE_LDS) {
        __syncthreads();
        F *dst = ws + (int64_t)blockIdx.x * n_out;
        for (int b = threadIdx.x; b < n_out; b += blockDim.x) dst[b] = (F)bins[b];
    }
}

// ---------------------------------------------------------------------------------------
// K6 fast paths (all rows, all columns): the CSR arrays are STREAMED with coalesced loads --
// lane <-> nonzero, independent of the row structure -- instead of being walked row by row.
// A wave owns CSR_RPW consecutive rows; their nonzeros [p0, p1) pass through a per-wave LDS
// buffer in chunks of CSR_CAP entries:
//   matvec :  stage  prod[e] = data[e] * v[ind[e]]  (v gathered from an LDS copy), then lane <->
//             row sums its own segment of the buffer and adds it to out[row];
//   rmatvec:  lane <-> row first fills  vrow[e] = v[row]  over its segment, then lane <-> nonzero
//             adds  data[e] * vrow[e]  to the workgroup's LDS bins[ind[e]]  (ds_add), partials
//             are combined by reduce_partials_kernel.
// ---------------------------------------------------------------------------------------
constexpr int CSR_RPW = 64;          // rows per wave
constexpr int CSR_CAP = 1024;        // staged entries per wave and chunk
constexpr int CSR_WAVES = 4;
constexpr int CSR_EPL = CSR_CAP / 64;   // entries per lane and tile

// Round 5: the TRANSPOSE stream kernel is software-pipelined (matvec: see below).  Round 2's versions loaded a tile with four loads per
// array in flight, worked on it, and only then asked for the next one: with the 16 waves of a CU in step the
// memory pipe idled through every compute phase (0.63 / 0.48 of the HBM peak).  Now a wave owns a CONTIGUOUS range
// of row chunks and walks its nonzeros as one stream of tiles of CSR_CAP entries; a tile is fetched with 16- / 8-byte
// vector loads (lane <-> two adjacent entries, eight pairs per lane, all sixteen loads issued back to back) into
// registers ONE TILE AHEAD of the tile being worked on, and the row pointers / vector entries of a chunk one chunk
// ahead.  A tile starts at an even entry (vector alignment): the entry in front of a chunk's first one, if any, is
// fetched with it and ignored.
// (the column pairs stay as loaded -- int32 pairs or one packed word of two uint16 -- and are taken apart where they
// are used: unpacking at load time made the compiler wait for every index load in turn)
template <typename F, typename IDX = int32_t>
struct CsrTile {
    F x[CSR_EPL];
    int32_t j[CSR_EPL];
    __device__ __forceinline__ void set_pair(int k, int32_t a, int32_t b) { j[2 * k] = a; j[2 * k + 1] = b; }
    __device__ __forceinline__ void load_pair(int k, const int32_t *p) {
        typedef int32_t i2 __attribute__((ext_vector_type(2)));
        const i2 v = __builtin_nontemporal_load(reinterpret_cast<const i2 *>(p));
        j[2 * k] = v[0];
        j[2 * k + 1] = v[1];
    }
    __device__ __forceinline__ int32_t col(int i) const { return j[i]; }
};
template <typename F>
struct CsrTile<F, uint16_t> {
    F x[CSR_EPL];
    uint32_t jw[CSR_EPL / 2];
    __device__ __forceinline__ void set_pair(int k, int32_t a, int32_t b) { jw[k] = (uint32_t)a | ((uint32_t)b << 16); }
    __device__ __forceinline__ void load_pair(int k, const uint16_t *p) {
        jw[k] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(p));
    }
    __device__ __forceinline__ int32_t col(int i) const {
        return (i & 1) ? (int32_t)(jw[i >> 1] >> 16) : (int32_t)(jw[i >> 1] & 0xffffu);
    }
};

// entries [c0, c0 + cnt) of the stream (data + c0 16- / 8-byte aligned, cnt <= CSR_CAP; nnz = entries in the arrays):
// lane holds e = 2 lane + 128 k + {0, 1}.  Pairs are fetched whole; a pair whose second entry lies behind the tile
// is fetched anyway (the consumers ignore slots >= cnt) unless it would leave the arrays -- only the first tile
// of an odd-aligned view (c0 = -1) and the last tile of the arrays can, and they take the entry-by-entry path.
template <typename F, typename IDX>
__device__ __forceinline__ void csr_tile_load(const F *__restrict__ data, const IDX *__restrict__ ind, int64_t c0,
                                              int cnt, int64_t nnz, int lane, CsrTile<F, IDX> &t) {
    typedef F f2 __attribute__((ext_vector_type(2)));
    const F *dp = data + c0 + 2 * lane;
    const IDX *ip = ind + c0 + 2 * lane;
    if (cnt == CSR_CAP && c0 >= 0) {
#pragma unroll
        for (int k = 0; k < CSR_EPL / 2; ++k) {
            const f2 xv = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(dp + 128 * k));
            t.load_pair(k, ip + 128 * k);
            t.x[2 * k] = xv[0];
            t.x[2 * k + 1] = xv[1];
        }
        return;
    }
    if (c0 >= 0 && c0 + cnt + 1 <= nnz) {
        // a partial tile inside the arrays: whole pairs under one predicate each -- no element-wise merges, so
        // none of these loads is waited for before its use (the entry-by-entry path below serialises them)
#pragma unroll
        for (int k = 0; k < CSR_EPL / 2; ++k) {
            f2 xv = f2{F(0), F(0)};
            t.set_pair(k, 0, 0);
            if (2 * lane + 128 * k < cnt) {
                xv = __builtin_nontemporal_load(reinterpret_cast<const f2 *>(dp + 128 * k));
                t.load_pair(k, ip + 128 * k);
            }
            t.x[2 * k] = xv[0];
            t.x[2 * k + 1] = xv[1];
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < CSR_EPL / 2; ++k) {
        const int e = 2 * lane + 128 * k;
        F x0 = F(0), x1 = F(0);
        int32_t j0 = 0, j1 = 0;
        if (e < cnt && c0 + e >= 0) {
            x0 = dp[128 * k];
            j0 = (int32_t)ip[128 * k];
        }
        if (e + 1 < cnt) {
            x1 = dp[128 * k + 1];
            j1 = (int32_t)ip[128 * k + 1];
        }
        t.x[2 * k] = x0;
        t.x[2 * k + 1] = x1;
        t.set_pair(k, j0, j1);
    }
}

// a wave-uniform 64-bit value out of lane `src` (scalar registers: the tile bookkeeping stays off the VALU)
__device__ __forceinline__ int64_t csr_uniform(int64_t v, int src) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)(uint64_t)v, src);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)((uint64_t)v >> 32), src);
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// The tile walk shared by both kernels.  chunk range [cb, ce) of this wave; per chunk c lane <-> row c * 64 + lane
// with its entry range [rlo, rhi); `Per` = per-row payload fetched with the pointers (rmatvec: v[row]).
// body(tile, c0, cnt, rlo, rhi, first, payload) works on one tile; end_chunk(row, payload) closes a chunk.
template <typename F, typename Per, typename IDX, typename LoadPer, typename Body, typename EndChunk>
__device__ __forceinline__ void csr_stream_walk(const F *__restrict__ data, const IDX *__restrict__ ind,
                                                const int64_t *__restrict__ ptr, int64_t n, int64_t cb, int64_t ce,
                                                int lane, LoadPer load_per, Body body, EndChunk end_chunk) {
    if (cb >= ce) return;
    // element parity of the arrays' first entry (a row-sliced view starts anywhere): tiles start where data + c0
    // is vector-aligned (the host checked that data and ind share the parity)
    const int64_t off = (int64_t)((reinterpret_cast<uintptr_t>(data) / sizeof(F)) & 1);
    const int64_t nnz = ptr[n];
    auto load_ptrs = [&](int64_t c, int64_t &lo, int64_t &hi, Per &pp) {
        const int64_t row = c * CSR_RPW + lane;
        lo = ptr[min(row, n)];
        hi = ptr[min(row + 1, n)];
        pp = load_per(row);
    };
    int64_t c = cb, rlo, rhi, rloN = 0, rhiN = 0;
    Per per, perN = Per();
    load_ptrs(c, rlo, rhi, per);
    if (c + 1 < ce) load_ptrs(c + 1, rloN, rhiN, perN);
    int64_t p0 = csr_uniform(rlo, 0), p1 = csr_uniform(rhi, 63);
    int64_t c0 = p0 - ((p0 + off) & 1);
    CsrTile<F, IDX> A, B;
    csr_tile_load(data, ind, c0, (int)min((int64_t)CSR_CAP, p1 - c0), nnz, lane, A);
    // one step: request the tile after (c, c0) into `nxt`, work on `cur`; false when `cur` was the last one
    auto step = [&](CsrTile<F, IDX> &cur, CsrTile<F, IDX> &nxt) -> bool {
        const bool same = c0 + CSR_CAP < p1;           // the chunk goes on in the next tile
        const bool more = same || c + 1 < ce;
        int64_t c0n = c0 + CSR_CAP, p0n = p0, p1n = p1;
        int64_t rlo2 = rlo, rhi2 = rhi;
        Per per2 = per;
        if (!same && more) {
            rlo2 = rloN;
            rhi2 = rhiN;
            per2 = perN;
            p0n = csr_uniform(rlo2, 0);
            p1n = csr_uniform(rhi2, 63);
            c0n = p0n - ((p0n + off) & 1);
            if (c + 2 < ce) load_ptrs(c + 2, rloN, rhiN, perN);
        }
        if (more) csr_tile_load(data, ind, c0n, (int)min((int64_t)CSR_CAP, p1n - c0n), nnz, lane, nxt);
        body(cur, c0, (int)min((int64_t)CSR_CAP, p1 - c0), rlo, rhi, (int)(p0 > c0 ? p0 - c0 : 0), per);
        if (!same) {
            end_chunk(c * CSR_RPW + lane, per);
            ++c;
        }
        c0 = c0n;
        p0 = p0n;
        p1 = p1n;
        rlo = rlo2;
        rhi = rhi2;
        per = per2;
        return more;
    };
    while (true) {
        if (!step(A, B)) break;
        if (!step(B, A)) break;
    }
}

// matvec keeps round 2's form (a tile of products staged in LDS with four loads per array in flight, lane <-> row
// sums its segment): the pipelined walk above brought it nothing -- same-box A/B 0.641 vs 0.625 ms, 0.63 vs 0.65 of
// the HBM peak -- because this kernel lives on the LDS pipe (16 random gathers of v + the stores and segment reads of
// the products per 1024 entries), not on its loads; an 8-lanes-per-row form without LDS staging took 0.87 ms
// (profiles/r5_matvec.txt).
// IDX: int32 column indices, or uint16 (a twin of the index array for blocks of at most 65536 columns: 10 instead
// of 12 bytes per entry leave HBM)
template <typename F, typename IDX = int32_t>
__global__ __launch_bounds__(CSR_WAVES * 64) void csr_matvec_stream_kernel(
    const F *__restrict__ data, const IDX *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ v, int64_t n, int m, F *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    F *vl = reinterpret_cast<F *>(smem_raw);                       // [m]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    F *prod = vl + m + wave * CSR_CAP;                             // [CSR_CAP] per wave
    for (int j = threadIdx.x; j < m; j += blockDim.x) vl[j] = v[j];
    __syncthreads();
    const int64_t nchunk = ceil_div_dev(n, (int64_t)CSR_RPW);
    for (int64_t c = (int64_t)blockIdx.x * CSR_WAVES + wave; c < nchunk;
         c += (int64_t)gridDim.x * CSR_WAVES) {
        const int64_t row = c * CSR_RPW + lane;
        const int64_t rlo = ptr[min(row, n)];
        const int64_t rhi = ptr[min(row + 1, n)];
        const int64_t p0 = __shfl(rlo, 0, 64);
        const int64_t p1 = ptr[min(c * CSR_RPW + CSR_RPW, n)];
        F acc = F(0);
        for (int64_t c0 = p0; c0 < p1; c0 += CSR_CAP) {
            const int cnt = (int)min((int64_t)CSR_CAP, p1 - c0);
#pragma unroll 4
            for (int e = lane; e < cnt; e += 64)
                prod[e] = __builtin_nontemporal_load(data + c0 + e) * vl[(int)__builtin_nontemporal_load(ind + c0 + e)];
            __builtin_amdgcn_wave_barrier();
            const int lo = (int)(max(rlo, c0) - c0);
            const int hi = (int)(min(rhi, c0 + CSR_CAP) - c0);
            for (int e = lo; e < hi; ++e) acc += prod[e];
            __builtin_amdgcn_wave_barrier();
        }
        if (row < n) out[row] += acc;
    }
}

template <typename F, typename IDX = int32_t>
__global__ __launch_bounds__(CSR_WAVES * 64) void csr_rmatvec_stream_kernel(
    const F *__restrict__ data, const IDX *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ v, int64_t n, int m, int64_t chunks_per_wave, F *__restrict__ ws,
    int square) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *bins = reinterpret_cast<lds_acc_t *>(smem_raw);     // [m] doubles
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    F *vrow = reinterpret_cast<F *>(bins + ((m + 1) & ~1)) + wave * CSR_CAP;    // [CSR_CAP] per wave, 16-byte aligned
    for (int j = threadIdx.x; j < m; j += blockDim.x) bins[j] = 0.0;
    __syncthreads();
    const int64_t nchunk = ceil_div_dev(n, (int64_t)CSR_RPW);
    const int64_t cb = ((int64_t)blockIdx.x * CSR_WAVES + wave) * chunks_per_wave;
    const int64_t ce = min(cb + chunks_per_wave, nchunk);
    typedef F f2 __attribute__((ext_vector_type(2)));
    csr_stream_walk<F, F, IDX>(
        data, ind, ptr, n, cb, ce, lane, [&](int64_t row) { return row < n ? v[row] : F(0); },
        [&](const CsrTile<F, IDX> &t, int64_t c0, int cnt, int64_t rlo, int64_t rhi, int first, F vr) {
            // lane <-> row: v[row] over the row's part of the tile, then lane <-> entry
            const int lo = (int)(max(rlo, c0) - c0);
            const int hi = (int)(min(rhi, c0 + cnt) - c0);
            for (int e = lo; e < hi; ++e) vrow[e] = vr;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k = 0; k < CSR_EPL / 2; ++k) {
                const int e = 2 * lane + 128 * k;
                const f2 vv = *reinterpret_cast<const f2 *>(vrow + e);
                // (slots in front of the chunk's first entry and behind the tile's last were not written: skipped)
                if (e >= first && e < cnt) {
                    const F x = t.x[2 * k];
                    atomic_add(&bins[t.col(2 * k)], (lds_acc_t)((square ? x * x : x) * vv[0]));
                }
                if (e + 1 >= first && e + 1 < cnt) {
                    const F x = t.x[2 * k + 1];
                    atomic_add(&bins[t.col(2 * k + 1)], (lds_acc_t)((square ? x * x : x) * vv[1]));
                }
            }
            __builtin_amdgcn_wave_barrier();
        },
        [](int64_t, F) {});
    __syncthreads();
    F *dst = ws + (int64_t)blockIdx.x * m;
    for (int j = threadIdx.x; j < m; j += blockDim.x) dst[j] = (F)bins[j];
}

// ---------------------------------------------------------------------------------------
// K3 (v1)  tile[col_map_A[i]][jb] += (A[k,i] * d[k]) * B[k, B_cols[j0 + jb]]
// The output [nA x nB] is split by B-COLUMN ranges of TB columns (parts = blockIdx.y) so that
// a whole nA x TB slice sits in LDS; SUB = 64 / TB rows are processed per wave step, the
// TB lanes of a sub-group hold d[k]*B[k, j0 + jb] in a register and walk the sparse row.
// ---------------------------------------------------------------------------------------
template <typename F, int TB, bool ORDER_F>
__global__ __launch_bounds__(256) void csr_dense_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ B, int64_t n, int64_t r, const F *__restrict__ d,
    const int32_t *__restrict__ rows, int64_t n_iter, int64_t rows_per_block,
    const int32_t *__restrict__ a_map, int nA, const int32_t *__restrict__ B_cols, int nB,
    F *__restrict__ ws, int64_t stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [nA][TB] doubles
    const int part = blockIdx.y;
    const int j0 = part * TB;
    const int nel = nA * TB;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();

    constexpr int SUB = 64 / TB;
    const int lane = threadIdx.x & 63;
    const int sub = lane / TB;
    const int jb = lane % TB;
    const int wave = threadIdx.x >> 6;
    const int nwave = blockDim.x >> 6;
    const int jc = j0 + jb;
    const bool jok = jc < nB;
    const int64_t jcol = jok ? (B_cols ? (int64_t)B_cols[jc] : jc) : 0;
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);

    for (int64_t base = t0 + (int64_t)wave * SUB; base < t1; base += (int64_t)nwave * SUB) {
        const int64_t t = base + sub;
        const bool rok = t < t1;
        const int64_t k = rok ? (rows ? (int64_t)rows[t] : t) : 0;
        F bd = F(0);
        int64_t p = 0, p1 = 0;
        if (rok) {
            p = ptr[k];
            p1 = ptr[k + 1];
            if (jok) bd = d[k] * (ORDER_F ? B[jcol * n + k] : B[k * r + jcol]);
        }
        for (; p < p1; ++p) {
            const int a = a_map ? a_map[ind[p]] : ind[p];
            if (a >= 0 && jok) atomic_add(&tile[a * TB + jb], (lds_acc_t)(data[p] * bd));
        }
    }
    __syncthreads();
    F *dst = ws + ((int64_t)part * gridDim.x + blockIdx.x) * stride;
    for (int b = threadIdx.x; b < nel; b += blockDim.x) dst[b] = (F)tile[b];
}

// ---------------------------------------------------------------------------------------
// K2 (v1)  sparse self sandwich.  Output columns (after col_map) are cut into chunks of TS;
// part (I, J), J <= I, owns the TS x TS tile out[I-chunk, J-chunk] in LDS.  One wave per row:
// the row's entries are compacted into the A list (columns in chunk I) and the B list (columns
// in chunk J) in per-wave LDS scratch, then the lanes enumerate the |A| x |B| pairs.
// For I == J only pairs with colB <= colA are taken (lower triangle incl. diagonal), exactly
// the `i > j: break` of ext/sparse.pyx:64-67; the mirror happens after the reduction.
// ---------------------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ F readlane_f(F v, int l);

template <>
__device__ __forceinline__ double readlane_f<double>(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <>
__device__ __forceinline__ float readlane_f<float>(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

constexpr int K2_WAVES = 16;

template <typename F>
struct __attribute__((aligned(16))) K2Entry {
    F val;
    int col;
};

template <typename F, int TS>
__global__ __launch_bounds__(K2_WAVES * 64) void sparse_sandwich_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind, const int64_t *__restrict__ ptr,
    const F *__restrict__ d, const int32_t *__restrict__ rows, int64_t n_iter,
    int64_t rows_per_block, const int32_t *__restrict__ col_map, int n_out,
    F *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);                   // [TS][TS] doubles
    typedef K2Entry<F> Ent;
    Ent *scratch = reinterpret_cast<Ent *>(smem_raw + sizeof(lds_acc_t) * TS * TS);   // [waves][2][64]
    // part -> (I, J), J <= I:  part = I (I + 1) / 2 + J
    int I = (int)((sqrtf(8.0f * (float)blockIdx.y + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.y) ++I;
    while (I * (I + 1) / 2 > (int)blockIdx.y) --I;
    const int J = (int)blockIdx.y - I * (I + 1) / 2;
    const int i0 = I * TS, j0 = J * TS;
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    Ent *sa = scratch + (wave * 2 + 0) * 64;
    Ent *sb = scratch + (wave * 2 + 1) * 64;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const int la = lane >> 3, lb = lane & 7;          // 8 x 8 pair block per wave step
    const int64_t t0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n_iter);

    // Software pipeline over the rows of this wave (t, t + 16, ...): while row t is processed the
    // first 64 entries of row t + 16 and the row pointers of row t + 32 are already in flight,
    // so the three dependent memory round trips (row pointers -> entries -> LDS) overlap.
    auto row_of = [&](int64_t t) -> int64_t { return rows ? (int64_t)rows[t] : t; };
    int64_t p0_2 = 0, p1_2 = 0;   // stage 2: pointers of row t + 2*W
    F d_2 = F(0);
    int64_t p0_1 = 0, p1_1 = 0;   // stage 1: pointers + first chunk of row t + W
    F d_1 = F(0), val_1 = F(0);
    int col_1 = -1;
    auto load_ptrs = [&](int64_t t) {
        p0_2 = p1_2 = 0;
        d_2 = F(0);
        if (t < t1) {
            const int64_t k = row_of(t);
            p0_2 = ptr[k];
            p1_2 = ptr[k + 1];
            d_2 = d[k];
        }
    };
    auto load_chunk = [&]() {   // stage 2 -> stage 1, issue the entry loads
        p0_1 = p0_2; p1_1 = p1_2; d_1 = d_2;
        col_1 = -1;
        val_1 = F(0);
        if (p0_1 + lane < p1_1) {
            col_1 = ind[p0_1 + lane];
            val_1 = data[p0_1 + lane];
        }
    };
    const int64_t tw = t0 + wave;
    load_ptrs(tw);
    load_chunk();
    load_ptrs(tw + K2_WAVES);

    for (int64_t t = tw; t < t1; t += K2_WAVES) {
        const int64_t p0 = p0_1, p1 = p1_1;
        const F dk = d_1;
        const int col_first = col_1;
        const F val_first = val_1;
        load_chunk();                       // row t + W: entries
        load_ptrs(t + 2 * K2_WAVES);        // row t + 2W: pointers
        if (p0 == p1) continue;
        for (int64_t qa = p0; qa < p1; qa += 64) {
            int col = -1;
            F val = F(0);
            if (qa == p0) {
                col = col_first;
                val = val_first;
                if (col >= 0 && col_map) col = col_map[col];
            } else if (qa + lane < p1) {
                const int j = ind[qa + lane];
                col = col_map ? col_map[j] : j;
                val = data[qa + lane];
            }
            const bool inA = col >= i0 && col < i0 + TS;
            const unsigned long long mA = __ballot(inA);
            if (mA == 0) continue;
            const int na = __popcll(mA);
            if (inA) {
                Ent e;
                e.val = val * dk;
                e.col = col - i0;
                sa[__popcll(mA & lt_mask)] = e;
            }
            for (int64_t qb = p0; qb < p1; qb += 64) {
                int colb = col;
                F valb = val;
                if (qb != qa) {
                    colb = -1;
                    valb = F(0);
                    if (qb + lane < p1) {
                        const int j = ind[qb + lane];
                        colb = col_map ? col_map[j] : j;
                        valb = data[qb + lane];
                    }
                }
                const bool inB = colb >= j0 && colb < j0 + TS;
                const unsigned long long mB = __ballot(inB);
                if (mB == 0) continue;
                const int nb = __popcll(mB);
                if (inB) {
                    Ent e;
                    e.val = valb;
                    e.col = colb - j0;
                    sb[__popcll(mB & lt_mask)] = e;
                }
                __builtin_amdgcn_wave_barrier();
                for (int a0 = 0; a0 < na; a0 += 8) {
                    const int a = a0 + la;
                    Ent ea;
                    ea.val = F(0);
                    ea.col = 0;
                    if (a < na) ea = sa[a];
                    for (int b0 = 0; b0 < nb; b0 += 8) {
                        const int b = b0 + lb;
                        if (a < na && b < nb) {
                            const Ent eb = sb[b];
                            if (I != J || eb.col <= ea.col)
                                atomic_add(&tile[ea.col * TS + (eb.col ^ ((ea.col & 15) << 3))],
                                           (lds_acc_t)(ea.val * eb.val));
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    // un-swizzle on the way out (tile column = col ^ ((row & 15) << 3): the 8 lanes that share
    // a B entry hit 8 different LDS banks instead of one)
    F *dst = ws + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (TS * TS);
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) {
        const int r = b / TS, c = b % TS;
        dst[b] = (F)tile[r * TS + (c ^ ((r & 15) << 3))];
    }
}

// ---------------------------------------------------------------------------------------
// K2 (v3)  unrestricted sparse self sandwich on precomputed per-row CHUNK POINTERS:
// The block is stored CHUNK-MAJOR: the entries are regrouped by 128-column chunk, inside a chunk by
// row (column order kept), so chunk c is a CSR matrix of its own whose rows follow each other in
// memory; cptr[c][k] (an [NCH][n + 1] table) is the start of row k in chunk c.  A tile (I, J)
// streams chunk I and chunk J: the lists of the 8 rows of a group are ADJACENT (one coalesced
// ~0.5 KB region per chunk) -- with the row-major CSR they were ~300 B apart and every (row, tile)
// pulled 4 half-used sectors (32 GB of HBM traffic for 3.2 GB of data; loads alone 4.4 ms).
// A wave takes 8 rows at a time, 8 lanes per row: lane t holds slot t of the row's I-list and of
// its J-list.  All 8 x 8 pairs of the 8 rows are formed in registers: step s = 0..7 moves the
// J entry of lane t ^ s to lane t with DPP (quad_perm / row_half_mirror) and issues one
// ds_add_f64 into the LDS tile; slots 8..15 are broadcast inside the row's 8 lanes (quad_perm +
// bank_mask).  No LDS scratch, no ballots, no compaction.
// ---------------------------------------------------------------------------------------
constexpr int K2_ED = 2;   // groups whose entry loads are in flight
constexpr int K2_NP = 2;   // further groups whose chunk-pointer loads are in flight

// slot K (0 .. S - 1) of the lane's own S-lane row group, broadcast to its S lanes
template <int S, int K>
__device__ __forceinline__ int k2_bcast_i32(int v, int x4) {
    if constexpr (S == 8) {
        return dpp_bcast8_i32<K>(v, x4);
    } else if constexpr (S == 4) {
        return __builtin_amdgcn_mov_dpp(v, K * 0x55, 0xF, 0xF, true);                    // quad_perm [K,K,K,K]
    } else {
        return __builtin_amdgcn_mov_dpp(v, K | (K << 2) | ((2 + K) << 4) | ((2 + K) << 6), 0xF, 0xF,
                                        true);                                           // [K,K,2+K,2+K]
    }
}
template <int S, int K>
__device__ __forceinline__ double k2_bcast(double v, double x4) {
    return __hiloint2double(k2_bcast_i32<S, K>(__double2hiint(v), __double2hiint(x4)),
                            k2_bcast_i32<S, K>(__double2loint(v), __double2loint(x4)));
}
template <int S, int K>
__device__ __forceinline__ float k2_bcast(float v, float x4) {
    return __int_as_float(k2_bcast_i32<S, K>(__float_as_int(v), __float_as_int(x4)));
}

// S = slots per row and list half (8, 4 or 2; 64 / S rows per wave step): 8 suits ~6 nonzeros per
// row and 128-column chunk (5 % density); a block with 1-2 nonzeros per row and chunk (wide and
// sparse: 2048 columns at 1.25 %) would fill 3 % of the lanes of its 8 ds_adds per 8 rows -- with
// S = 2 the same pairs take 2 ds_adds per 32 rows.
// U8: `ind` points at BYTES, the column of an entry inside its 128-column chunk (as in sparse_blocks.hip: the entry
// gathers are charged for the bytes they move).
template <typename F, int TS, int S, int NW = K2_WAVES, bool U8 = false>
__global__ __launch_bounds__(NW * 64) void sparse_sandwich_chunked_kernel(
    const F *__restrict__ data, const int32_t *__restrict__ ind,
    const int32_t *__restrict__ cptr, int nch, const F *__restrict__ d, int64_t n,
    int64_t nnz1, int nb_diag, int nb_off, int max_nb, F *__restrict__ ws, int pairs,
    WgLogBuf *__restrict__ wglog) {
    const unsigned long long t_begin = wg_log_begin(wglog);
    // pairs = 1: row-restricted form.  `cptr` is then a table [nch][n][2] of {start, end} of the
    // SELECTED rows (ascending) in every chunk, d the selected weights, n their number: the same
    // pipeline walks a row list at a cost proportional to its length (the reference's
    // `for k in rows`, ext/sparse.pyx:46-48) -- the host gathers the table from the chunk pointers.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    lds_acc_t *tile = reinterpret_cast<lds_acc_t *>(smem_raw);  // [TS][TS] doubles, column-swizzled
    // 1-D grid: the nch diagonal tiles come first with nb_diag workgroups each, then the
    // off-diagonal tiles with nb_off each (a diagonal tile has ~40 % fewer pairs per row).
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
            const int o = (b - ndiag_blocks) / nb_off;       // o-th off-diagonal tile
            blk = (b - ndiag_blocks) % nb_off;
            nblk_part = nb_off;
            int Io = (int)((sqrtf(8.0f * (float)o + 1.0f) - 1.0f) * 0.5f);   // (Io+1, Jo), Jo <= Io
            while ((Io + 1) * (Io + 2) / 2 <= o) ++Io;
            while (Io * (Io + 1) / 2 > o) --Io;
            const int Jo = o - Io * (Io + 1) / 2;
            part = (Io + 1) * (Io + 2) / 2 + Jo;
        }
    }
    const int64_t rows_per_block = (n + nblk_part - 1) / nblk_part;
    int I = (int)((sqrtf(8.0f * (float)part + 1.0f) - 1.0f) * 0.5f);
    while ((I + 1) * (I + 2) / 2 <= part) ++I;
    while (I * (I + 1) / 2 > part) --I;
    const int J = part - I * (I + 1) / 2;
    const int i0 = U8 ? 0 : I * TS, j0 = U8 ? 0 : J * TS;       // (byte columns are chunk-relative)
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int RG = 64 / S;                      // rows per group (wave step)
    const int lr = lane / S, lt = lane % S;         // load phase: row-in-group, entry slot
    const int64_t t0 = (int64_t)blk * rows_per_block;
    const int64_t t1 = min(t0 + rows_per_block, n);
    const int64_t pstride = pairs ? 2 * n : n + 1;   // cptr is [nch][n + 1] (chunk-major twin)

    // A wave owns GROUPS of 8 consecutive rows (8 lanes per row).  Software pipeline over the
    // groups: the entry loads of a group need its chunk pointers (two dependent memory round
    // trips), and one round trip (~3 us under this kernel's load) is longer than the LDS work of a
    // group (~1 us).  So the entries of K2_ED groups and the pointers of K2_NP further groups
    // are in flight while group g is processed: per-group time = max(work, latency / K2_ED)
    // instead of the latency (PMC before: 62 % of the wave cycles in s_waitcnt, diagonal tiles as
    // slow as off-diagonal ones although they issue half the atomics).
    // Every load below is issued unconditionally with a clamped address (no exec-masked
    // branches, no arithmetic on a loaded value before the next load goes out): the number of
    // outstanding loads per iteration is then a compile-time constant and the s_waitcnt for a
    // value loaded two iterations ago leaves everything younger in flight.
    // All indices are 32-bit offsets relative to the workgroup's row range [t0, t1) and to the
    // first entry of that range in chunk I / chunk J (uniform bases in SGPRs): the clamps are single
    // v_min_u32 and the loads use the SGPR-base + 32-bit-offset form instead of 64-bit arithmetic.
    struct Ptr { int a0, a1, b0, b1; F d; bool valid; };
    struct Grp { int pA, nA, pB, nB; F d; F va, vb, va2, vb2; int ca, cb, ca2, cb2; };
    const int nrows = (int)(t1 - t0);                       // < 2^31 (host: rows_per_block)
    const int kmax = (int)max((int64_t)0, min(n - 1 - t0, (int64_t)0x7fffffff));
    const int32_t *cpA = cptr + (int64_t)I * pstride + (t0 << pairs);
    const int32_t *cpB = cptr + (int64_t)J * pstride + (t0 << pairs);
    const int last = pairs ? 2 * nrows - 1 : nrows;          // where the range's last row ends
    const F *dW = d + t0;
    const int baseA = __builtin_amdgcn_readfirstlane(cpA[0]);
    const int baseB = __builtin_amdgcn_readfirstlane(cpB[0]);
    const unsigned spanA1 = (unsigned)max(__builtin_amdgcn_readfirstlane(cpA[last]) - baseA - 1, 0);
    const unsigned spanB1 = (unsigned)max(__builtin_amdgcn_readfirstlane(cpB[last]) - baseB - 1, 0);
    const F *dataA = data + min((int64_t)baseA, nnz1), *dataB = data + min((int64_t)baseB, nnz1);
    const int32_t *indA = ind + min((int64_t)baseA, nnz1), *indB = ind + min((int64_t)baseB, nnz1);
    const unsigned char *ind8A = reinterpret_cast<const unsigned char *>(ind) + min((int64_t)baseA, nnz1);
    const unsigned char *ind8B = reinterpret_cast<const unsigned char *>(ind) + min((int64_t)baseB, nnz1);
    // The whole pipeline is instantiated twice: DIAG (I == J: one list per row, loaded once, pairs
    // b <= a) and off-diagonal (two lists); the choice is workgroup-uniform and made once, outside
    // the load pipeline (a branch inside it would bring the conservative s_waitcnt back).
    auto run_tile = [&](auto diag_c) {
    constexpr bool DIAG = decltype(diag_c)::value;
    auto load_ptrs = [&](int g) {              // g: first row of the group, relative to t0
        Ptr q;
        const int k = g + lr;
        q.valid = k < nrows;
        const unsigned kc = (unsigned)min(k, kmax);
        const unsigned kb = kc << (2 + pairs);
        q.d = *reinterpret_cast<const F *>(reinterpret_cast<const char *>(dW) + kc * (unsigned)sizeof(F));
        q.a0 = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(cpA) + kb);
        q.a1 = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(cpA) + kb + 4);
        if constexpr (!DIAG) {      // a diagonal tile pairs the I-list with itself: loaded once
            q.b0 = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(cpB) + kb);
            q.b1 = *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(cpB) + kb + 4);
        } else {
            q.b0 = q.b1 = 0;
        }
        return q;
    };
    auto load_entries = [&](const Ptr &q) {   // slots lt and lt + S of both lists: 2 S entries per list
        Grp e;
        const bool on = q.valid && q.d != F(0);      // rows with d == 0 contribute nothing
        e.d = q.d;
        e.pA = q.a0;
        e.pB = q.b0;
        e.nA = on ? q.a1 - q.a0 : 0;
        e.nB = DIAG ? 0 : (on ? q.b1 - q.b0 : 0);
        const unsigned rA = (unsigned)(q.a0 - baseA), rB = (unsigned)(q.b0 - baseB);
        const unsigned lA = (unsigned)max(e.nA - 1, 0), lB = (unsigned)max(e.nB - 1, 0);
        const unsigned iA = min(rA + min((unsigned)lt, lA), spanA1);
        const unsigned iA2 = min(rA + min((unsigned)lt + (unsigned)S, lA), spanA1);
        // 32-bit BYTE offsets (host: < 2^32 per workgroup range) -> SGPR base + VGPR offset loads
        auto ldi = [](const int32_t *base, unsigned i) {
            return *reinterpret_cast<const int32_t *>(reinterpret_cast<const char *>(base) + (i << 2));
        };
        auto ldf = [](const F *base, unsigned i) {
            return *reinterpret_cast<const F *>(reinterpret_cast<const char *>(base) +
                                                (i * (unsigned)sizeof(F)));
        };
        e.ca = U8 ? (int)ind8A[iA] : ldi(indA, iA);
        e.va = ldf(dataA, iA);
        e.ca2 = U8 ? (int)ind8A[iA2] : ldi(indA, iA2);
        e.va2 = ldf(dataA, iA2);
        if constexpr (!DIAG) {
            const unsigned iB = min(rB + min((unsigned)lt, lB), spanB1);
            const unsigned iB2 = min(rB + min((unsigned)lt + (unsigned)S, lB), spanB1);
            e.cb = U8 ? (int)ind8B[iB] : ldi(indB, iB);
            e.vb = ldf(dataB, iB);
            e.cb2 = U8 ? (int)ind8B[iB2] : ldi(indB, iB2);
            e.vb2 = ldf(dataB, iB2);
        } else {
            // the B side of a diagonal tile is read from the A fields in process() (no copies
            // of values that are still in flight)
            e.cb = e.cb2 = 0;
            e.vb = e.vb2 = F(0);
        }
        return e;
    };
    const int gstep = NW * RG;
    const int gw = wave * RG;
    Grp ea[K2_ED], eb[K2_ED];      // two register sets: the loop is unrolled by two turns
    Ptr ps[K2_NP];
#pragma unroll
    for (int i = 0; i < K2_ED; ++i) ea[i] = load_entries(load_ptrs(gw + i * gstep));
#pragma unroll
    for (int i = 0; i < K2_NP; ++i) ps[i] = load_ptrs(gw + (K2_ED + i) * gstep);
    const int pa = lane >> 3, pb = lane & 7;       // (a, b) of an 8 x 8 block (long-list fallback)
    static_assert(K2_ED == 2 && K2_NP == 2, "two groups per turn");
    // Pair keys, pre-scaled to byte offsets of the tile (<< SH):
    //   B entry: kb = col << SH, or BIGKEY for a slot beyond its list;
    //   A entry: limit la = (col << SH) | offmask (-8 for an empty slot) and base
    //            ba = row base | column swizzle, so that the target of (a, b) is tile + (kb ^ ba)
    //            (the swizzle bits and the row-base bits are disjoint from each other and the row
    //            base from kb) and the pair is wanted iff kb <= la -- one signed compare covering
    //            both validities and, on diagonal tiles (offmask 0), the b <= a triangle.
    constexpr int SH = 3;                          // byte offsets into a tile of doubles
    constexpr int BIGKEY = 0x7ffffff0;
    constexpr int offmask = DIAG ? 0 : 0x70000000;
    char *const tile_bytes = reinterpret_cast<char *>(tile);
    auto a_lim = [&](int col) { return (col < 0 ? -8 : col << SH) | offmask; };
    auto a_base = [&](int col) { return (int)(((unsigned)col << (7 + SH)) | ((col & 15) << (3 + SH))); };
    auto b_key = [&](int col) { return col < 0 ? BIGKEY : col << SH; };
    auto add_pair = [&](int kb, int la, int ba, F prod) {
        if (kb <= la)
            atomic_add(reinterpret_cast<lds_acc_t *>(tile_bytes + (unsigned)(kb ^ ba)), (lds_acc_t)prod);
    };
    auto process = [&](const Grp &cur) {
        const int nA = cur.nA, nB = DIAG ? cur.nA : cur.nB, pA0 = cur.pA, pB0 = DIAG ? cur.pA : cur.pB;
        const F dk = cur.d;
        if (!__any(nA > 0 && nB > 0)) return;       // no row of the group has a pair in this tile
        // first halves (slots 0..7): every lane pairs its own A entry with the B entry of lane
        // (lane ^ s), s = 0..7, fetched with DPP moves inside the 8 lanes of its row -- 8 x 64
        // pairs = all 8 x 8 combinations of the 8 rows, no LDS scratch traffic
        const int colA = lt < nA ? cur.ca - i0 : -1;
        const int colB = DIAG ? colA : (lt < nB ? cur.cb - j0 : -1);
        const int la = a_lim(colA), ba = a_base(colA), kb = b_key(colB);
        const F av = cur.va * dk, vb = DIAG ? cur.va : cur.vb;
        add_pair(kb, la, ba, av * vb);
        if constexpr (S >= 2) add_pair(dpp_xor_i32<1>(kb), la, ba, av * dpp_xor<1>(vb));
        if constexpr (S >= 4) {
            add_pair(dpp_xor_i32<2>(kb), la, ba, av * dpp_xor<2>(vb));
            add_pair(dpp_xor_i32<3>(kb), la, ba, av * dpp_xor<3>(vb));
        }
        if constexpr (S == 8) {
            const int kb4 = dpp_xor_i32<4>(kb);
            const F vb4 = dpp_xor<4>(vb);
            add_pair(kb4, la, ba, av * vb4);
            add_pair(dpp_xor_i32<1>(kb4), la, ba, av * dpp_xor<1>(vb4));
            add_pair(dpp_xor_i32<2>(kb4), la, ba, av * dpp_xor<2>(vb4));
            add_pair(dpp_xor_i32<3>(kb4), la, ba, av * dpp_xor<3>(vb4));
        }
        // second halves (slots S .. 2 S - 1; at S = 8 and 5 % density ~19 % of the rows have one):
        // slot S + k of a row is broadcast to the row's S lanes, k = 0 .. (longest overhang of the
        // group's rows) - 1
        const bool anyA2 = __any(nA > S), anyB2 = __any(nB > S);
        const int colB2 = lt + S < nB ? (DIAG ? cur.ca2 : cur.cb2) - j0 : -1;
        const int kb2 = b_key(colB2);
        const F vb2 = DIAG ? cur.va2 : cur.vb2;
        if (anyA2) {
            // A overhang x (B first half, B overhang)
            const int colA2 = lt + S < nA ? cur.ca2 - i0 : -1;
            const int la2 = a_lim(colA2), ba2 = a_base(colA2);
            const F av2 = cur.va2 * dk;
            // (the S = 8 broadcast needs the value of the other quad as a second operand)
            const int la2x = S == 8 ? dpp_xor_i32<4>(la2) : 0, ba2x = S == 8 ? dpp_xor_i32<4>(ba2) : 0;
            const F av2x = S == 8 ? dpp_xor<4>(av2) : F(0);
            bool go = true;
            static_for<S>([&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if (go) {
                    if (!__any(nA > S + K)) {
                        go = false;
                    } else {
                        const int lk = k2_bcast_i32<S, K>(la2, la2x), bk = k2_bcast_i32<S, K>(ba2, ba2x);
                        const F ak = k2_bcast<S, K>(av2, av2x);
                        add_pair(kb, lk, bk, ak * vb);
                        if (anyB2) add_pair(kb2, lk, bk, ak * vb2);
                    }
                }
            });
        }
        if (!DIAG && anyB2) {
            // A first half x B overhang (empty on diagonal tiles: those columns are all > a's)
            const int kb2x = S == 8 ? dpp_xor_i32<4>(kb2) : 0;
            const F vb2x = S == 8 ? dpp_xor<4>(vb2) : F(0);
            bool go = true;
            static_for<S>([&](auto kc) {
                constexpr int K = decltype(kc)::value;
                if (go) {
                    if (!__any(nB > S + K)) {
                        go = false;
                    } else {
                        add_pair(k2_bcast_i32<S, K>(kb2, kb2x), la, ba, av * k2_bcast<S, K>(vb2, vb2x));
                    }
                }
            });
        }
        if (__any(nA > 2 * S) || __any(nB > 2 * S)) {
            // very long lists (> 2 S entries of one row in one 128-column chunk): remaining
            // 8 x 8 blocks straight from the CSR arrays
            for (int r = 0; r < RG; ++r) {
                const int nAr = __builtin_amdgcn_readlane(nA, r * S);
                const int nBr = __builtin_amdgcn_readlane(nB, r * S);
                if ((nAr <= 2 * S && nBr <= 2 * S) || nAr == 0 || nBr == 0) continue;
                const int pAr = __builtin_amdgcn_readlane(pA0, r * S);
                const int pBr = __builtin_amdgcn_readlane(pB0, r * S);
                const F dr = readlane_f<F>(dk, r * S);
                for (int a0 = 0; a0 < nAr; a0 += 8) {
                    const int a = a0 + pa;
                    int ca = 0;
                    F va = F(0);
                    if (a < nAr) {
                        ca = (U8 ? (int)reinterpret_cast<const unsigned char *>(ind)[pAr + a] : ind[pAr + a]) - i0;
                        va = data[pAr + a] * dr;
                    }
                    // (pairs with a < 2 S and b < 2 S were formed above)
                    for (int b0 = (a0 + 8 <= 2 * S ? (2 * S) / 8 * 8 : 0); b0 < nBr; b0 += 8) {
                        const int b = b0 + pb;
                        if (a < nAr && b < nBr && (a >= 2 * S || b >= 2 * S)) {
                            const int cb = (U8 ? (int)reinterpret_cast<const unsigned char *>(ind)[pBr + b] : ind[pBr + b]) - j0;
                            const F vb = data[pBr + b];
                            if (!DIAG || cb <= ca)
                                atomic_add(&tile[ca * TS + (cb ^ ((ca & 15) << 3))], (lds_acc_t)(va * vb));
                        }
                    }
                }
            }
        }
    };
    // Two groups per turn: the loads of the next turn (entries of g + 2, g + 3 from the pointers
    // fetched last turn; pointers of g + 4, g + 5) are issued back to back, then the two groups of
    // this turn are processed -- every load has two groups of work to land before it is touched.
    // Two turns per iteration with alternating entry registers: no register copies of values that
    // are still in flight (a copy would wait for the load).
    for (int g = gw; g < nrows; g += 4 * gstep) {
        eb[0] = load_entries(ps[0]);
        eb[1] = load_entries(ps[1]);
        ps[0] = load_ptrs(g + 4 * gstep);
        ps[1] = load_ptrs(g + 5 * gstep);
        process(ea[0]);
        if (g + gstep < nrows) process(ea[1]);
        if (g + 2 * gstep >= nrows) break;
        ea[0] = load_entries(ps[0]);
        ea[1] = load_entries(ps[1]);
        ps[0] = load_ptrs(g + 6 * gstep);
        ps[1] = load_ptrs(g + 7 * gstep);
        process(eb[0]);
        if (g + 3 * gstep < nrows) process(eb[1]);
    }
    };
    if (I == J) run_tile(std::true_type{});
    else run_tile(std::false_type{});
    __syncthreads();
    if (threadIdx.x == 0) wg_log_end(wglog, t_begin, WG_K2);
    F *dst = ws + ((int64_t)part * max_nb + blk) * (TS * TS);
    for (int b = threadIdx.x; b < TS * TS; b += blockDim.x) {
        const int r = b / TS, c = b % TS;
        dst[b] = (F)tile[r * TS + (c ^ ((r & 15) << 3))];
    }
}

// out[i][j] (n_out x n_out) from the reduced tile buffer [part][TS*TS]; mirror included.
template <typename F, int TS>
__global__ void sparse_sandwich_assemble_kernel(const F *__restrict__ tiles, int n_out, int nchunk,
                                                F *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (i >= n_out || j >= n_out) return;
    const int hi = max(i, j), lo = min(i, j);
    const int I = hi / TS, J = lo / TS;
    const int part = I * (I + 1) / 2 + J;
    (void)nchunk;
    out[(int64_t)i * n_out + j] = tiles[(int64_t)part * TS * TS + (hi % TS) * TS + (lo % TS)];
}

template <typename F>
__global__ void untile_kernel(const F *__restrict__ tmp, int64_t nA, int64_t nB, int TB,
                              F *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= nA * nB) return;
    const int64_t a = e / nB, j = e % nB;
    out[e] = tmp[((j / TB) * nA + a) * TB + (j % TB)];
}

// ---------------------------------------------------------------------------------------
// host drivers
// ---------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

// the stream kernels fetch PAIRS of entries: element-aligned arrays whose first entries share their parity (any
// row-sliced view of arrays that were allocated together does)
template <typename F>
static inline bool csr_stream_aligned(const F *data, const int32_t *ind) {
    const uintptr_t d = reinterpret_cast<uintptr_t>(data), i = reinterpret_cast<uintptr_t>(ind);
    return d % sizeof(F) == 0 && i % 4 == 0 && ((d / sizeof(F)) & 1) == ((i / 4) & 1);
}

template <typename F>
static int run_csr_matvec_u16(const F *data, const uint16_t *ind16, const int64_t *ptr, int64_t n, int64_t m,
                              const F *v, F *out, hipStream_t st) {
    if (n == 0 || m == 0) return TM_OK;
    TM_REQUIRE(m <= 65536 && sizeof(F) * (size_t)(m + CSR_WAVES * CSR_CAP) <= 64 * 1024,
               "csr_matvec (16-bit columns): the coefficient vector must fit the LDS beside the staging buffers");
    const size_t lds = sizeof(F) * (size_t)(m + CSR_WAVES * CSR_CAP);
    auto kern = &csr_matvec_stream_kernel<F, uint16_t>;
    if (lds > 48 * 1024)
        TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nchunk = ceil_div(n, CSR_RPW);
    const int64_t nblk = std::min<int64_t>(ceil_div(nchunk, CSR_WAVES), NUM_CU * 4);
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(CSR_WAVES * 64), lds, st, data, ind16, ptr, v, n, (int)m, out);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_csr_matvec(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n,
                          int64_t m, const F *v, const int32_t *rows, int64_t n_rows,
                          const int32_t *cols, int64_t n_cols, F *out, hipStream_t st) {
    const int64_t n_iter = rows ? n_rows : n;
    if (n_iter == 0 || m == 0) return TM_OK;
    if (cols && n_cols == 0) return TM_OK;
    if (!rows && !cols && sizeof(F) * (size_t)(m + CSR_WAVES * CSR_CAP) <= 64 * 1024) {
        const size_t lds = sizeof(F) * (size_t)(m + CSR_WAVES * CSR_CAP);
        auto kern = &csr_matvec_stream_kernel<F, int32_t>;
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int64_t nchunk = ceil_div(n, CSR_RPW);
        const int64_t nblk = std::min<int64_t>(ceil_div(nchunk, CSR_WAVES), NUM_CU * 4);
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(CSR_WAVES * 64), lds, st, data, ind, ptr,
                           v, n, (int)m, out);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return TM_OK;
    }
    int32_t *col_map = nullptr;
    if (cols) {
        void *wsv = nullptr;
        int rc = get_workspace(align256(sizeof(int32_t) * (size_t)m), &wsv, st);
        if (rc) return rc;
        col_map = reinterpret_cast<int32_t *>(wsv);
        rc = build_col_map(col_map, m, cols, n_cols, st);
        if (rc) return rc;
    }
    constexpr int G = 16;
    const int64_t nblk = std::min<int64_t>(ceil_div(n_iter * G, 256), NUM_CU * 16);
    prof_begin(st);
    hipLaunchKernelGGL((csr_matvec_kernel<F, G>), dim3((unsigned)nblk), dim3(256), 0, st, data, ind,
                       ptr, v, rows, n_iter, col_map, out);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_csr_rmatvec_u16(const F *data, const uint16_t *ind16, const int64_t *ptr, int64_t n, int64_t m,
                               const F *v, F *out, hipStream_t st, int square) {
    if (n == 0 || m == 0) return TM_OK;
    const uintptr_t dpa = reinterpret_cast<uintptr_t>(data), ipa = reinterpret_cast<uintptr_t>(ind16);
    TM_REQUIRE(m <= 65536 && sizeof(lds_acc_t) * (size_t)(m + 1) + sizeof(F) * (size_t)(CSR_WAVES * CSR_CAP) <= SP_LDS_MAX,
               "csr_rmatvec (16-bit columns): the accumulators must fit the LDS");
    TM_REQUIRE(dpa % sizeof(F) == 0 && ipa % 2 == 0 && ((dpa / sizeof(F)) & 1) == ((ipa / 2) & 1),
               "csr_rmatvec (16-bit columns): values and columns must start at entries of the same parity");
    const size_t lds = sizeof(lds_acc_t) * (size_t)((m + 1) & ~(int64_t)1) + sizeof(F) * (size_t)(CSR_WAVES * CSR_CAP);
    auto kern = &csr_rmatvec_stream_kernel<F, uint16_t>;
    if (lds > 48 * 1024)
        TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t nchunk = ceil_div(n, CSR_RPW);
    const int64_t nwave = std::min<int64_t>(nchunk, (int64_t)NUM_CU * 4 * CSR_WAVES);
    const int64_t cpw = ceil_div(nchunk, nwave);
    const int64_t nblk = ceil_div(ceil_div(nchunk, cpw), CSR_WAVES);
    void *wsv = nullptr;
    int rc = get_workspace(sizeof(F) * (size_t)(nblk * m) + 256, &wsv, st);
    if (rc) return rc;
    F *ws = reinterpret_cast<F *>(wsv);
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(CSR_WAVES * 64), lds, st, data, ind16, ptr, v, n, (int)m, cpw,
                       ws, square);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return launch_reduce_partials<F>(ws, m, (int)nblk, 1, out, m, true, st);
}

template <typename F>
static int run_csr_rmatvec(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n,
                           int64_t m, const F *v, const int32_t *rows, int64_t n_rows,
                           const int32_t *cols, int64_t n_cols, F *out, hipStream_t st,
                           int square = 0) {
    const int64_t n_iter = rows ? n_rows : n;
    const int64_t n_out = cols ? n_cols : m;
    if (n_iter == 0 || n_out == 0) return TM_OK;
    if (!rows && !cols &&
        sizeof(lds_acc_t) * (size_t)(m + 1) + sizeof(F) * (size_t)(CSR_WAVES * CSR_CAP) <= SP_LDS_MAX &&
        csr_stream_aligned(data, ind)) {
        const size_t lds = sizeof(lds_acc_t) * (size_t)((m + 1) & ~(int64_t)1) + sizeof(F) * (size_t)(CSR_WAVES * CSR_CAP);
        auto kern = &csr_rmatvec_stream_kernel<F, int32_t>;
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int64_t nchunk = ceil_div(n, CSR_RPW);
        const int64_t nwave = std::min<int64_t>(nchunk, (int64_t)NUM_CU * 4 * CSR_WAVES);
        const int64_t cpw = ceil_div(nchunk, nwave);
        const int64_t nblk = ceil_div(ceil_div(nchunk, cpw), CSR_WAVES);
        void *wsv = nullptr;
        int rc = get_workspace(sizeof(F) * (size_t)(nblk * m) + 256, &wsv, st);
        if (rc) return rc;
        F *ws = reinterpret_cast<F *>(wsv);
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(CSR_WAVES * 64), lds, st, data, ind, ptr,
                           v, n, (int)m, cpw, ws, square);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return launch_reduce_partials<F>(ws, m, (int)nblk, 1, out, m, true, st);
    }
    const size_t map_bytes = cols ? align256(sizeof(int32_t) * (size_t)m) : 0;
    const bool use_lds = sizeof(lds_acc_t) * (size_t)n_out <= SP_LDS_MAX;
    // (a short, wide block -- the reference's 'sparse_wide' design is 40k x 10k -- still gets a workgroup per
    // 256 rows: 20 workgroups of 2048 rows ran at 0.09 TB/s)
    int64_t nblk = std::min<int64_t>(std::max<int64_t>(1, ceil_div(n_iter, n_out > 4096 ? 256 : 2048)), NUM_CU * 2);
    const int64_t rpb = ceil_div(n_iter, nblk);
    nblk = ceil_div(n_iter, rpb);
    void *wsv = nullptr;
    int rc = get_workspace(map_bytes + (use_lds ? sizeof(F) * (size_t)(nblk * n_out) : 0) + 256,
                           &wsv, st);
    if (rc) return rc;
    int32_t *col_map = nullptr;
    if (cols) {
        col_map = reinterpret_cast<int32_t *>(wsv);
        rc = build_col_map(col_map, m, cols, n_cols, st);
        if (rc) return rc;
    }
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + map_bytes);
    constexpr int G = 16;
    if (use_lds) {
        const size_t lds = sizeof(lds_acc_t) * (size_t)n_out;
        auto kern = &csr_rmatvec_kernel<F, G, true>;
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, data, ind, ptr, v, rows,
                           n_iter, rpb, col_map, (int)n_out, ws, out, square);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return launch_reduce_partials<F>(ws, n_out, (int)nblk, 1, out, n_out, true, st);
    }
    hipLaunchKernelGGL((csr_rmatvec_kernel<F, G, false>), dim3((unsigned)nblk), dim3(256), 0, st,
                       data, ind, ptr, v, rows, n_iter, rpb, col_map, (int)n_out, ws, out, square);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_csr_dense(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n,
                         int64_t m, const F *B, int64_t r, int order_f, const F *d,
                         const int32_t *rows, int64_t n_rows, const int32_t *A_cols, int64_t nA_in,
                         const int32_t *B_cols, int64_t nB_in, F *out, hipStream_t st) {
    const int64_t nA = A_cols ? nA_in : m;
    const int64_t nB = B_cols ? nB_in : r;
    const int64_t n_iter = rows ? n_rows : n;
    const int64_t total = nA * nB;
    if (total == 0) return TM_OK;
    TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
    if (n_iter == 0) return TM_OK;
    // TB: B-columns per part (power of two <= 64) such that nA x TB fits the LDS budget
    int TB = 64;
    while (TB > 1 && sizeof(lds_acc_t) * (size_t)nA * TB > SP_LDS_MAX) TB >>= 1;
    while (TB > 1 && TB / 2 >= nB) TB >>= 1;
    if (sizeof(lds_acc_t) * (size_t)nA * TB > SP_LDS_MAX) {
        set_error("csr_dense_sandwich: %lld selected sparse columns exceed the LDS tile",
                  (long long)nA);
        return TM_EUNSUPPORTED;
    }
    const int64_t n_parts = ceil_div(nB, TB);
    const int64_t stride = nA * TB;
    const size_t lds = sizeof(lds_acc_t) * (size_t)stride;
    const int blocks_per_cu = lds > 64 * 1024 ? 1 : 2;
    int64_t nblk = std::max<int64_t>(1, (NUM_CU * blocks_per_cu) / n_parts);
    nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, ceil_div(n_iter, 512)));
    const int64_t rpb = ceil_div(n_iter, nblk);
    nblk = ceil_div(n_iter, rpb);
    const size_t map_bytes = A_cols ? align256(sizeof(int32_t) * (size_t)m) : 0;
    const size_t tmp_bytes = align256(sizeof(F) * (size_t)(n_parts * stride));
    void *wsv = nullptr;
    int rc = get_workspace(map_bytes + tmp_bytes + sizeof(F) * (size_t)(n_parts * nblk * stride) + 256,
                           &wsv, st);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(wsv);
    int32_t *a_map = nullptr;
    if (A_cols) {
        a_map = reinterpret_cast<int32_t *>(base);
        rc = build_col_map(a_map, m, A_cols, nA, st);
        if (rc) return rc;
    }
    F *tmp = reinterpret_cast<F *>(base + map_bytes);       // [n_parts][nA][TB]
    F *ws = reinterpret_cast<F *>(base + map_bytes + tmp_bytes);

    auto go = [&](auto kern) -> int {
        if (lds > 48 * 1024)
            TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        prof_begin(st);
        hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts), dim3(256), lds, st, data,
                           ind, ptr, B, n, r, d, rows, n_iter, rpb, a_map, (int)nA, B_cols, (int)nB,
                           ws, stride);
        prof_end(st);
        TM_LAUNCH_CHECK();
        return TM_OK;
    };
#define TM_CSRD_CASE(TBV)                                                   \
    case TBV:                                                               \
        rc = order_f ? go(&csr_dense_kernel<F, TBV, true>) : go(&csr_dense_kernel<F, TBV, false>); \
        break;
    switch (TB) {
        TM_CSRD_CASE(64)
        TM_CSRD_CASE(32)
        TM_CSRD_CASE(16)
        TM_CSRD_CASE(8)
        TM_CSRD_CASE(4)
        TM_CSRD_CASE(2)
        TM_CSRD_CASE(1)
        default:
            rc = TM_EINVAL;
    }
#undef TM_CSRD_CASE
    if (rc) return rc;
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, (int)n_parts, tmp, n_parts * stride,
                                   false, st);
    if (rc) return rc;
    // tmp is [part][nA][TB] -> out[nA][nB]
    hipLaunchKernelGGL((untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, tmp,
                       nA, nB, TB, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_sparse_sandwich(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n,
                               int64_t m, const F *d, const int32_t *rows, int64_t n_rows,
                               const int32_t *cols, int64_t n_cols, F *out, hipStream_t st) {
    const int64_t n_out = cols ? n_cols : m;
    const int64_t n_iter = rows ? n_rows : n;
    if (n_out == 0) return TM_OK;
    if (n_iter == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)(n_out * n_out), st));
        return TM_OK;
    }
    constexpr int TS = 128;   // 128 KB tile of doubles + pair scratch
    const int nchunk = (int)ceil_div(n_out, TS);
    const int n_parts = nchunk * (nchunk + 1) / 2;
    TM_REQUIRE(n_parts <= 65535, "too many sparse columns for the tiled sandwich");
    const size_t lds = sizeof(lds_acc_t) * (size_t)(TS * TS) + sizeof(K2Entry<F>) * K2_WAVES * 2 * 64;
    int64_t nblk = std::max<int64_t>(1, NUM_CU / n_parts);
    nblk = std::min<int64_t>(nblk, std::max<int64_t>(1, ceil_div(n_iter, 256)));
    const int64_t rpb = ceil_div(n_iter, nblk);
    nblk = ceil_div(n_iter, rpb);
    const size_t map_bytes = cols ? align256(sizeof(int32_t) * (size_t)m) : 0;
    const size_t ij_bytes = 0;
    const size_t tmp_bytes = align256(sizeof(F) * (size_t)n_parts * TS * TS);
    void *wsv = nullptr;
    int rc = get_workspace(map_bytes + ij_bytes + tmp_bytes +
                               sizeof(F) * (size_t)n_parts * (size_t)nblk * TS * TS + 256,
                           &wsv, st);
    if (rc) return rc;
    char *base = reinterpret_cast<char *>(wsv);
    int32_t *col_map = nullptr;
    if (cols) {
        col_map = reinterpret_cast<int32_t *>(base);
        rc = build_col_map(col_map, m, cols, n_cols, st);
        if (rc) return rc;
    }
    F *tmp = reinterpret_cast<F *>(base + map_bytes + ij_bytes);
    F *ws = reinterpret_cast<F *>(base + map_bytes + ij_bytes + tmp_bytes);
    auto kern = &sparse_sandwich_kernel<F, TS>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts), dim3(K2_WAVES * 64), lds, st,
                       data, ind, ptr, d, rows, n_iter, rpb, col_map, (int)n_out, ws);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, (int64_t)TS * TS, (int)nblk, n_parts, tmp,
                                   (int64_t)n_parts * TS * TS, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((sparse_sandwich_assemble_kernel<F, TS>),
                       dim3((unsigned)ceil_div(n_out, 64), (unsigned)n_out), dim3(64), 0, st, tmp,
                       (int)n_out, nchunk, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F, bool U8 = false>
static int run_sparse_sandwich_chunked(const F *data, const int32_t *ind, const int32_t *cptr,
                                       int64_t n, int64_t m, int64_t nnz, const F *d, F *out,
                                       hipStream_t st, int pairs = 0) {
    if (m == 0) return TM_OK;
    if (n == 0 || nnz == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)(m * m), st));
        return TM_OK;
    }
    constexpr int TS = 128;
    const int nchunk = (int)ceil_div(m, TS);
    const int n_parts = nchunk * (nchunk + 1) / 2;
    TM_REQUIRE(n_parts <= 65535, "too many sparse columns for the tiled sandwich");
    const size_t lds = sizeof(lds_acc_t) * (size_t)(TS * TS);   // the tile only: pairs are formed in registers
    // workgroups per tile.  A diagonal tile (one list per row, pairs b <= a, no B-overhang phase)
    // costs ~0.75 of an off-diagonal one per row (measured: 22 / 28 workgroups per tile is the
    // optimum at 512 columns, 4.74 ms against 5.10 ms for 24 / 26): split the CUs by that weight.
    const int n_off = n_parts - nchunk;
    int nb_off = n_off > 0 ? std::max(1, (int)(NUM_CU / (n_off + 0.75 * nchunk))) : 0;
    int nb_diag = std::max(1, (NUM_CU - n_off * nb_off) / nchunk);
    if (n_off == 0) nb_off = nb_diag;
    const int cap = (int)std::max<int64_t>(1, ceil_div(n, 1024));      // small n: fewer workgroups
    nb_diag = std::min(nb_diag, cap);
    nb_off = std::min(nb_off, cap);
    // (byte offsets: rows per workgroup * 128 entries * 8 bytes must stay below 2^32)
    {
        const int min_nb = (int)ceil_div(n, (int64_t)1 << 21);
        nb_diag = std::max(nb_diag, min_nb);
        nb_off = std::max(nb_off, min_nb);
    }
    TM_REQUIRE(nnz < (1ll << 31), "sparse block too large for the tiled sandwich (nnz >= 2^31)");
    const int64_t nblk = std::max(nb_diag, nb_off);   // stride of the partial-tile buffer
    const size_t tmp_bytes = align256(sizeof(F) * (size_t)n_parts * TS * TS);
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)n_parts * (size_t)nblk * TS * TS + 256,
                           &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    // slots per row and list half by the mean number of nonzeros per row and 128-column chunk
    const double per_chunk = (double)nnz / ((double)n * nchunk);
    const int force_s = (int)tune("k2_slots", 0);      // (experiments: 8 / 4 / 2 slots per row and chunk; 0 = by density)
    // measured at 2M rows (profiles/r2_microbench.txt): 8 slots win above ~4.5 nonzeros per row and
    // chunk, 2 slots below ~0.9
    const int slots = force_s ? force_s : (per_chunk > 4.5 ? 8 : per_chunk > 0.9 ? 4 : 2);
    // waves per workgroup: 16 when the kernel has the CU to itself; 8 / 12 leave registers for a
    // co-resident MFMA kernel on the same CU (tm_tune_set("k2_waves", ...), S = 8 only)
    const int nw = slots == 8 ? (int)tune("k2_waves", K2_WAVES) : K2_WAVES;
    auto kern = slots == 8   ? (nw == 8    ? &sparse_sandwich_chunked_kernel<F, TS, 8, 8, U8>
                                : nw == 12 ? &sparse_sandwich_chunked_kernel<F, TS, 8, 12, U8>
                                           : &sparse_sandwich_chunked_kernel<F, TS, 8, K2_WAVES, U8>)
                : slots == 4 ? &sparse_sandwich_chunked_kernel<F, TS, 4, K2_WAVES, U8>
                             : &sparse_sandwich_chunked_kernel<F, TS, 2, K2_WAVES, U8>;
    const int threads = (slots == 8 && (nw == 8 || nw == 12) ? nw : K2_WAVES) * 64;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // every tile's partials are reduced over nblk slots: tiles with fewer workgroups leave theirs 0
    if (nb_diag != nb_off)
        TM_HIP(hipMemsetAsync(ws, 0, sizeof(F) * (size_t)n_parts * (size_t)nblk * TS * TS, st));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nchunk * nb_diag + n_off * nb_off)),
                       dim3(threads), lds, st, data, ind, cptr, nchunk, d, n, nnz - 1, nb_diag,
                       nb_off, (int)nblk, ws, pairs,
                       wg_log_ptr());
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

using namespace tmh;

extern "C" {

#define TM_CSR_MV_ENTRY(NAME, F, RUN)                                                           \
    int NAME(const F *data, const int32_t *ind, const int64_t *ptr, int64_t n, int64_t m,       \
             const F *v, const int32_t *rows, int64_t n_rows, const int32_t *cols,              \
             int64_t n_cols, F *out, void *stream) {                                            \
        TM_REQUIRE(n >= 0 && m >= 0, "negative shape");                                         \
        return RUN<F>(data, ind, ptr, n, m, v, rows, n_rows, cols, n_cols, out,                 \
                      as_stream(stream));                                                       \
    }

// out[i] += sum_k data[k] * v[ind16[k]] over row i (all rows, all columns), the column indices as uint16
int tm_csr_matvec_u16_f32(const float *data, const uint16_t *ind16, const int64_t *ptr, int64_t n, int64_t m,
                          const float *v, float *out, void *stream) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    return run_csr_matvec_u16<float>(data, ind16, ptr, n, m, v, out, as_stream(stream));
}
int tm_csr_matvec_u16_f64(const double *data, const uint16_t *ind16, const int64_t *ptr, int64_t n, int64_t m,
                          const double *v, double *out, void *stream) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    return run_csr_matvec_u16<double>(data, ind16, ptr, n, m, v, out, as_stream(stream));
}

// out[j] += sum_i v[i] * X[i, j] (all rows, all columns) on the 16-bit column twin
int tm_csr_rmatvec_u16_f32(const float *data, const uint16_t *ind16, const int64_t *ptr, int64_t n, int64_t m,
                           const float *v, float *out, void *stream) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    return run_csr_rmatvec_u16<float>(data, ind16, ptr, n, m, v, out, as_stream(stream), 0);
}
int tm_csr_rmatvec_u16_f64(const double *data, const uint16_t *ind16, const int64_t *ptr, int64_t n, int64_t m,
                           const double *v, double *out, void *stream) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    return run_csr_rmatvec_u16<double>(data, ind16, ptr, n, m, v, out, as_stream(stream), 0);
}

TM_CSR_MV_ENTRY(tm_csr_matvec_f32, float, run_csr_matvec)
TM_CSR_MV_ENTRY(tm_csr_matvec_f64, double, run_csr_matvec)
TM_CSR_MV_ENTRY(tm_csr_rmatvec_f32, float, run_csr_rmatvec)
TM_CSR_MV_ENTRY(tm_csr_rmatvec_f64, double, run_csr_rmatvec)
TM_CSR_MV_ENTRY(tm_sparse_sandwich_f32, float, run_sparse_sandwich)
TM_CSR_MV_ENTRY(tm_sparse_sandwich_f64, double, run_sparse_sandwich)

/* K7: out[j] += sum_i w[i] * X[i,j]^2 (ext/sparse.pyx:262-282) */
int tm_csr_col_sq_f32(const float *csr_data, const int32_t *csr_indices, const int64_t *csr_indptr,
                      int64_t n, int64_t m, const float *w, float *out, void *stream) {
    return run_csr_rmatvec<float>(csr_data, csr_indices, csr_indptr, n, m, w, nullptr, 0, nullptr,
                                  0, out, as_stream(stream), 1);
}
int tm_csr_col_sq_f64(const double *csr_data, const int32_t *csr_indices,
                      const int64_t *csr_indptr, int64_t n, int64_t m, const double *w,
                      double *out, void *stream) {
    return run_csr_rmatvec<double>(csr_data, csr_indices, csr_indptr, n, m, w, nullptr, 0, nullptr,
                                   0, out, as_stream(stream), 1);
}

int tm_csr_dense_sandwich_f32(const float *csr_data, const int32_t *csr_indices,
                              const int64_t *csr_indptr, int64_t n, int64_t m, const float *B,
                              int64_t r, int order_f, const float *d, const int32_t *rows,
                              int64_t n_rows, const int32_t *A_cols, int64_t nA,
                              const int32_t *B_cols, int64_t nB, float *out, void *stream) {
    return run_csr_dense<float>(csr_data, csr_indices, csr_indptr, n, m, B, r, order_f, d, rows,
                                n_rows, A_cols, nA, B_cols, nB, out, as_stream(stream));
}
int tm_csr_dense_sandwich_f64(const double *csr_data, const int32_t *csr_indices,
                              const int64_t *csr_indptr, int64_t n, int64_t m, const double *B,
                              int64_t r, int order_f, const double *d, const int32_t *rows,
                              int64_t n_rows, const int32_t *A_cols, int64_t nA,
                              const int32_t *B_cols, int64_t nB, double *out, void *stream) {
    return run_csr_dense<double>(csr_data, csr_indices, csr_indptr, n, m, B, r, order_f, d, rows,
                                 n_rows, A_cols, nA, B_cols, nB, out, as_stream(stream));
}

}  // extern "C"

namespace tmh {

// =======================================================================================
// K3 (v2)  sparse x dense cross sandwich as a slab-blocked GATHER with static register
// accumulators ("segmented wavefront reduction over column nonzeros").
//
// Format (built once per sparse block, see tabmat_amd/ext/_types.py SlabCsc): rows are cut
// into slabs of SLAB_R rows; inside a slab the nonzeros are ordered by (column, row), so the
// entries of one (slab, column) pair are one contiguous run:
//     vals[e]  value A[k, i]
//     koff[e]  (k - slab*SLAB_R) * 64 * sizeof(F)   -- byte offset of row k in the LDS slab
//     cnt [slab][column]   run length (uint16), columns padded to a multiple of 64
//     gptr[slab][group]    start of the run of column-group `group` (64 columns) in that slab
//
// Kernel: a workgroup of 8 waves owns a range of slabs and a 64-column part of the dense
// operand.  Per slab it stages  dB[k, j] = d[k] * B[k, j0 + j]  (SLAB_R x 64) into LDS with
// coalesced 16-byte loads (double-buffered, one barrier per slab).  Wave w owns sparse
// columns [64 w, 64 w + 64): lane <-> dense column j, and the 64 accumulators
// out[64 w + c, j0 + lane], c = 0..63, are STATIC registers (the column loop is fully
// unrolled), kept for the whole slab range.  Each nonzero costs one ds_read_b64 of the LDS slab
// row and one v_fma_f64; the (value, offset) stream is fetched 64 entries at a time with one
// coalesced vector load and broadcast with v_readlane.  No atomics, no LDS writes in the
// inner loop; partial results are reduced by reduce_partials_kernel.
// =======================================================================================
constexpr int SLAB_R = 128;

constexpr int GATHER_CPW = 32;                      // sparse columns (static accumulators) per wave
constexpr int GATHER_NW = 16;                       // waves per workgroup -> 512 sparse columns
constexpr int GATHER_THREADS = GATHER_NW * 64;

// Stream entries are broadcast to the 64 lanes in two ways: the VALUE through a small per-wave LDS
// ring of 64 doubles (uniform-address ds_read_b64), the ROW OFFSET with v_readlane from the register
// that holds the current 64-entry chunk (lane <-> entry), so that the slab read does not wait for
// an LDS round trip and the LDS sees 4 instead of 6 cycles per nonzero.  The ring / register are
// rewritten in place when their 64 entries are consumed.
#define TM_GATHER_STEP(A, X, LIDX)                                                        \
    const F A = ring[(LIDX)];                                                             \
    const F X = *reinterpret_cast<const F *>(                                             \
        slab + (unsigned)__builtin_amdgcn_readlane((int)kcur, (LIDX)) + lane_off);

// SCALE: the dense slab in LDS holds B unscaled (async global->LDS copy); d is folded into the
// stream value when an entry enters the ring, and entries of rows with d == 0 are redirected to
// an all-zero LDS row so that excluded rows contribute exactly nothing.
// CM: column-major LDS slab (F-ordered B): the stream's row offset row * 64 * sizeof(F) becomes
// row * sizeof(F).
template <typename F, bool SCALE, bool CM>
__device__ __forceinline__ void make_entry(F a, unsigned ko, const F *__restrict__ dl,
                                           unsigned zero_off, F &a_out, unsigned &ko_out) {
    if (SCALE) {
        const F dk = dl[ko / (64u * (unsigned)sizeof(F))];
        a_out = a * dk;
        ko_out = dk != F(0) ? ko : zero_off;
    } else {
        a_out = a;
        ko_out = CM ? ko >> 6 : ko;
    }
}

template <typename F, bool SCALE, bool CM, int C>
struct ColLoop {
    // processes static column C of the wave's group, then recurses to C + 1
    static __device__ __forceinline__ void run(F (&acc)[GATHER_CPW],
                                               const unsigned char *__restrict__ slab,
                                               F *__restrict__ ring, unsigned &kcur,
                                               const F *__restrict__ dl, unsigned zero_off,
                                               int cntv, int &pos, F &na, unsigned &nk,
                                               const F *__restrict__ vals,
                                               const unsigned *__restrict__ koff, int64_t base,
                                               int total, int lane, int lane_off) {
        int nc = __builtin_amdgcn_readlane(cntv, C);
        while (nc > 0) {
            // stay inside the current 64-entry chunk: no rotation test in the hot loop
            const int l0 = pos & 63;
            const int m = min(nc, 64 - l0);
            int t = l0;
            const int tend = l0 + m;
            for (; t + 4 <= tend; t += 4) {   // 8 independent LDS reads in flight
                TM_GATHER_STEP(a0, x0, t)
                TM_GATHER_STEP(a1, x1, t + 1)
                TM_GATHER_STEP(a2, x2, t + 2)
                TM_GATHER_STEP(a3, x3, t + 3)
                acc[C] = fma(a0, x0, acc[C]);
                acc[C] = fma(a1, x1, acc[C]);
                acc[C] = fma(a2, x2, acc[C]);
                acc[C] = fma(a3, x3, acc[C]);
            }
            for (; t < tend; ++t) {
                TM_GATHER_STEP(a0, x0, t)
                acc[C] = fma(a0, x0, acc[C]);
            }
            pos += m;
            nc -= m;
            if ((pos & 63) == 0) {
                // chunk exhausted: the prefetched chunk replaces it, next prefetch is issued
                F a_new;
                make_entry<F, SCALE, CM>(na, nk, dl, zero_off, a_new, kcur);
                __builtin_amdgcn_wave_barrier();
                ring[lane] = a_new;
                __builtin_amdgcn_wave_barrier();
                const int nxt = pos + 64 + lane;
                if (nxt < total) {
                    na = vals[base + nxt];
                    nk = koff[base + nxt];
                }
            }
        }
        if constexpr (C + 1 < GATHER_CPW)
            ColLoop<F, SCALE, CM, C + 1>::run(acc, slab, ring, kcur, dl, zero_off, cntv, pos, na, nk,
                                          vals, koff, base, total, lane, lane_off);
    }
};

// ORDER_F: the slab is kept COLUMN-major in LDS ([64 columns][SLAB_R + 1 rows], rows fastest,
// odd column stride): the vectors of an F-ordered B (consecutive rows of one column) are stored
// contiguously, and the gather's read  lane <-> column, uniform row  hits 64 distinct addresses
// (lane * (SLAB_R + 1) + row) without bank conflicts.  Row offsets of the stream (row * ROWB for the
// row-major slab) are divided by 64 when a chunk enters the ring.
template <typename F, bool ORDER_F>
struct GatherLds {
    static constexpr int ROWB = 64 * (int)sizeof(F);             // bytes per LDS slab row (row-major)
    static constexpr int COLB = (SLAB_R + 1) * (int)sizeof(F);   // bytes per LDS slab column (column-major)
    static constexpr int SLABB = ORDER_F ? ((64 * COLB + 15) / 16) * 16 : SLAB_R * ROWB;  // per buffer
    static constexpr int ZERO_OFF = 2 * SLABB;                   // all-zero row
    static constexpr int DL_OFF = ZERO_OFF + ROWB;               // d of the slab rows, 2 buffers
    static constexpr int RING_OFF = DL_OFF + 2 * SLAB_R * (int)sizeof(F);
    static constexpr int TOTAL = RING_OFF + GATHER_NW * 64 * (int)sizeof(F);
};

template <typename F, bool ORDER_F, bool VEC_OK>
__global__ __launch_bounds__(GATHER_THREADS) void csr_dense_gather_kernel(
    const F *__restrict__ vals, const unsigned *__restrict__ koff,
    const unsigned short *__restrict__ cnt, const int64_t *__restrict__ gptr, int n_groups,
    int64_t n_slabs, int64_t slabs_per_block, const F *__restrict__ B, int64_t n, int64_t r,
    int nB, const F *__restrict__ d, F *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = GatherLds<F, ORDER_F>;
    constexpr bool ASYNC = !ORDER_F && VEC_OK;   // global_load_lds staging, d folded into the stream
    constexpr int VEC = 16 / (int)sizeof(F);
    constexpr int ROWB = L::ROWB;
    constexpr int SLABB = L::SLABB;
    constexpr int NV = SLAB_R * 64 / VEC / GATHER_THREADS;  // 16-byte vectors staged per thread
    static_assert(NV >= 1 && (64 / VEC) % NV == 0, "staging: NV vectors of one row per thread");
    typedef F vec_t __attribute__((ext_vector_type(VEC)));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform -> SGPRs / s_load
    const int group = blockIdx.z * GATHER_NW + wave;
    const bool active = group < n_groups;
    const int j0 = blockIdx.y * 64;
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    const int lane_off = ORDER_F ? lane * L::COLB : lane * (int)sizeof(F);
    F *dl_all = reinterpret_cast<F *>(smem_raw + L::DL_OFF);
    F *ring = reinterpret_cast<F *>(smem_raw + L::RING_OFF) + wave * 64;

    F acc[GATHER_CPW];
#pragma unroll
    for (int c = 0; c < GATHER_CPW; ++c) acc[c] = F(0);
    for (int i = tid; i < ROWB / (int)sizeof(F); i += GATHER_THREADS)
        reinterpret_cast<F *>(smem_raw + L::ZERO_OFF)[i] = F(0);

    // ---------------- staging of the dense slab ----------------
    vec_t stage[ASYNC ? 1 : NV];
    vec_t dstage[1];   // ORDER_F: d of the thread's row vector (the same rows for all of its NV vectors)
    F dsc = F(0);
    // ASYNC: every wave copies NV KiB-sized pieces (64 lanes x 16 B) of the slab straight into
    // LDS with global_load_lds (no VGPRs, no ds_write pass); d of the slab rows goes to LDS (dl).
    // Otherwise (F-ordered or unaligned B): loads into registers, scaled by d when written to LDS.
    auto issue_slab = [&](int64_t s, int buf) {
        if (ASYNC) {
            constexpr int RPP = 1024 / ROWB;             // slab rows per 1 KiB piece
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int piece = wave * NV + i;
                const int row = piece * RPP + (lane * 16) / ROWB;
                const int c = ((lane * 16) % ROWB) / (int)sizeof(F);
                const int64_t k = min(s * SLAB_R + row, n - 1);
                const int cc = min(j0 + c, nB - VEC);
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(B + k * r + cc),
                    (__attribute__((address_space(3))) void *)(smem_raw + buf * SLABB + piece * 1024),
                    16, 0, 0);
            }
            if (tid < SLAB_R) dsc = d[min(s * SLAB_R + tid, n - 1)];
        } else if (!ORDER_F) {
            constexpr int VPR = 64 / VEC;        // vectors per slab row
            constexpr int TPR = VPR / NV;        // threads per slab row (NV consecutive vectors each)
            const int row = tid / TPR;
            const int c0 = (tid % TPR) * NV * VEC;
            const int64_t k = min(s * SLAB_R + row, n - 1);
            dsc = d[k];
#pragma unroll
            for (int i = 0; i < NV; ++i) {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    stage[ASYNC ? 0 : i][e] = B[k * r + min(j0 + c0 + i * VEC + e, nB - 1)];
            }
        } else {
            constexpr int RPC = SLAB_R / VEC;  // row-vectors per column
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = tid + i * GATHER_THREADS;
                const int c = min(j0 + q / RPC, nB - 1);
                const int row = (q % RPC) * VEC;
                const int64_t k = s * SLAB_R + row;
                if (VEC_OK) {
                    const int64_t kk = min(k, n - VEC);
                    stage[ASYNC ? 0 : i] = *reinterpret_cast<const vec_t *>(B + (int64_t)c * n + kk);
                    if (i == 0) dstage[0] = *reinterpret_cast<const vec_t *>(d + kk);
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const int64_t kk = min(k + e, n - 1);
                        stage[ASYNC ? 0 : i][e] = B[(int64_t)c * n + kk];
                        if (i == 0) dstage[0][e] = d[kk];
                    }
                }
            }
        }
    };
    auto finish_slab = [&](int64_t s, int buf) {
        unsigned char *dst = smem_raw + buf * SLABB;
        if (ASYNC) {
            if (tid < SLAB_R) dl_all[buf * SLAB_R + tid] = (s * SLAB_R + tid < n) ? dsc : F(0);
        } else if (!ORDER_F) {
            constexpr int VPR = 64 / VEC;
            constexpr int TPR = VPR / NV;
            const int row = tid / TPR;
            const int c0 = (tid % TPR) * NV * VEC;
            const bool rok = (s * SLAB_R + row < n) && dsc != F(0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = c0 + i * VEC;
                vec_t v;
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    v[e] = (rok && j0 + c + e < nB) ? dsc * stage[ASYNC ? 0 : i][e] : F(0);
                *reinterpret_cast<vec_t *>(dst + row * ROWB + c * (int)sizeof(F)) = v;
            }
        } else {
            constexpr int RPC = SLAB_R / VEC;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int q = tid + i * GATHER_THREADS;
                const int c = q / RPC;
                const int row = (q % RPC) * VEC;
                const bool cok = j0 + c < nB;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const F dv = dstage[0][e];
                    const bool ok = cok && (s * SLAB_R + row + e < n) && dv != F(0);
                    *reinterpret_cast<F *>(dst + c * L::COLB + (row + e) * (int)sizeof(F)) =
                        ok ? dv * stage[ASYNC ? 0 : i][e] : F(0);
                }
            }
        }
    };

    // Two-deep software pipeline over the slabs so that no load of an iteration depends on
    // another load issued in the same iteration (each would cost a full HBM round trip):
    //   iteration s issues   meta(s + 2)  = run lengths + stream base/total      (stage M)
    //                        head(s + 1)  = first two 64-entry chunks, from meta(s + 1) (stage H)
    //                        slab(s + 1)  = dense rows of the next slab           (stage S)
    //   then computes slab s out of LDS, completes slab(s + 1) in the other LDS buffer, barrier.
    int m_cnt = 0, m_total = 0;          // meta of slab s + 2 (after stage M)
    int64_t m_base = 0;
    int h_cnt = 0, h_total = 0;          // head of slab s + 1 (after stage H)
    int64_t h_base = 0;
    F h_va = F(0), h_na = F(0);
    unsigned h_vk = 0, h_nk = 0;
    auto load_meta = [&](int64_t s) {
        m_cnt = 0; m_total = 0; m_base = 0;
        if (!active || s >= s1) return;
        m_cnt = lane < GATHER_CPW ? (int)cnt[(s * n_groups + group) * GATHER_CPW + lane] : 0;
        m_base = gptr[s * n_groups + group];
        m_total = (int)(gptr[s * n_groups + group + 1] - m_base);
    };
    auto load_head = [&]() {             // consumes meta, issues the chunk loads
        h_cnt = m_cnt; h_total = m_total; h_base = m_base;
        h_va = F(0); h_na = F(0); h_vk = 0; h_nk = 0;
        if (lane < h_total) {
            h_va = vals[h_base + lane];
            h_vk = koff[h_base + lane];
        }
        if (64 + lane < h_total) {
            h_na = vals[h_base + 64 + lane];
            h_nk = koff[h_base + 64 + lane];
        }
    };

    if (s0 < s1) {
        load_meta(s0);
        load_head();                      // head(s0)
        load_meta(s0 + 1);                // meta(s0 + 1)
        issue_slab(s0, 0);
        finish_slab(s0, 0);
    }
    __syncthreads();
    for (int64_t s = s0; s < s1; ++s) {
        const int buf = (int)((s - s0) & 1);
        // take over the prefetched head of this slab
        const int cntv = h_cnt, total = h_total;
        const int64_t base = h_base;
        F va = h_va, na = h_na;
        unsigned vk = h_vk, nk = h_nk;
        if (s + 1 < s1) {
            load_head();                  // head(s + 1) from meta(s + 1), loaded last iteration
            issue_slab(s + 1, buf ^ 1);
        }
        load_meta(s + 2);
        if (active && total > 0) {
            int pos = 0;
            const F *dl = dl_all + buf * SLAB_R;
            const unsigned zero_off = (unsigned)(L::ZERO_OFF - buf * SLABB);
            F a_cur;
            unsigned kcur;
            make_entry<F, ASYNC, ORDER_F>(va, vk, dl, zero_off, a_cur, kcur);
            __builtin_amdgcn_wave_barrier();
            ring[lane] = a_cur;
            __builtin_amdgcn_wave_barrier();
            ColLoop<F, ASYNC, ORDER_F, 0>::run(acc, smem_raw + buf * SLABB, ring, kcur, dl, zero_off, cntv, pos,
                                      na, nk, vals, koff, base, total, lane, lane_off);
        }
        if (s + 1 < s1) finish_slab(s + 1, buf ^ 1);
        __syncthreads();
    }
    if (active) {
        // ws layout: [part = blockIdx.y][blockIdx.x][n_groups * CPW cols][64]
        F *dst = ws + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_groups + group) *
                          (GATHER_CPW * 64);
#pragma unroll
        for (int c = 0; c < GATHER_CPW; ++c) dst[c * 64 + lane] = acc[c];
    }
}

// =======================================================================================
// K3 (ELL)  sparse x dense on an interleaved-ELL twin: static iterations with skip masks.
//
// The run loop of the kernel above spends ~10 cycles per nonzero and CU on a 5 % dense block (its
// dynamic control flow, not the LDS or the FMAs: scripts/ubench/gather_dyn.hip reproduces the rate
// without any staging).  Here a (slab, 32-column group) block is I ITERATIONS of 64 slots,
// slot it*64 + 2c + u = the (2 it + u)-th nonzero (rows ascending) of column c, padded to the
// longest run of the group (I = ceil(longest / 2)).  An iteration is one coalesced 64-lane load
// {value, row offset}; its code is straight-line with STATIC accumulators and constant lanes:
// per slot v_readlane (row offset -> SGPR), v_add, ds_read_b64 of the slab row, half a uniform
// 16-byte ring read (two values), v_fma_f64.  The padding (x1.9 at 5 % density) is not executed:
// a ballot of the real slots gives a 64-bit mask, and every batch of 4 slots (2 columns) whose
// mask bits are all clear is skipped with one scalar test -- 6.9 cycles per real nonzero and CU in
// isolation (scripts/ubench/gather_ellmask.hip) against 10.0 for the run loop.
// Staging, d folding and the zero row are those of csr_dense_gather_kernel's asynchronous path.
// =======================================================================================
constexpr unsigned ELL_PADKEY = 0xFFFFFFFFu;

template <typename F>
__global__ __launch_bounds__(GATHER_THREADS) void csr_dense_ell_kernel(
    const F *__restrict__ vals, const unsigned *__restrict__ koff, const int64_t *__restrict__ gptr,
    int n_groups, int64_t n_slabs, int64_t slabs_per_block, const F *__restrict__ B, int64_t n,
    int64_t r, int nB, const F *__restrict__ d, F *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = GatherLds<F, false>;
    constexpr int VEC = 16 / (int)sizeof(F);
    constexpr int ROWB = L::ROWB;
    constexpr int SLABB = L::SLABB;
    constexpr int NV = SLAB_R * 64 / VEC / GATHER_THREADS;
    constexpr int RPP = 1024 / ROWB;
    constexpr int NA = 16 / (int)sizeof(F);          // values per 16-byte ring read
    typedef F avec_t __attribute__((ext_vector_type(NA)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.z * GATHER_NW + wave;
    const bool active = group < n_groups;
    const int j0 = blockIdx.y * 64;
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    const unsigned lane_off = lane * (unsigned)sizeof(F);
    F *dl_all = reinterpret_cast<F *>(smem_raw + L::DL_OFF);
    F *ring = reinterpret_cast<F *>(smem_raw + L::RING_OFF) + wave * 64;

    F acc[GATHER_CPW];
#pragma unroll
    for (int c = 0; c < GATHER_CPW; ++c) acc[c] = F(0);
    for (int i = tid; i < ROWB / (int)sizeof(F); i += GATHER_THREADS)
        reinterpret_cast<F *>(smem_raw + L::ZERO_OFF)[i] = F(0);

    F dsc = F(0);
    auto issue_slab = [&](int64_t s, int buf) {       // async copy of B[slab rows, j0 .. j0 + 64)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int piece = wave * NV + i;
            const int row = piece * RPP + (lane * 16) / ROWB;
            const int c = ((lane * 16) % ROWB) / (int)sizeof(F);
            const int64_t k = min(s * SLAB_R + row, n - 1);
            const int cc = min(j0 + c, nB - VEC);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(B + k * r + cc),
                (__attribute__((address_space(3))) void *)(smem_raw + buf * SLABB + piece * 1024), 16, 0,
                0);
        }
        if (tid < SLAB_R) dsc = d[min(s * SLAB_R + tid, n - 1)];
    };
    auto finish_slab = [&](int64_t s, int buf) {
        if (tid < SLAB_R) dl_all[buf * SLAB_R + tid] = (s * SLAB_R + tid < n) ? dsc : F(0);
    };
    // value scaled by d, row offset redirected to the zero row for padding and d == 0 rows
    auto enter = [&](F a, unsigned ko, const F *dl, unsigned zero_off, F &a_out, unsigned &k_out) {
        const bool real = ko != ELL_PADKEY;
        const F dk = real ? dl[ko / (unsigned)ROWB] : F(0);
        a_out = real ? a * dk : F(0);
        k_out = dk != F(0) ? ko : zero_off;
        return dk != F(0);
    };

    // meta(s + 2) / head(s + 1) pipeline as in csr_dense_gather_kernel
    int h_iters = 0;
    int64_t h_base = 0;
    int64_t mv_base = 0, mv_end = 0;    // meta of slab s + 2 as loaded (per-lane copies)
    bool m_valid = false;
    int vzero = 0;
    asm volatile("" : "+v"(vzero));     // a zero the compiler cannot see through (kept in a VGPR)
    F h_va = F(0), h_na = F(0);
    unsigned h_vk = ELL_PADKEY, h_nk = ELL_PADKEY;
    auto load_meta = [&](int64_t s) {
        m_valid = false;
        if (!active || s >= s1) return;
        // VECTOR loads on purpose (lane-dependent zero offset): a scalar load would share lgkmcnt
        // with the LDS reads, and since scalar loads return out of order the compiler waits for
        // it (one memory round trip per slab) before the next LDS access
        // (kept in VGPRs until load_head turns them into wave-uniform values one slab later)
        const int64_t *gp = gptr + s * n_groups + group + vzero;
        mv_base = gp[0];
        mv_end = gp[1];
        m_valid = true;
    };
    auto load_head = [&]() {
        h_base = m_valid ? readfirstlane_i64(mv_base) : 0;
        h_iters = m_valid ? (int)((readfirstlane_i64(mv_end) - h_base) >> 6) : 0;
        if (h_iters > 0) {
            h_va = vals[h_base + lane];
            h_vk = koff[h_base + lane];
            const int64_t q = h_base + (h_iters > 1 ? 64 : 0) + lane;
            h_na = vals[q];
            h_nk = koff[q];
        }
    };

    if (s0 < s1) {
        load_meta(s0);
        load_head();
        load_meta(s0 + 1);
        issue_slab(s0, 0);
        finish_slab(s0, 0);
    }
    __syncthreads();
    for (int64_t s = s0; s < s1; ++s) {
        const int buf = (int)((s - s0) & 1);
        const int iters = h_iters;
        const int64_t base = h_base;
        F va = h_va, na = h_na;
        unsigned vk = h_vk, nk = h_nk;
        if (s + 1 < s1) {
            load_head();
            issue_slab(s + 1, buf ^ 1);
        }
        load_meta(s + 2);
        if (active && iters > 0) {
            const F *dl = dl_all + buf * SLAB_R;
            const unsigned zero_off = (unsigned)(L::ZERO_OFF - buf * SLABB);
            const unsigned char *slab = smem_raw + buf * SLABB;
            F a_cur;
            unsigned k_cur;
            unsigned long long live = __builtin_amdgcn_ballot_w64(enter(va, vk, dl, zero_off, a_cur, k_cur));
            __builtin_amdgcn_wave_barrier();
            ring[lane] = a_cur;
            __builtin_amdgcn_wave_barrier();
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int g0 = 0; g0 < 64; g0 += 4) {
                    if (((live >> g0) & 0xFull) == 0) continue;      // 2 columns of padding / d == 0
                    F x[4], a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        x[e] = *reinterpret_cast<const F *>(
                            slab + (unsigned)__builtin_amdgcn_readlane((int)k_cur, g0 + e) + lane_off);
#pragma unroll
                    for (int e = 0; e < 4; e += NA) {
                        const avec_t av = *reinterpret_cast<const avec_t *>(ring + g0 + e);
#pragma unroll
                        for (int q = 0; q < NA; ++q) a[e + q] = av[q];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[(g0 + e) / 2] = fma(a[e], x[e], acc[(g0 + e) / 2]);
                    // pin the FMAs here (else the selector sinks them behind all loads)
                    asm volatile("" : "+v"(acc[g0 / 2]));
                    asm volatile("" : "+v"(acc[g0 / 2 + 1]));
                }
                if (it + 1 < iters) {
                    // the prefetched chunk replaces the consumed one, the next prefetch is issued
                    // (unconditional clamped loads: nothing waits for them before the next turn)
                    live = __builtin_amdgcn_ballot_w64(enter(na, nk, dl, zero_off, a_cur, k_cur));
                    __builtin_amdgcn_wave_barrier();
                    ring[lane] = a_cur;
                    __builtin_amdgcn_wave_barrier();
                    const int64_t q = base + (int64_t)min(it + 2, iters - 1) * 64 + lane;
                    na = vals[q];
                    nk = koff[q];
                }
            }
        }
        if (s + 1 < s1) finish_slab(s + 1, buf ^ 1);
        __syncthreads();
    }
    if (active) {
        F *dst = ws + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_groups + group) *
                          (GATHER_CPW * 64);
#pragma unroll
        for (int c = 0; c < GATHER_CPW; ++c) dst[c * 64 + lane] = acc[c];
    }
}

// =======================================================================================
// K3 (wide ELL)  the same static-iteration gather with lane <-> TWO dense columns: the LDS slab
// holds ELLW_R = 64 rows x 128 dense columns, a nonzero costs one 16-byte LDS read + two FMAs and
// its {value, row offset} is streamed and broadcast ONCE for 128 dense columns instead of once per
// 64-column part (the 64-column kernel spends 2 x 6.9 cycles per nonzero and CU on 128 columns,
// this geometry 11.0: scripts/ubench/gather_ellwide.hip; the floor is the LDS read bandwidth,
// 8 cycles per KiB row).  A wave owns ELLW_C = 16 sparse columns (32 accumulators), an iteration
// is 16 columns x 4 slots, slot it*64 + 4c + u = the (4 it + u)-th nonzero of column c; padding
// is skipped two slots at a time.  A workgroup covers 256 sparse columns; wider blocks use
// blockIdx.z, whose workgroups read the same slabs at the same time on the same XCD (L2 hits).
// =======================================================================================
constexpr int ELLW_R = 64;
constexpr int ELLW_C = 16;
constexpr int ELLW_NW = 256 / ELLW_C;               // waves per workgroup: 256 sparse columns
constexpr int ELLW_THREADS = ELLW_NW * 64;
constexpr int ELLW_U = 64 / ELLW_C;                  // slots per column and iteration
constexpr int ELLW_W = 128;                          // dense columns per part
constexpr int ELLW_HC = 3;                           // chunks of a block prefetched one slab ahead
constexpr int ELLW_SK = 2;                           // slots per skip batch

template <typename F>
struct EllwLds {
    static constexpr int ROWB = ELLW_W * (int)sizeof(F);
    static constexpr int SLABB = ELLW_R * ROWB;
    static constexpr int ZERO_OFF = 2 * SLABB;
    static constexpr int DL_OFF = ZERO_OFF + ROWB;
    static constexpr int RING_OFF = DL_OFF + 2 * ELLW_R * (int)sizeof(F);
    static constexpr int TOTAL = RING_OFF + ELLW_NW * 64 * (int)sizeof(F);
};

template <typename F>
__global__ __launch_bounds__(ELLW_THREADS) void csr_dense_ellw_kernel(
    const F *__restrict__ vals, const unsigned *__restrict__ koff, const int64_t *__restrict__ gptr,
    int n_groups, int64_t n_slabs, int64_t slabs_per_block, const F *__restrict__ B, int64_t n,
    int64_t r, int nB, const F *__restrict__ d, F *__restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = EllwLds<F>;
    constexpr int VEC = 16 / (int)sizeof(F);
    constexpr int ROWB = L::ROWB;
    constexpr int SLABB = L::SLABB;
    constexpr int NV = SLABB / 16 / ELLW_THREADS;      // 16-byte pieces staged per thread
    constexpr int RPP = 1024 / ROWB;                     // slab rows per 1 KiB wave piece (1 or 2)
    typedef F pair_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.z * ELLW_NW + wave;
    const bool active = group < n_groups;
    const int j0 = blockIdx.y * ELLW_W;
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int64_t s1 = min(s0 + slabs_per_block, n_slabs);
    const unsigned lane_off = lane * 2u * (unsigned)sizeof(F);
    F *dl_all = reinterpret_cast<F *>(smem_raw + L::DL_OFF);
    F *ring = reinterpret_cast<F *>(smem_raw + L::RING_OFF) + wave * 64;

    pair_t acc[ELLW_C];
#pragma unroll
    for (int c = 0; c < ELLW_C; ++c) acc[c] = pair_t{F(0), F(0)};
    for (int i = tid; i < ROWB / (int)sizeof(F); i += ELLW_THREADS)
        reinterpret_cast<F *>(smem_raw + L::ZERO_OFF)[i] = F(0);

    F dsc = F(0);
    auto issue_slab = [&](int64_t s, int buf) {       // async copy of B[slab rows, j0 .. j0 + 128)
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int piece = wave * NV + i;
            const int row = piece * RPP + (lane * 16) / ROWB;
            const int c = ((lane * 16) % ROWB) / (int)sizeof(F);
            const int64_t k = min(s * ELLW_R + row, n - 1);
            const int cc = min(j0 + c, nB - VEC);
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(B + k * r + cc),
                (__attribute__((address_space(3))) void *)(smem_raw + buf * SLABB + piece * 1024), 16, 0,
                0);
        }
        if (tid < ELLW_R) dsc = d[min(s * ELLW_R + tid, n - 1)];
    };
    auto finish_slab = [&](int64_t s, int buf) {
        if (tid < ELLW_R) dl_all[buf * ELLW_R + tid] = (s * ELLW_R + tid < n) ? dsc : F(0);
    };
    auto enter = [&](F a, unsigned ko, const F *dl, unsigned zero_off, F &a_out, unsigned &k_out) {
        const bool real = ko != ELL_PADKEY;
        const F dk = real ? dl[ko / (unsigned)ROWB] : F(0);
        a_out = real ? a * dk : F(0);
        k_out = dk != F(0) ? ko : zero_off;
        return dk != F(0);
    };

    // Pipeline over the slabs: meta(s + 2) = block start / length, head(s + 1) = the first
    // ELLW_HC chunks of the block (from meta(s + 1)), slab(s + 1) = async copy of the dense rows.
    // The head holds ALL chunks of a typical block (~2 iterations at 5 % density): vector loads
    // complete in order, so a chunk requested inside the slab loop would sit behind the 64 KB slab
    // copy issued at the top of the iteration and the wave would wait for that round trip after
    // its first chunk (measured: 6.4 ms with 2 head chunks + in-loop prefetch).
    int h_iters = 0;
    int64_t h_base = 0;
    int64_t mv_base = 0, mv_end = 0;    // meta of slab s + 2 as loaded (per-lane copies)
    bool m_valid = false;
    int vzero = 0;
    asm volatile("" : "+v"(vzero));     // a zero the compiler cannot see through (kept in a VGPR)
    F h_v[ELLW_HC];
    unsigned h_k[ELLW_HC];
#pragma unroll
    for (int c = 0; c < ELLW_HC; ++c) { h_v[c] = F(0); h_k[c] = ELL_PADKEY; }
    auto load_meta = [&](int64_t s) {
        m_valid = false;
        if (!active || s >= s1) return;
        // VECTOR loads on purpose (lane-dependent zero offset): a scalar load would share lgkmcnt
        // with the LDS reads, and since scalar loads return out of order the compiler waits for
        // it (one memory round trip per slab) before the next LDS access
        // (kept in VGPRs until load_head turns them into wave-uniform values one slab later)
        const int64_t *gp = gptr + s * n_groups + group + vzero;
        mv_base = gp[0];
        mv_end = gp[1];
        m_valid = true;
    };
    auto load_head = [&]() {
        h_base = m_valid ? readfirstlane_i64(mv_base) : 0;
        h_iters = m_valid ? (int)((readfirstlane_i64(mv_end) - h_base) >> 6) : 0;
        if (h_iters > 0) {
#pragma unroll
            for (int c = 0; c < ELLW_HC; ++c) {
                const int64_t q = h_base + (int64_t)min(c, h_iters - 1) * 64 + lane;
                h_v[c] = vals[q];
                h_k[c] = koff[q];
            }
        }
    };

    if (s0 < s1) {
        load_meta(s0);
        load_head();
        load_meta(s0 + 1);
        issue_slab(s0, 0);
        finish_slab(s0, 0);
    }
    __syncthreads();
    for (int64_t s = s0; s < s1; ++s) {
        const int buf = (int)((s - s0) & 1);
        const int iters = h_iters;
        const int64_t base = h_base;
        F cv[ELLW_HC];
        unsigned ck[ELLW_HC];
#pragma unroll
        for (int c = 0; c < ELLW_HC; ++c) { cv[c] = h_v[c]; ck[c] = h_k[c]; }
        if (s + 1 < s1) {
            load_head();
            issue_slab(s + 1, buf ^ 1);
        }
        load_meta(s + 2);
        if (active && iters > 0) {
            const F *dl = dl_all + buf * ELLW_R;
            const unsigned zero_off = (unsigned)(L::ZERO_OFF - buf * SLABB);
            const unsigned char *slab = smem_raw + buf * SLABB;
            auto do_chunk = [&](F va, unsigned vk) {
                F a_cur;
                unsigned k_cur;
                const unsigned long long live =
                    __builtin_amdgcn_ballot_w64(enter(va, vk, dl, zero_off, a_cur, k_cur));
                __builtin_amdgcn_wave_barrier();
                ring[lane] = a_cur;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int g8 = 0; g8 < 64; g8 += 8) {
                if (((live >> g8) & 0xFFull) == 0) continue;       // two whole columns dead
#pragma unroll
                for (int g0 = g8; g0 < g8 + 8; g0 += ELLW_SK) {
                    if (((live >> g0) & ((1ull << ELLW_SK) - 1)) == 0) continue;   // padding / d == 0 slots
                    pair_t x[ELLW_SK];
                    F a[ELLW_SK];
#pragma unroll
                    for (int e = 0; e < ELLW_SK; ++e)
                        x[e] = *reinterpret_cast<const pair_t *>(
                            slab + (unsigned)__builtin_amdgcn_readlane((int)k_cur, g0 + e) + lane_off);
#pragma unroll
                    for (int e = 0; e < ELLW_SK; e += 2) {
                        const pair_t av = *reinterpret_cast<const pair_t *>(ring + g0 + e);
                        a[e] = av[0];
                        a[e + 1] = av[1];
                    }
                    pair_t &A = acc[g0 / ELLW_U];
#pragma unroll
                    for (int e = 0; e < ELLW_SK; ++e) {
                        A[0] = fma(a[e], x[e][0], A[0]);
                        A[1] = fma(a[e], x[e][1], A[1]);
                    }
                    asm volatile("" : "+v"(A));      // pin the FMAs here
                }
                }
                __builtin_amdgcn_wave_barrier();     // ring is rewritten by the next chunk
            };
#pragma unroll
            for (int c = 0; c < ELLW_HC; ++c)
                if (c < iters) do_chunk(cv[c], ck[c]);
            for (int it = ELLW_HC; it < iters; ++it) {      // long blocks: the rest synchronously
                const int64_t q = base + (int64_t)it * 64 + lane;
                do_chunk(vals[q], koff[q]);
            }
        }
        if (s + 1 < s1) finish_slab(s + 1, buf ^ 1);
        __syncthreads();
    }
    if (active) {
        // ws layout: [part][block][n_groups * ELLW_C columns][128]
        F *dst = ws + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_groups + group) *
                          (ELLW_C * ELLW_W);
#pragma unroll
        for (int c = 0; c < ELLW_C; ++c)
            *reinterpret_cast<pair_t *>(dst + c * ELLW_W + lane * 2) = acc[c];
    }
}

// tmp [part][n_groups*64][64] -> out[m][nB]
template <typename F, int W = 64>
__global__ void gather_untile_kernel(const F *__restrict__ tmp, int64_t m, int64_t nB,
                                     int64_t mpad, F *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * nB) return;
    const int64_t i = e / nB, j = e % nB;
    out[e] = tmp[((j / W) * mpad + i) * W + (j % W)];
}

template <typename F>
static int run_csr_dense_gather(const F *vals, const unsigned *koff, const unsigned short *cnt,
                                const int64_t *gptr, int64_t n, int64_t m, const F *B, int64_t r,
                                int order_f, const F *d, F *out, hipStream_t st) {
    const int64_t nB = r;
    const int64_t total = m * nB;
    if (total == 0) return TM_OK;
    const int64_t n_slabs = ceil_div(n, SLAB_R);
    const int n_groups = (int)ceil_div(m, GATHER_CPW);
    const int64_t mpad = (int64_t)n_groups * GATHER_CPW;
    const int n_parts = (int)ceil_div(nB, 64);
    const int nz = (int)ceil_div(n_groups, GATHER_NW);
    if (n_slabs == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    int64_t nblk = std::max<int64_t>(1, NUM_CU / ((int64_t)n_parts * nz));
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const int64_t stride = mpad * 64;  // per (part, block)
    const size_t tmp_bytes = align256(sizeof(F) * (size_t)(n_parts * stride));
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_parts * nblk * stride) + 256,
                           &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    const size_t lds = order_f ? (size_t)GatherLds<F, true>::TOTAL : (size_t)GatherLds<F, false>::TOTAL;
    // 16-byte vector loads need aligned bases and row/column strides that keep every vector
    // 16-byte aligned and entirely inside the matrix
    constexpr int VEC = 16 / (int)sizeof(F);
    const bool base_ok = (reinterpret_cast<uintptr_t>(B) & 15) == 0 && n >= VEC && nB >= VEC;
    const bool vec_ok = order_f ? (base_ok && n % VEC == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0)
                                : (base_ok && r % VEC == 0);
    auto kern = order_f ? (vec_ok ? &csr_dense_gather_kernel<F, true, true>
                                  : &csr_dense_gather_kernel<F, true, false>)
                        : (vec_ok ? &csr_dense_gather_kernel<F, false, true>
                                  : &csr_dense_gather_kernel<F, false, false>);
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts, (unsigned)nz), dim3(GATHER_THREADS), lds,
                       st, vals, koff, cnt, gptr, n_groups, n_slabs, spb, B, n, r, (int)nB, d, ws);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_parts, tmp, n_parts * stride, false,
                                   st);
    if (rc) return rc;
    hipLaunchKernelGGL((gather_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       st, tmp, m, nB, mpad, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_csr_dense_ell(const F *vals, const unsigned *koff, const int64_t *gptr, int64_t n,
                             int64_t m, const F *B, int64_t r, const F *d, F *out, hipStream_t st) {
    const int64_t nB = r;
    const int64_t total = m * nB;
    if (total == 0) return TM_OK;
    constexpr int VEC = 16 / (int)sizeof(F);
    if ((reinterpret_cast<uintptr_t>(B) & 15) != 0 || r % VEC != 0 || nB < VEC) {
        set_error("tm_csr_dense_sandwich_ell: B must be C-ordered with 16-byte aligned rows");
        return TM_EUNSUPPORTED;
    }
    const int64_t n_slabs = ceil_div(n, SLAB_R);
    const int n_groups = (int)ceil_div(m, GATHER_CPW);
    const int64_t mpad = (int64_t)n_groups * GATHER_CPW;
    const int n_parts = (int)ceil_div(nB, 64);
    const int nz = (int)ceil_div(n_groups, GATHER_NW);
    if (n_slabs == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    int64_t nblk = std::max<int64_t>(1, NUM_CU / ((int64_t)n_parts * nz));
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const int64_t stride = mpad * 64;  // per (part, block)
    const size_t tmp_bytes = align256(sizeof(F) * (size_t)(n_parts * stride));
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_parts * nblk * stride) + 256,
                           &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    const size_t lds = (size_t)GatherLds<F, false>::TOTAL;
    auto kern = &csr_dense_ell_kernel<F>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts, (unsigned)nz), dim3(GATHER_THREADS),
                       lds, st, vals, koff, gptr, n_groups, n_slabs, spb, B, n, r, (int)nB, d, ws);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_parts, tmp, n_parts * stride, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((gather_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                       st, tmp, m, nB, mpad, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_csr_dense_ellw(const F *vals, const unsigned *koff, const int64_t *gptr, int64_t n,
                              int64_t m, const F *B, int64_t r, const F *d, F *out, hipStream_t st) {
    const int64_t nB = r;
    const int64_t total = m * nB;
    if (total == 0) return TM_OK;
    constexpr int VEC = 16 / (int)sizeof(F);
    if ((reinterpret_cast<uintptr_t>(B) & 15) != 0 || r % VEC != 0 || nB < VEC) {
        set_error("tm_csr_dense_sandwich_ellw: B must be C-ordered with 16-byte aligned rows");
        return TM_EUNSUPPORTED;
    }
    if (m % ELLW_C != 0) {
        set_error("tm_csr_dense_sandwich_ellw: m must be a multiple of tm_ellw_group_cols()");
        return TM_EINVAL;
    }
    const int64_t n_slabs = ceil_div(n, ELLW_R);
    const int n_groups = (int)(m / ELLW_C);
    const int n_parts = (int)ceil_div(nB, ELLW_W);
    const int nz = (int)ceil_div(n_groups, ELLW_NW);
    if (n_slabs == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        return TM_OK;
    }
    int64_t nblk = std::max<int64_t>(1, NUM_CU / ((int64_t)n_parts * nz));
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const int64_t stride = m * ELLW_W;  // per (part, block)
    const size_t tmp_bytes = align256(sizeof(F) * (size_t)(n_parts * stride));
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + sizeof(F) * (size_t)((int64_t)n_parts * nblk * stride) + 256,
                           &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    const size_t lds = (size_t)EllwLds<F>::TOTAL;
    auto kern = &csr_dense_ellw_kernel<F>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts, (unsigned)nz), dim3(ELLW_THREADS),
                       lds, st, vals, koff, gptr, n_groups, n_slabs, spb, B, n, r, (int)nB, d, ws);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_parts, tmp, n_parts * stride, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((gather_untile_kernel<F, ELLW_W>), dim3((unsigned)ceil_div(total, 256)),
                       dim3(256), 0, st, tmp, m, nB, m, out);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_sparse_chunk_cols(void) { return 128; }
int tm_sparse_sandwich_chunked_f32(const float *csr_data, const int32_t *csr_indices,
                                   const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                   const float *d, float *out, void *stream) {
    return tmh::run_sparse_sandwich_chunked<float>(csr_data, csr_indices, cptr, n, m, nnz, d, out,
                                                   tmh::as_stream(stream));
}
int tm_sparse_sandwich_chunked_f64(const double *csr_data, const int32_t *csr_indices,
                                   const int32_t *cptr, int64_t n, int64_t m, int64_t nnz,
                                   const double *d, double *out, void *stream) {
    return tmh::run_sparse_sandwich_chunked<double>(csr_data, csr_indices, cptr, n, m, nnz, d, out,
                                                    tmh::as_stream(stream));
}

// the same two forms with the columns as one byte per chunk-major entry (column inside its 128-column chunk)
int tm_sparse_sandwich_chunked_u8_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *cptr, int64_t n,
                                      int64_t m, int64_t nnz, const float *d, float *out, void *stream) {
    return tmh::run_sparse_sandwich_chunked<float, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), cptr, n, m,
                                                         nnz, d, out, tmh::as_stream(stream));
}
int tm_sparse_sandwich_chunked_u8_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *cptr, int64_t n,
                                      int64_t m, int64_t nnz, const double *d, double *out, void *stream) {
    return tmh::run_sparse_sandwich_chunked<double, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), cptr, n, m,
                                                          nnz, d, out, tmh::as_stream(stream));
}
int tm_sparse_sandwich_chunked_rows_u8_f32(const float *cm_data, const uint8_t *cm_col8, const int32_t *row_ranges,
                                           int64_t n_sel, int64_t m, int64_t nnz, const float *d_sel, float *out,
                                           void *stream) {
    return tmh::run_sparse_sandwich_chunked<float, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), row_ranges,
                                                         n_sel, m, nnz, d_sel, out, tmh::as_stream(stream), 1);
}
int tm_sparse_sandwich_chunked_rows_u8_f64(const double *cm_data, const uint8_t *cm_col8, const int32_t *row_ranges,
                                           int64_t n_sel, int64_t m, int64_t nnz, const double *d_sel, double *out,
                                           void *stream) {
    return tmh::run_sparse_sandwich_chunked<double, true>(cm_data, reinterpret_cast<const int32_t *>(cm_col8), row_ranges,
                                                          n_sel, m, nnz, d_sel, out, tmh::as_stream(stream), 1);
}

int tm_sparse_sandwich_chunked_rows_f32(const float *cm_data, const int32_t *cm_indices,
                                        const int32_t *row_ranges, int64_t n_sel, int64_t m,
                                        int64_t nnz, const float *d_sel, float *out, void *stream) {
    return tmh::run_sparse_sandwich_chunked<float>(cm_data, cm_indices, row_ranges, n_sel, m, nnz, d_sel,
                                                   out, tmh::as_stream(stream), 1);
}
int tm_sparse_sandwich_chunked_rows_f64(const double *cm_data, const int32_t *cm_indices,
                                        const int32_t *row_ranges, int64_t n_sel, int64_t m,
                                        int64_t nnz, const double *d_sel, double *out, void *stream) {
    return tmh::run_sparse_sandwich_chunked<double>(cm_data, cm_indices, row_ranges, n_sel, m, nnz, d_sel,
                                                    out, tmh::as_stream(stream), 1);
}

int tm_ellw_rows(void) { return tmh::ELLW_R; }
int tm_ellw_group_cols(void) { return tmh::ELLW_C; }
int tm_csr_dense_sandwich_ellw_f32(const float *vals, const uint32_t *koff, const int64_t *gptr,
                                   int64_t n, int64_t m, const float *B, int64_t r, const float *d,
                                   float *out, void *stream) {
    return tmh::run_csr_dense_ellw<float>(vals, koff, gptr, n, m, B, r, d, out, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_ellw_f64(const double *vals, const uint32_t *koff, const int64_t *gptr,
                                   int64_t n, int64_t m, const double *B, int64_t r, const double *d,
                                   double *out, void *stream) {
    return tmh::run_csr_dense_ellw<double>(vals, koff, gptr, n, m, B, r, d, out, tmh::as_stream(stream));
}
int tm_slab_rows(void) { return tmh::SLAB_R; }
int tm_slab_group_cols(void) { return tmh::GATHER_CPW; }

int tm_csr_dense_sandwich_ell_f32(const float *vals, const uint32_t *koff, const int64_t *gptr,
                                  int64_t n, int64_t m, const float *B, int64_t r, const float *d,
                                  float *out, void *stream) {
    return tmh::run_csr_dense_ell<float>(vals, koff, gptr, n, m, B, r, d, out, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_ell_f64(const double *vals, const uint32_t *koff, const int64_t *gptr,
                                  int64_t n, int64_t m, const double *B, int64_t r, const double *d,
                                  double *out, void *stream) {
    return tmh::run_csr_dense_ell<double>(vals, koff, gptr, n, m, B, r, d, out, tmh::as_stream(stream));
}

int tm_csr_dense_sandwich_slab_f32(const float *vals, const uint32_t *koff, const uint16_t *cnt,
                                   const int64_t *gptr, int64_t n, int64_t m, const float *B,
                                   int64_t r, int order_f, const float *d, float *out, void *stream) {
    return tmh::run_csr_dense_gather<float>(vals, koff, cnt, gptr, n, m, B, r, order_f, d, out,
                                            tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_slab_f64(const double *vals, const uint32_t *koff, const uint16_t *cnt,
                                   const int64_t *gptr, int64_t n, int64_t m, const double *B,
                                   int64_t r, int order_f, const double *d, double *out, void *stream) {
    return tmh::run_csr_dense_gather<double>(vals, koff, cnt, gptr, n, m, B, r, order_f, d, out,
                                             tmh::as_stream(stream));
}

}  // extern "C"
