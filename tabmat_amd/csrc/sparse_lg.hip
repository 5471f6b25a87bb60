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
// Round 0 of every block sits at a fixed stride (no pointer chase).  A column's 9th, 10th ...
// nonzero of a slab (0.4 % of the columns at 5 %) is an overflow ENTRY of 16 bytes {value, koff,
// column} in a separate array; the block header -- number of entries in koff bits 20..31 of slot
// 0 of chunk 0, index of the first in bits 20..31 of slots 1..3 -- travels with round 0, the
// entries come in through SCALAR loads and run through the chunk code as one-entry chunks.
// Positions of a column are compacted (nonzeros first), so "position `it` of column j is used by
// either half" is monotone in `it`; the first LG_UNC positions run unconditionally, the rest in
// pairs behind one scalar bit test.
#include "common.hpp"
#include "reduce.hpp"

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
    static constexpr int ROWB = LG_W * (int)sizeof(F);      // bytes of a slab row
    // LDS row stride = the row itself: the 1 KiB / 512 B gather reads are contiguous
    static constexpr int RSB = ROWB;
    static constexpr int BUFB = (LG_R + 1) * RSB;           // [zero row][LG_R rows]
    static constexpr int DL_OFF = 2 * BUFB;
    static constexpr int DLN = LG_R + 2;                       // [0][d of LG_R rows][pad]
    static constexpr int TOTAL = DL_OFF + 2 * DLN * (int)sizeof(F);
};

// Buffer descriptor over [base, base + bytes): loads beyond the range return 0 (the ragged last
// slab needs no per-lane clamping).  Device pass only; the host pass sees empty stand-ins (it only
// needs the kernel's stub).
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t lg_rsrc_t;
__device__ __forceinline__ lg_rsrc_t lg_rsrc(const void *base, int64_t bytes) {
    const unsigned nb = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)nb, 0x00020000);
}
__device__ __forceinline__ void lg_buf_to_lds16(lg_rsrc_t rs, void *lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff,
                                             soff, 0, 0);
}
template <typename F>
__device__ __forceinline__ F lg_buf_load(lg_rsrc_t rs, int voff) {
    if constexpr (sizeof(F) == 8)
        return __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 0));
    else
        return __builtin_bit_cast(F, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
}
#else
struct lg_rsrc_t {};
inline lg_rsrc_t lg_rsrc(const void *, int64_t) { return {}; }
inline void lg_buf_to_lds16(lg_rsrc_t, void *, int, int) {}
template <typename F>
inline F lg_buf_load(lg_rsrc_t, int) { return F(0); }
#endif

// Which of a wave's NV copy pieces are issued before chunk cq of NCQ: spread over the first
// LG_FRONT chunks of the iteration (the copy must have landed at the barrier that ends it).
#ifndef LG_FRONT
#define LG_FRONT 4
#endif
constexpr int lg_piece_lo(int cq, int nv, int ncq) {
    const int f = LG_FRONT < ncq ? LG_FRONT : ncq;
    return cq >= f ? nv : cq * nv / f;
}

template <int SEL>
__device__ __forceinline__ unsigned lg_bcast_add(unsigned k, unsigned off) {
    unsigned r;
    asm("v_add_u32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "=v"(r)
        : "v"(k), "v"(off), "n"(SEL));
    return r;
}
template <int SEL>
__device__ __forceinline__ void lg_fmac(double &acc, double a, double x) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(a), "v"(x), "n"(SEL));
}
template <int SEL>
__device__ __forceinline__ void lg_fmac(float &acc, float a, float x) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf"
        : "+v"(acc)
        : "v"(a), "v"(x), "n"(SEL));
}

// Addressing: everything a wave touches per slab advances by a CONSTANT stride, so the kernel
// keeps running pointers (one 64-bit add per stream and slab) and per-lane byte offsets computed
// once.  The first version recomputed `B + min(s * 64 + row, n - 1) * r + c` and the stream index
// with 64-bit multiplies for every 1 KiB piece and every chunk: ~300 scalar / address
// instructions per wave and slab, 3.3 ms of the kernel's 6.1 ms with all arithmetic and all LDS
// traffic removed (profiles/r2_k3_ablation.txt).  Only the LAST slab of the matrix can hold rows
// beyond n; it uses clamped offsets (tail_*).
// NG = groups per wave (1: 16 waves x 16 columns, the shipped geometry).
//
// CSUM = true: the column sums  A^T d  (length m, kernel column order) come out of the same pass --
// `value * d` is formed for every slot anyway, one v_add per chunk accumulates it per lane, the 8
// positions of a column are summed at the end.  StandardizedMatrix.sandwich then needs no second
// pass over the sparse block (reference: standardized_mat.py:149-150 calls transpose_matvec).
//
// CP = true: the COMPACT stream (round 3).  The padded stream spends 12 bytes on every slot, 60 % of them
// padding at 5 % density (7.7 GB at cfg4 for 3.1 GB of nonzeros).  Compact form of a (slab, group) block:
//   cmap  uint8 [32 slots][4 chunks]   1 + row in slab of the slot's nonzero, 0 = padding (one dword per lane)
//   cvals F[...]                        the values of the block's real slots only, chunk after chunk in slot order
//   crec  {int64 first value, uint32 overflow entries, uint32 first entry}   one 16-byte record per block
// (passed through the vals / koff / xptr arguments).  A lane finds its value at first + (real slots of the
// earlier chunks) + (real slots below its own in the chunk): one ballot + two popcounts per chunk.  The
// map of slab w + 2 and the record of slab w + 2 (a SCALAR load) are requested at the top of slab w, the
// values of slab w + 1 between the chunks of slab w as before: 2.7 GB instead of 7.7.
template <typename F, int UNC, int NG, bool CSUM, bool CP = false>
__global__ __launch_bounds__(LG_THREADS / NG) void csr_dense_lg_kernel(
    const F *__restrict__ vals, const unsigned *__restrict__ koff, const int64_t *__restrict__ xptr,
    const F *__restrict__ xvals, const unsigned *__restrict__ xkoff, int n_groups, int64_t n_slabs,
    int64_t slabs_per_block, const F *__restrict__ B, int64_t n, int64_t r, int nB,
    const F *__restrict__ d, F *__restrict__ ws, F *__restrict__ ws_csum, int *__restrict__ prog) {
    const uint4 *__restrict__ xent = reinterpret_cast<const uint4 *>(xkoff);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = LgLds<F>;
    constexpr int VEC = 16 / (int)sizeof(F);             // dense columns per 16-byte read
    constexpr int NRD = L::ROWB / 512;                    // reads per nonzero and lane (2 / 1)
    constexpr int ROWB = L::ROWB;
    constexpr int SLABB = LG_R * ROWB;
    constexpr int NWV = LG_NW / NG;                        // waves of this geometry
    constexpr int NTH = NWV * 64;
    constexpr int NV = SLABB / 16 / NTH;                   // 1 KiB pieces copied per wave
    constexpr int RPP = 1024 / ROWB;                       // slab rows per 1 KiB wave piece
    constexpr int RSB = L::RSB;
    static_assert(RSB == 1024 || RSB == 512, "row stride");
    typedef F vec_t __attribute__((ext_vector_type(VEC)));
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.z * LG_NW + wave;          // first group of the wave; q-th: + q * NWV
    const bool active = group < n_groups;
    const int j0 = blockIdx.y * LG_W;
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int ns = (int)(min(s0 + slabs_per_block, n_slabs) - s0);   // slabs of this workgroup
    const unsigned lane_off = (unsigned)(lane & 31) * 16u;
    const int lane32 = (lane >> 5) * 16 + (lane & 15);     // the slot this lane loads
    F *dl_all = reinterpret_cast<F *>(smem_raw + L::DL_OFF);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_byte *)smem_raw;

    F acc[NG][8][NRD][VEC];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int u = 0; u < NRD; ++u)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[q][j][u][e] = F(0);
    // ---- column sums of value * d per slot (CSUM only) ----
    F csum[NG][LG_CHUNKS];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int c = 0; c < LG_CHUNKS; ++c) csum[q][c] = F(0);
    // zero rows of both buffers, d = 0 for them
    for (int i = tid; i < 2 * ROWB / (int)sizeof(F); i += NTH) {
        const int b = i / (ROWB / (int)sizeof(F)), c = i % (ROWB / (int)sizeof(F));
        reinterpret_cast<F *>(smem_raw + b * L::BUFB)[c] = F(0);       // (the pad of a row is never read)
    }
    if (tid < 2) dl_all[tid * L::DLN] = F(0);
    if (ns <= 0) return;

    // ---- per-lane constants of the slab copy: piece i of this wave = 1 KiB = RPP slab rows ----
    // B and d are read through buffer descriptors rebuilt per slab (base = the slab's first
    // byte, range = what is left of the array): rows beyond n - 1 in the ragged last slab are out
    // of range and read as 0 without any per-lane clamping.
    const int cc = min(j0 + ((lane * 16) % ROWB) / (int)sizeof(F), nB - VEC);
    // byte offset of this lane's 16 bytes inside a slab of B for piece 0; piece i lies RPP rows
    // further (a uniform stride, added to the scalar base)
    const int row0 = wave * NV * RPP + (lane * 16) / ROWB;
    const unsigned boff0 = (unsigned)((row0 * r + cc) * (int64_t)sizeof(F));
    const int64_t pstride = (int64_t)RPP * r * (int64_t)sizeof(F);
    const int64_t bstride = (int64_t)LG_R * r * (int64_t)sizeof(F);
    const char *bnext = reinterpret_cast<const char *>(B) + s0 * bstride;        // slab to copy next
    int64_t bleft = n * r * (int64_t)sizeof(F) - s0 * bstride;                     // bytes from there on
    // (uniform bases + small per-lane offsets: scalar-base addressing, no 64-bit VGPR pointers)
    const F *dnext = d + s0 * LG_R;                                                // its d
    int64_t dleft = (n - s0 * LG_R) * (int64_t)sizeof(F);
    const int64_t sstride = (int64_t)n_groups * (LG_CHUNKS * LG_SLOTS);
    const F *vnext = vals + (s0 * n_groups + group) * (int64_t)(LG_CHUNKS * LG_SLOTS);
    const unsigned *knext = koff + (s0 * n_groups + group) * (int64_t)(LG_CHUNKS * LG_SLOTS);
    // compact stream: map dwords and block records of the slab whose map is requested next
    static_assert(!CP || NG == 1, "compact stream: one group per wave");
#if defined(LG_ABLATE_SAME_GROUP)     // timing only: every wave walks the stream of the workgroup's FIRST group
    const int sgroup = blockIdx.z * LG_NW;   // (identical work per slab in all 16 waves: no barrier skew)
#else
    const int sgroup = group;
#endif
    const unsigned *mnext = koff + (s0 * n_groups + sgroup) * (int64_t)LG_SLOTS;
    const int64_t *rnext = xptr + (s0 * n_groups + sgroup) * (int64_t)2;
    constexpr int KSH = RSB == 1024 ? 10 : 9;     // koff = (1 + row) << KSH

    F dsc = F(0);
    // `which` = index of the slab inside the workgroup's range (0 .. ns - 1)
    auto issue_piece = [&](int buf, int i) {
        // piece = RPP slab rows (1 for f64, 2 unpadded ones for f32): 64 x 16 contiguous bytes
        lg_buf_to_lds16(lg_rsrc(bnext, bleft), smem_raw + buf * L::BUFB + RSB + (wave * NV + i) * RPP * RSB,
                        (int)boff0, (int)(i * pstride));
    };
    auto load_d = [&]() {
        if (tid < LG_R) {
            dsc = lg_buf_load<F>(lg_rsrc(dnext, dleft), tid * (int)sizeof(F));
        }
    };
    auto finish_slab = [&](int buf) {
        if (tid < LG_R) dl_all[buf * L::DLN + 1 + tid] = dsc;
    };
    auto advance = [&]() {                       // the "next" pointers move one slab on
        bnext += bstride;
        bleft -= bstride;
        dnext += LG_R;
        dleft -= LG_R * (int64_t)sizeof(F);
        vnext += sstride;
        knext += sstride;
    };
    // ---- compact stream state: maps of the current / next / next-but-one slab, records likewise
    // (the record travels as one dword per lane -- lane & 3 -- and is taken apart with v_readlane when its
    // slab comes up: decoded where it is loaded, the compiler waits for the load at once)
    unsigned mcur = 0u, mnxt = 0u, mnx2 = 0u, rv_nxt = 0u, rv_nx2 = 0u;
    int64_t vb_nxt = 0;                          // first value of the next slab's block (uniform)
    unsigned nrec_cur = 0u, rec0_cur = 0u, nrec_nxt = 0u, rec0_nxt = 0u;
    int cpre = 0;                                // real slots of the chunks already requested (uniform)
    auto load_map = [&](unsigned &mm, unsigned &rv) {
        mm = mnext[lane32];
        rv = reinterpret_cast<const unsigned *>(rnext)[lane & 3];
        mnext += (int64_t)n_groups * LG_SLOTS;
        rnext += (int64_t)n_groups * 2;
    };
    auto decode_nxt = [&]() {                    // the record of the slab whose values are requested next
        vb_nxt = (int64_t)(((uint64_t)(unsigned)__builtin_amdgcn_readlane((int)rv_nxt, 1) << 32) |
                           (unsigned)__builtin_amdgcn_readlane((int)rv_nxt, 0));
        nrec_nxt = (unsigned)__builtin_amdgcn_readlane((int)rv_nxt, 2);
        rec0_nxt = (unsigned)__builtin_amdgcn_readlane((int)rv_nxt, 3);
        cpre = 0;
    };

    // the four chunks of round 0 of the NEXT slab, requested while the current one is worked on
    F pv[NG][LG_CHUNKS];
    unsigned pk[NG][LG_CHUNKS];
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int c = 0; c < LG_CHUNKS; ++c) { pv[q][c] = F(0); pk[q][c] = 0u; }
    bool act[NG];
#pragma unroll
    for (int q = 0; q < NG; ++q) act[q] = group + q * NWV < n_groups;
    auto load_chunk = [&](int q, int c) {
        if (act[q]) {
            if constexpr (CP) {
                // value of this lane's slot in chunk c of the NEXT slab (map mnxt, first value vb_nxt)
                const bool real = ((mnxt >> (8 * c)) & 0xffu) != 0u;
                const unsigned long long b = __builtin_amdgcn_ballot_w64(real);
                const unsigned m32 = ((unsigned)b & 0xFFFFu) | (((unsigned)(b >> 32) & 0xFFFFu) << 16);
                const int rank = __builtin_popcount(m32 & ((1u << lane32) - 1u));
                const int idx = real ? cpre + rank : 0;          // (padding: any valid address; masked on use)
                pv[q][c] = vals[vb_nxt + idx];
                cpre += __builtin_popcount(m32);
            } else {
                pv[q][c] = vnext[(q * NWV * LG_CHUNKS + c) * LG_SLOTS + lane32];
                pk[q][c] = knext[(q * NWV * LG_CHUNKS + c) * LG_SLOTS + lane32];
            }
        }
    };

    // prologue: slab 0 of the range into buffer 0
#pragma unroll
    for (int i = 0; i < NV; ++i) issue_piece(0, i);
    load_d();
    if (active) {
        if constexpr (CP) {
            // maps / records of slabs 0 and 1; the values of slab 0 (addressed through its map)
            load_map(mnxt, rv_nxt);
            if (ns > 1) load_map(mnx2, rv_nx2);
            decode_nxt();
        }
#pragma unroll
        for (int q = 0; q < NG; ++q)
#pragma unroll
            for (int c = 0; c < LG_CHUNKS; ++c) load_chunk(q, c);
        if constexpr (CP) {
            mcur = mnxt; nrec_cur = nrec_nxt; rec0_cur = rec0_nxt;
            mnxt = mnx2; rv_nxt = rv_nx2;
        }
    }
    advance();
    finish_slab(0);
    __syncthreads();
#if defined(LG_ABLATE_NO_STREAM)      // (the re-used slots must not announce overflow entries)
#pragma unroll
    for (int q = 0; q < NG; ++q)
#pragma unroll
        for (int c = 0; c < LG_CHUNKS; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pk[q][c] &= LG_KMASK;
        }
#endif

    unsigned slab_base = lds_base;                // LDS address of the current buffer
    unsigned next_base = lds_base + (unsigned)L::BUFB;
    const F *dl = dl_all;
    const F *dl_next = dl_all + L::DLN;
    // soft lockstep of the workgroups that share a slab range (blockIdx.z): nobody runs more than one
    // slab ahead of its neighbour, so the second reader of a slab of B finds it in the XCD's L2
    int *const my_prog = prog ? prog + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * gridDim.z : nullptr;
    const int zn = (int)((blockIdx.z + 1) % gridDim.z);
    bool waiting = true;               // given up for good after one timeout (neighbour not resident)
    for (int w = 0; w < ns; ++w) {
        const int buf = w & 1;
        const bool more = w + 1 < ns;
        if (more) load_d();
        if constexpr (CP) {
            if (active && more) decode_nxt();    // (its map and record were requested a slab ago)
            if (active && w + 2 < ns) load_map(mnx2, rv_nx2);
        }
        int pseen = 0x7fffffff;
        if (my_prog != nullptr && tid == 0 && gridDim.z > 1)
            pseen = __hip_atomic_load(my_prog + zn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!active) {                  // a wave without columns still copies its share
            if (more) {
#pragma unroll
                for (int i = 0; i < NV; ++i) issue_piece(buf ^ 1, i);
            }
        } else {
            // one chunk: fold d into the values, redirect d == 0 rows to the zero row, then the
            // positions of columns j = 2c, 2c + 1 of both halves
            auto do_chunk = [&](auto qc_, auto cc_, F v, unsigned kraw) {
                constexpr int c = decltype(cc_)::value;
                constexpr int q = decltype(qc_)::value;
                const unsigned kk = kraw & LG_KMASK;
                const unsigned long long real = __builtin_amdgcn_ballot_w64(kk != 0u);
                const unsigned comb = ((unsigned)real | (unsigned)(real >> 32)) & 0xFFFFu;
                if (comb == 0u) return;
                const F dk = dl[kk >> (RSB == 1024 ? 10 : 9)];          // row = kk / RSB
                F a = v * dk;
                if constexpr (CP) a = kk != 0u ? a : F(0);               // (compact stream: no zero stored for padding)
                if constexpr (CSUM) csum[q][c] += a;                     // (padding: v = 0, dk = d[-1] = 0)
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
                            x[decltype(e)::value][u] = *reinterpret_cast<
                                const __attribute__((address_space(3))) vec_t *>(
                                (lds_byte *)(uintptr_t)(addr + u * 512));
                    });
                    static_for<cnt>([&](auto e) {
                        constexpr int SEL = jl * 8 + it + decltype(e)::value;
#pragma unroll
                        for (int u = 0; u < NRD; ++u)
#pragma unroll
                            for (int qq = 0; qq < VEC; ++qq)
                                lg_fmac<SEL>(acc[q][2 * c + jl][u][qq], a, x[decltype(e)::value][u][qq]);
                    });
                };
                static_for<2>([&](auto jlc) {
                    constexpr int jl = decltype(jlc)::value;
                    if (!(comb & (1u << (jl * 8)))) return;       // neither half has a nonzero here
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
            // Pass 0: the four chunks of round 0.  The copy of the next slab and the loads of its
            // chunks are issued BETWEEN the chunks: issued in one burst at the top of the iteration
            // the 16 waves queue up behind the vector-memory pipe for ~2000 cycles before any of
            // them computes.
            // Passes 1 .. nrec: one overflow entry each (a column's 9th, 10th ... nonzero of the
            // slab), turned into a chunk that holds just that entry and sent through the same code.
            // The block header travels with round 0 and the entries are fetched with SCALAR loads
            // (first one here, at the top): a vector load at the point of use would make its
            // s_waitcnt vmcnt(0) wait for the slab copy issued a moment earlier -- one full memory
            // round trip in the middle of the iteration of 63 % of the workgroups (6.0 -> 4.9 ms).
            static_for<NG>([&](auto qc_) {
            constexpr int q = decltype(qc_)::value;
            if (!act[q]) return;
            int nrec;
            int64_t rec0 = 0;
            uint4 e = uint4{0u, 0u, 0u, 0u};
            if constexpr (CP) {
                nrec = (int)nrec_cur;
                rec0 = (int64_t)rec0_cur;
                if (nrec > 0) e = xent[rec0];
            } else {
                nrec = __builtin_amdgcn_readlane((int)pk[q][0], 0) >> 20 & 0xFFF;
                if (nrec > 0) {
                    rec0 = (int64_t)((unsigned)__builtin_amdgcn_readlane((int)pk[q][0], 1) >> 20) |
                           (int64_t)((unsigned)__builtin_amdgcn_readlane((int)pk[q][0], 2) >> 20) << 12 |
                           (int64_t)((unsigned)__builtin_amdgcn_readlane((int)pk[q][0], 3) >> 20) << 24;
                    e = xent[rec0];
                }
            }
            // pass 0 of chunk c (shared by both forms)
            auto round0 = [&](auto cc_, F &v, unsigned &k) {
                constexpr int c = decltype(cc_)::value;
                constexpr int cq = q * LG_CHUNKS + c;      // chunk number inside the iteration
                constexpr int NCQ = NG * LG_CHUNKS;
                if (more) {
#pragma unroll
                    for (int i = lg_piece_lo(cq, NV, NCQ); i < lg_piece_lo(cq + 1, NV, NCQ); ++i)
#if !defined(LG_ABLATE_NO_COPY)       // timing only: the slab of B is not refreshed
                        issue_piece(buf ^ 1, i);
#else
                        (void)i;
#endif
                }
                if constexpr (CP) {
                    k = ((mcur >> (8 * c)) & 0xffu) << KSH;
                    v = pv[q][c];                 // (a padding slot holds SOME value of the block: masked in do_chunk)
                } else {
                    v = pv[q][c];
                    k = pk[q][c];
                }
#if defined(LG_ABLATE_NO_STREAM)      // timing only: every slab re-uses the first slab's slots (no stream traffic)
                (void)0;
#else
                if (more) load_chunk(q, c);
#endif
            };
            auto entry_of = [&](int pass, int &ec, int &eslot, F &eval) {
                if (pass > 1) e = xent[rec0 + pass - 1];     // a second entry in a block is rare
                const int wcol = (int)(e.w & 15u);            // column 8h + j of the group
                ec = (wcol & 7) >> 1;
                eslot = (wcol >> 3) * 16 + (wcol & 1) * 8;    // position 0 of that column
                if constexpr (sizeof(F) == 8)
                    eval = __builtin_bit_cast(F, ((unsigned long long)e.y << 32) | e.x);
                else
                    eval = __builtin_bit_cast(F, e.x);
            };
            for (int pass = 0; pass <= nrec; ++pass) {
                int ec, eslot;
                F eval;
                entry_of(pass, ec, eslot, eval);
                static_for<LG_CHUNKS>([&](auto cc_) {
                    constexpr int c = decltype(cc_)::value;
                    F v;
                    unsigned k;
                    if (pass == 0) {
                        round0(cc_, v, k);
                    } else {
                        const bool hit = ec == c && lane32 == eslot;
                        v = hit ? eval : F(0);
                        k = hit ? e.z : 0u;
                    }
                    // (one call site: a second instantiation of the chunk code for the overflow
                    // passes costs 18 registers)
                    if (pass == 0 || ec == c) do_chunk(qc_, cc_, v, k);
                });
            }
            });
        }
        if (more) {
            advance();
            finish_slab(buf ^ 1);
        }
        if constexpr (CP) {
            mcur = mnxt; nrec_cur = nrec_nxt; rec0_cur = rec0_nxt;
            mnxt = mnx2; rv_nxt = rv_nx2;
        }
        if (my_prog != nullptr && tid == 0 && gridDim.z > 1) {
            __hip_atomic_store(my_prog + blockIdx.z, w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (pseen was read at the top of the iteration: the round trip is hidden; a neighbour that
            // is not resident must not hang us: bounded spin)
            // The value stays opaque until HERE: the compiler otherwise evaluates `pseen < w` right
            // behind the load, and that s_waitcnt vmcnt(0) at the top of the iteration also waits
            // for the stream loads and LDS-DMA pieces issued just before it.
            asm volatile("" : "+v"(pseen));
            int spin = 0;
            for (; waiting && pseen < w && spin < 2048; ++spin) {
                __builtin_amdgcn_s_sleep(8);
                pseen = __hip_atomic_load(my_prog + zn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (spin == 2048) waiting = false;
        }
        __syncthreads();
        const unsigned tb = slab_base; slab_base = next_base; next_base = tb;
        const F *td = dl; dl = dl_next; dl_next = td;
    }
    if constexpr (CSUM) {
        // slot h * 16 + (j & 1) * 8 + it of chunk c = position `it` of column 8 h + 2 c + (j & 1):
        // sum over the 8 positions; lanes 0-15 and 32-47 hold the 32 distinct slots (16-31 / 48-63
        // loaded the same ones).  One workgroup row of column sums per (block x): the dense part
        // y = 0 writes them.
        if (active && blockIdx.y == 0) {
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                if (!act[q]) continue;
#pragma unroll
                for (int c = 0; c < LG_CHUNKS; ++c) {
                    F v = csum[q][c];
                    v += __shfl_xor(v, 1, 64);
                    v += __shfl_xor(v, 2, 64);
                    v += __shfl_xor(v, 4, 64);
                    if ((lane & 7) == 0 && (lane & 16) == 0)
                        ws_csum[(int64_t)blockIdx.x * (n_groups * LG_C) + (group + q * NWV) * LG_C +
                                8 * (lane >> 5) + 2 * c + ((lane >> 3) & 1)] = v;
                }
            }
        }
    }
    if (active) {
        // ws layout: [part][block][n_groups * LG_C kernel columns][128]
        const int h = lane >> 5;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            if (!act[q]) continue;
            F *dst = ws + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_groups + group + q * NWV) *
                              (LG_C * LG_W);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int u = 0; u < NRD; ++u) {
                    vec_t o;
#pragma unroll
                    for (int qq = 0; qq < VEC; ++qq) o[qq] = acc[q][j][u][qq];
                    *reinterpret_cast<vec_t *>(dst + (8 * h + j) * LG_W + u * (512 / (int)sizeof(F)) +
                                               (lane & 31) * VEC) = o;
                }
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

// column sums: csum[c] = sum over the workgroups (fixed order) of their partial sums
template <typename F>
__global__ void lg_csum_kernel(const F *__restrict__ part, int nblk, int64_t m, F *__restrict__ csum) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    double a = 0.0;
    for (int b = 0; b < nblk; ++b) a += (double)part[(int64_t)b * m + c];
    csum[c] = (F)a;
}

template <typename F>
static int run_csr_dense_lg(const F *vals, const unsigned *koff, const int64_t *xptr, const F *xvals,
                            const unsigned *xkoff, int64_t n, int64_t m, const F *B, int64_t r,
                            const F *d, int unc, F *out, F *colsum, bool compact, hipStream_t st) {
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
    const bool want_csum = colsum != nullptr;
    if (n_slabs == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        if (want_csum) TM_HIP(hipMemsetAsync(colsum, 0, sizeof(F) * (size_t)m, st));
        return TM_OK;
    }
    int64_t nblk = std::max<int64_t>(1, tune("lg_rounds", 1) * NUM_CU / ((int64_t)n_parts * nz));
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const int64_t stride = m * LG_W;  // per (part, block)
    const size_t tmp_bytes = (sizeof(F) * (size_t)(n_parts * stride) + 255) / 256 * 256;
    const size_t part_bytes = (sizeof(F) * (size_t)((int64_t)n_parts * nblk * stride) + 255) / 256 * 256;
    const size_t csum_bytes = ((want_csum ? sizeof(F) * (size_t)(nblk * m) : 0) + 255) / 256 * 256 + 256;
    // soft lockstep of the column-half workgroups (21.2 -> 18.1 GB of HBM traffic at cfg4, same
    // time); tm_tune_set("lg_lockstep", 0) switches it off
    const int lockstep = (int)tune("lg_lockstep", 1);
    const size_t prog_bytes = (sizeof(int) * (size_t)(n_parts * nblk * nz) + 255) / 256 * 256;
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + part_bytes + csum_bytes + prog_bytes + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    F *ws_csum = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes + part_bytes);
    int *prog = nullptr;
    if (lockstep && nz > 1) {
        prog = reinterpret_cast<int *>(reinterpret_cast<char *>(wsv) + tmp_bytes + part_bytes + csum_bytes);
        TM_HIP(hipMemsetAsync(prog, 0, prog_bytes, st));
    }
    const size_t lds = (size_t)LgLds<F>::TOTAL;
    // (groups with no columns at all leave their slots of the partial sums untouched)
    if (want_csum) TM_HIP(hipMemsetAsync(ws_csum, 0, sizeof(F) * (size_t)(nblk * m), st));
    // (f64 with 4 unconditional positions has no registers left for the column sums: UNC = 2 there)
    auto kern = want_csum ? (unc >= 4 && sizeof(F) == 4 ? &csr_dense_lg_kernel<F, 4, 1, true>
                                                        : &csr_dense_lg_kernel<F, 2, 1, true>)
                          : (unc >= 4 ? &csr_dense_lg_kernel<F, 4, 1, false> : &csr_dense_lg_kernel<F, 2, 1, false>);
    if (compact)       // (vals / koff / xptr carry cvals / cmap / crec)
        kern = want_csum ? (unc >= 4 && sizeof(F) == 4 ? &csr_dense_lg_kernel<F, 4, 1, true, true>
                                                       : &csr_dense_lg_kernel<F, 2, 1, true, true>)
                         : (unc >= 4 ? &csr_dense_lg_kernel<F, 4, 1, false, true>
                                     : &csr_dense_lg_kernel<F, 2, 1, false, true>);
    const int threads = LG_THREADS;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts, (unsigned)nz), dim3(threads),
                       lds, st, vals, koff, xptr, xvals, xkoff, n_groups, n_slabs, spb, B, n, r,
                       (int)nB, d, ws, ws_csum, prog);
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_parts, tmp, n_parts * stride, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((lg_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st,
                       tmp, m, nB, out);
    TM_LAUNCH_CHECK();
    if (want_csum) {
        hipLaunchKernelGGL((lg_csum_kernel<F>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, ws_csum,
                           (int)nblk, m, colsum);
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

}  // namespace tmh

extern "C" {
int tm_lg_rows(void) { return tmh::LG_R; }
int tm_lg_group_cols(void) { return tmh::LG_C; }
int tm_lg_row_bytes(int elem_size) {
    return elem_size == 8 ? tmh::LgLds<double>::RSB : tmh::LgLds<float>::RSB;
}

int tm_csr_dense_sandwich_lg_f32(const float *vals, const uint32_t *koff, const int64_t *xptr,
                                 const float *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                 const float *B, int64_t r, const float *d, int unconditional,
                                 float *out, void *stream) {
    return tmh::run_csr_dense_lg<float>(vals, koff, xptr, xvals, xkoff, n, m, B, r, d, unconditional,
                                        out, nullptr, false, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_lg_f64(const double *vals, const uint32_t *koff, const int64_t *xptr,
                                 const double *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                 const double *B, int64_t r, const double *d, int unconditional,
                                 double *out, void *stream) {
    return tmh::run_csr_dense_lg<double>(vals, koff, xptr, xvals, xkoff, n, m, B, r, d,
                                         unconditional, out, nullptr, false, tmh::as_stream(stream));
}
/* the same pass, additionally colsum (length m, kernel column order) = A' d */
int tm_csr_dense_sandwich_lg_xtd_f32(const float *vals, const uint32_t *koff, const int64_t *xptr,
                                     const float *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                     const float *B, int64_t r, const float *d, int unconditional,
                                     float *out, float *colsum, void *stream) {
    if (!colsum) {
        tmh::set_error("tm_csr_dense_sandwich_lg_xtd: colsum is NULL");
        return TM_EINVAL;
    }
    return tmh::run_csr_dense_lg<float>(vals, koff, xptr, xvals, xkoff, n, m, B, r, d, unconditional,
                                        out, colsum, false, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_lg_xtd_f64(const double *vals, const uint32_t *koff, const int64_t *xptr,
                                     const double *xvals, const uint32_t *xkoff, int64_t n, int64_t m,
                                     const double *B, int64_t r, const double *d, int unconditional,
                                     double *out, double *colsum, void *stream) {
    if (!colsum) {
        tmh::set_error("tm_csr_dense_sandwich_lg_xtd: colsum is NULL");
        return TM_EINVAL;
    }
    return tmh::run_csr_dense_lg<double>(vals, koff, xptr, xvals, xkoff, n, m, B, r, d,
                                         unconditional, out, colsum, false, tmh::as_stream(stream));
}


/* the compact stream (round 3): cvals = the values of the real slots only, cmap = uint8 [block][32 slots][4
 * chunks] (1 + row in slab, 0 = padding) read as one dword per lane, crec = int64 [block][2] {index of the
 * block's first value in cvals, overflow entries | first overflow entry << 32}; xkoff as above.
 * colsum may be NULL. */
int tm_csr_dense_sandwich_lgc_f32(const float *cvals, const uint32_t *cmap, const int64_t *crec,
                                  const uint32_t *xkoff, int64_t n, int64_t m, const float *B, int64_t r,
                                  const float *d, int unconditional, float *out, float *colsum, void *stream) {
    return tmh::run_csr_dense_lg<float>(cvals, cmap, crec, nullptr, xkoff, n, m, B, r, d, unconditional, out,
                                        colsum, true, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_lgc_f64(const double *cvals, const uint32_t *cmap, const int64_t *crec,
                                  const uint32_t *xkoff, int64_t n, int64_t m, const double *B, int64_t r,
                                  const double *d, int unconditional, double *out, double *colsum,
                                  void *stream) {
    return tmh::run_csr_dense_lg<double>(cvals, cmap, crec, nullptr, xkoff, n, m, B, r, d, unconditional, out,
                                         colsum, true, tmh::as_stream(stream));
}

}  // extern "C"
