// K3 (entry-list form, round 4): sparse x dense sandwich  out = A^T diag(d) B  for a C-ordered dense
// operand with more than 64 columns (reference: ext/sparse.pyx:211-260 csr_dense_sandwich ->
// ext/sparse_helpers-tmpl.cpp:23-146 _csr_denseC_sandwich).
//
// Why another form.  The lane-group kernel (sparse_lg.hip) keeps the accumulators of a sparse column in
// STATIC registers, so it has to walk its stream column by column: two columns side by side in the two
// halves of the wave, padded to the longer one (1.5 LDS reads per nonzero), two positions per scalar bit
// test -- 19 short dependent LDS round trips per wave and slab, LDS pipe 53 % busy, 56 % of the wave
// cycles parked (profiles/r3_sq_counters.txt); removing its HBM waste (compact stream, round 3) did not
// move its time.  Here the accumulator of a nonzero is picked AT RUN TIME with the VGPR index mode of
// gfx9 (s_set_gpr_idx_on: M0[7:0] is added to the register number of the enabled operands of the
// following VALU instructions), so the stream is a plain list of ENTRIES {value, row in slab, column in
// group} in batches of 16 -- no padding inside a batch, no branch, one LDS read per nonzero, EN_NSLOT
// reads in flight per wave:
//     v_add_u32_dpp   row offset of entry i (row_newbcast inside every row of 16 lanes) + lane offset
//     ds_read_b128    the whole wave reads the 1 KiB row of B: lane <-> 2 of the 128 dense columns
//     v_readlane_b32  4 * column  ->  SGPR
//     s_set_gpr_idx_on / 2 x v_fmac_f64_dpp (value by row_newbcast) / s_set_gpr_idx_off
// scripts/ubench/gather_idx.hip: 6.8 cycles per entry and CU with 16 waves (the lane-group kernel runs at
// 13.2 at BASELINE configs[3]; the LDS floor of one ds_read_b128 per nonzero is 4.4).
//
// Stream ("entry twin", built once per block by SlabEnt.from_csr): rows in slabs of EN_R = 64, columns
// dealt to groups of EN_C = 16 = one wave.  The entries of a (group, slab) block are padded to whole
// BATCHES of 16 slots (padding: value 0, the row of the block's first entry); the blocks of a group follow
// one another, slab after slab, so a wave walks ONE contiguous stream:
//     vals F[T]              value
//     meta uint16[T]         (slab & 63) << 10 | row in slab << 4 | column in group   (round 6; uint32 row << 4 | column
//                            before: 12 -> 10 bytes per slot.  An empty block at every 32nd slab holds one padding batch,
//                            so that two batches of a group are never 64 slabs apart and the 6-bit tag is unambiguous.)
//     bstart uint32[G][S+1]  first batch of block (group, slab); entry S = the end of the group's stream
// 10 bytes per slot.  A wave takes its stream in SUPERBATCHES of 64 slots, lane <-> slot: one coalesced load
// of the values and one of the meta words a whole superbatch (~5000 cycles) ahead, the d of every slot's row
// gathered from global memory (L2-resident lines) half a superbatch ahead; then, still lane <-> slot, the
// FOLD: a = value * d, kq = LDS address of the row of B (the all-zero row if value == 0 or d == 0, so inf * 0
// of an excluded row cannot leak), jv = accumulator offset of the column -- 64 slots per instruction -- and
// {a, kq, jv | slab << 8} goes to a 1 KiB per-wave LDS scratch, from where each batch fetches its 16 slots
// with one ds_read_b128 (all four rows of 16 lanes read the same 16 addresses: the broadcast the DPP
// row_newbcast operands need).  Slab boundaries come from the entries themselves.  (A first version walked
// per-slab unit counts with a scalar state machine and folded batch by batch: ~150 wave instructions of
// skeleton per batch of 16, as long as the batch itself -- profiles/r4_k3_ent.txt.)
//
// The accumulators (v[64:127]) and the landing registers of the LDS reads (below v64) are pinned with
// physical-register constraints: the index mode needs a base register known when the code is written.
#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr int EN_R = 64;            // rows per slab
constexpr int EN_C = 16;            // sparse columns per wave
constexpr int EN_NW = 16;           // waves per workgroup (256 sparse columns)
constexpr int EN_THREADS = EN_NW * 64;
constexpr int EN_W = 128;           // dense columns per part
constexpr int EN_B = 16;            // slots per batch
constexpr int EN_SB = 64;           // slots per superbatch (lane <-> slot)
// (four LDS reads in flight per wave: landing slots v[48 : 63]; eight measured no faster)

template <typename F>
struct EnLds {
    static constexpr int ROWB = EN_W * (int)sizeof(F);      // bytes of a slab row (1024 / 512)
    static constexpr int SLABB = EN_R * ROWB;
    // [zero row][buffer 0: EN_R rows][buffer 1: EN_R rows]; row r (0-based) of buffer b at
    // b * SLABB + (1 + r) * ROWB
    static constexpr int SCR_OFF = ROWB + 2 * SLABB;         // per-wave scratch: 64 folded slots of 16 bytes
    static constexpr int CS_OFF = SCR_OFF + EN_NW * EN_SB * 16;   // column sums of v * d (CSUM), doubles
    static constexpr int TOTAL = CS_OFF + EN_NW * EN_C * (int)sizeof(double);
};

// Registers the compiler may use: v0 .. v[EN_CVGPR - 1] (amdgpu_num_vgpr); everything above belongs to the
// inline asm below: landing slots of the LDS reads just under v64, accumulators v[64:127] (f64: column j =
// v[64 + 4 j : 64 + 4 j + 3]; f32: v[64 + 2 j : 64 + 2 j + 1]).  The index mode needs a base register
// known when the code is written, and values the register allocator does not know about cannot be moved
// or spilled by it (a first version passed the tuples as "+{v[64:95]}" operands: the allocator parked
// them elsewhere between the asm statements and re-loaded all 64 from scratch in every batch).
// Register map: compiler v0 .. v23; LDS addresses of a batch v[24 : 39]; stream / d loads v[40 : 45] (en_take
// hands them to the compiler behind the wait); landing slots of the LDS reads v[48 : 63] (f32: v[56 : 63]);
// accumulators from v64 up.  scripts/gen_ent_batch.py writes the batch code against the same map.
constexpr int EN_CVGPR = 24;
#define EN_LD "40"

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t en_rsrc_t;
__device__ __forceinline__ en_rsrc_t en_rsrc(const void *base, int64_t bytes) {
    const unsigned nb = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)nb, 0x00020000);
}
__device__ __forceinline__ void en_buf_to_lds16(en_rsrc_t rs, void *lds, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff,
                                             soff, 0, 0);
}
// all accumulators = 0; the clobber of the highest register is what sizes the wave's register allocation
// (.amdhsa_next_free_vgpr): the compiler itself stays below EN_CVGPR
template <typename F>
__device__ __forceinline__ void en_zero_acc() {
    if constexpr (sizeof(F) == 8) {
        static_for<16>([&](auto jc) {
            asm volatile("v_mov_b32 v[64+4*%0], 0\n\tv_mov_b32 v[64+4*%0+1], 0\n\t"
                         "v_mov_b32 v[64+4*%0+2], 0\n\tv_mov_b32 v[64+4*%0+3], 0"
                         :: "n"(decltype(jc)::value) : "v127");
        });
    } else {
        static_for<16>([&](auto jc) {
            asm volatile("v_mov_b32 v[64+2*%0], 0\n\tv_mov_b32 v[64+2*%0+1], 0" :: "n"(decltype(jc)::value) : "v95");
        });
    }
}
// accumulator register K (0 .. 63 / 0 .. 31) of this lane
template <int K>
__device__ __forceinline__ unsigned en_read_acc() {
    unsigned x;
    asm volatile("v_mov_b32 %0, v[64+%1]" : "=v"(x) : "n"(K));
    return x;
}
// Stream / d loads land in RESERVED registers (value v[LD : LD+1], meta v[LD+2], d v[LD+4 : LD+5]) and are
// handed to the compiler only behind the wait, inside ONE asm statement.  (With the destinations as ordinary
// "=v" outputs and the wait as a separate asm with "+v" operands the compiler is free to copy the
// still-in-flight registers BEFORE the wait -- it did, in one of the three wait variants of the f32 kernel.)
template <typename F>
__device__ __forceinline__ void en_load_stream(unsigned voff_v, const F *vp, unsigned voff_m, const unsigned short *mp) {
    // (the meta word is 16 bits since round 6: global_load_ushort zero-extends into the register)
    if constexpr (sizeof(F) == 8)
        asm volatile("global_load_dwordx2 v[" EN_LD ":" EN_LD "+1], %0, %1\n\t"
                     "global_load_ushort v[" EN_LD "+2], %2, %3" :: "v"(voff_v), "s"(vp), "v"(voff_m), "s"(mp) : "memory");
    else
        asm volatile("global_load_dword v[" EN_LD "], %0, %1\n\t"
                     "global_load_ushort v[" EN_LD "+2], %2, %3" :: "v"(voff_v), "s"(vp), "v"(voff_m), "s"(mp) : "memory");
}
template <typename F>
__device__ __forceinline__ void en_load_d(unsigned voff, const F *dp) {
    if constexpr (sizeof(F) == 8)
        asm volatile("global_load_dwordx2 v[" EN_LD "+4:" EN_LD "+5], %0, %1" :: "v"(voff), "s"(dp) : "memory");
    else
        asm volatile("global_load_dword v[" EN_LD "+4], %0, %1" :: "v"(voff), "s"(dp) : "memory");
}
// wait until at most N of the most recently issued vector-memory operations are outstanding (they complete
// in order), then take the stream registers and the d register
template <int N>
__device__ __forceinline__ void en_take(double &v, unsigned &m, double &dd) {
    unsigned vl, vh, dl, dh;
    asm volatile("s_waitcnt vmcnt(%5)\n\t"
                 "v_mov_b32 %0, v[" EN_LD "]\n\tv_mov_b32 %1, v[" EN_LD "+1]\n\tv_mov_b32 %2, v[" EN_LD "+2]\n\t"
                 "v_mov_b32 %3, v[" EN_LD "+4]\n\tv_mov_b32 %4, v[" EN_LD "+5]"
                 : "=v"(vl), "=v"(vh), "=v"(m), "=v"(dl), "=v"(dh) : "n"(N) : "memory");
    v = __hiloint2double((int)vh, (int)vl);
    dd = __hiloint2double((int)dh, (int)dl);
}
template <int N>
__device__ __forceinline__ void en_take(float &v, unsigned &m, float &dd) {
    unsigned vl, dl;
    asm volatile("s_waitcnt vmcnt(%3)\n\t"
                 "v_mov_b32 %0, v[" EN_LD "]\n\tv_mov_b32 %1, v[" EN_LD "+2]\n\tv_mov_b32 %2, v[" EN_LD "+4]"
                 : "=v"(vl), "=v"(m), "=v"(dl) : "n"(N) : "memory");
    v = __uint_as_float(vl);
    dd = __uint_as_float(dl);
}
#else
struct en_rsrc_t {};
__device__ inline en_rsrc_t en_rsrc(const void *, int64_t) { return {}; }
__device__ inline void en_buf_to_lds16(en_rsrc_t, void *, int, int) {}
template <typename F> __device__ void en_zero_acc() {}
template <int K> __device__ unsigned en_read_acc() { return 0u; }
template <typename F> __device__ void en_load_stream(unsigned, const F *, unsigned, const unsigned short *) {}
template <typename F> __device__ void en_load_d(unsigned, const F *) {}
template <int N, typename F> __device__ void en_take(F &, unsigned &, F &) {}
#endif

// One batch of N = 4 / 8 / 12 / 16 entries as ONE asm statement (generated: scripts/gen_ent_batch.py): a = value * d
// and kq = LDS address of the row of B (the zero row for padding / d == 0) with entry i in lane i of every row
// of 16 lanes; p0 .. p3 = the accumulator register offsets of the entries of quad 0 .. 3, one byte each.  Per entry:
// s_waitcnt / s_set_gpr_idx_idx / 2 x v_fmac_dpp (value by row_newbcast, accumulator by the index) / the LDS read
// of the entry four further on; the N LDS addresses (v_add_u32_dpp) are formed before the index mode is switched
// on, because while it is on every vector-ALU instruction is indexed.
template <int N, typename F>
__device__ __forceinline__ void en_batch_asm(unsigned &p0, unsigned &p1, unsigned &p2, unsigned &p3, F a, unsigned kq,
                                             unsigned lane_off);
#if defined(__HIP_DEVICE_COMPILE__)
#include "sparse_ent_batch.inc"
#else
template <int N, typename F>
__device__ void en_batch_asm(unsigned &, unsigned &, unsigned &, unsigned &, F, unsigned, unsigned) {}
#endif

// CSUM = true: the column sums A^T d (length m, kernel column order) come out of the same pass
// (StandardizedMatrix.sandwich: reference standardized_mat.py:149-150 calls transpose_matvec): lanes 0-15
// add their entry's value * d to a per-wave LDS array of 16 doubles, one ds_add_f64 per batch.
template <typename F, bool CSUM>
__global__ __launch_bounds__(EN_THREADS) __attribute__((amdgpu_num_vgpr(EN_CVGPR))) void csr_dense_ent_kernel(
    const F *__restrict__ vals, const unsigned short *__restrict__ meta, const unsigned *__restrict__ bstart,
    int n_groups, int64_t n_slabs, int64_t slabs_per_block, const F *__restrict__ B, int64_t n, int64_t r,
    int nB, const F *__restrict__ d, F *__restrict__ ws, F *__restrict__ ws_csum, long long *__restrict__ prof) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using L = EnLds<F>;
    constexpr int ROWB = L::ROWB;
    constexpr int SLABB = L::SLABB;
    constexpr int NV = SLABB / 16 / EN_THREADS;            // 1 KiB pieces copied per wave (4 / 2)
    constexpr int RPP = 1024 / ROWB;                       // slab rows per piece (1 / 2)
    constexpr int RSH = sizeof(F) == 8 ? 10 : 9;           // log2(ROWB)
    constexpr int JSH = sizeof(F) == 8 ? 2 : 1;            // accumulator registers per column: 4 / 2
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int l16 = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.z * EN_NW + wave;
    const bool active = group < n_groups;
    const int j0 = blockIdx.y * EN_W;
    const int64_t s0 = (int64_t)blockIdx.x * slabs_per_block;
    const int ns = (int)(min(s0 + slabs_per_block, n_slabs) - s0);   // slabs of this workgroup
    const unsigned lane_off = (unsigned)lane * 16u / (sizeof(F) == 8 ? 1u : 2u);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_byte *)smem_raw;
    u4 *scratch = reinterpret_cast<u4 *>(smem_raw + L::SCR_OFF) + wave * EN_SB;
    double *cs_lds = reinterpret_cast<double *>(smem_raw + L::CS_OFF) + wave * EN_C;

    en_zero_acc<F>();
    // the zero row (and the column-sum slots)
    for (int i = tid; i < ROWB / 4; i += EN_THREADS) reinterpret_cast<unsigned *>(smem_raw)[i] = 0u;
    if (CSUM && lane < EN_C) cs_lds[lane] = 0.0;
    if (ns <= 0) return;

    // ---- slab copy (LDS-DMA, as in sparse_lg.hip): piece i of this wave = 1 KiB = RPP slab rows; B is read
    // through a buffer descriptor rebuilt per slab (base = the slab's first byte, range = what is left of the
    // array): rows beyond n - 1 in the ragged last slab read as 0 without per-lane clamping.
    constexpr int VEC = 16 / (int)sizeof(F);
    const int cc = min(j0 + ((lane * 16) % ROWB) / (int)sizeof(F), nB - VEC);
    const int row0 = wave * NV * RPP + (lane * 16) / ROWB;
    const unsigned boff0 = (unsigned)((row0 * r + cc) * (int64_t)sizeof(F));
    const int64_t pstride = (int64_t)RPP * r * (int64_t)sizeof(F);
    const int64_t bstride = (int64_t)EN_R * r * (int64_t)sizeof(F);
    const char *bnext = reinterpret_cast<const char *>(B) + s0 * bstride;        // slab to copy next
    int64_t bleft = n * r * (int64_t)sizeof(F) - s0 * bstride;                     // bytes from there on
    auto issue_pieces = [&](int buf, int lo, int hi) {
#if defined(EN_ABL_NOCOPY)            // timing only: the slab of B is never refreshed
        if (bnext != reinterpret_cast<const char *>(B) + s0 * bstride) return;
#endif
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (i >= lo && i < hi)
            en_buf_to_lds16(en_rsrc(bnext, bleft), smem_raw + ROWB + buf * SLABB + (wave * NV + i) * RPP * ROWB,
                            (int)boff0, (int)(i * pstride));
    };

    // ---- the wave's stream: batches [b0, b0 + nbw) of its group ----
    const unsigned *brow = bstart + (int64_t)(active ? group : 0) * (n_slabs + 1);
    const unsigned b0 = active ? brow[s0] : 0u;
    const int nbw = active ? (int)(brow[s0 + ns] - b0) : 0;
    // superbatch k = slots (b0 + 4 k) * 16 + lane; registers: X {vx, mx} = the superbatch being loaded,
    // Y {vy, my} = the one that has arrived (folded next), dy = the d of Y's rows (being gathered)
    const F *vbase = vals + (int64_t)b0 * EN_B;
    const unsigned short *mbase = meta + (int64_t)b0 * EN_B;
    F vx = F(0), vy = F(0), dy = F(0);
    unsigned mx = 0u, my = 0u;
    const unsigned lane_v = (unsigned)lane * (unsigned)sizeof(F), lane_m = (unsigned)lane * 2u;
    // (the loads are written as asm so that THIS code places the waits: the compiler would wait with
    // vmcnt(0) at every use, i.e. also for the copy pieces of the next slab issued a moment before)
    auto request_stream = [&](int k) {
        en_load_stream<F>(lane_v, vbase + (int64_t)k * EN_SB, lane_m, mbase + (int64_t)k * EN_SB);
    };
    // The stream's meta word is 16 bits {slab & 63, row in slab, column in group}; the slot's slab is rebuilt from the
    // 6-bit tag and a wave-uniform running slab (the stream is ordered by slab and the builder leaves no gap of 64 slabs
    // between two batches of a group): my = row << 4 | column as before round 6, row = slab * 64 + row in slab.
    const unsigned s0u = (unsigned)s0;
    unsigned cur_slab = s0u;
    auto expand = [&](unsigned m16) -> unsigned {
        const unsigned slab = cur_slab + (((m16 >> 10) - cur_slab) & 63u);
        cur_slab = (unsigned)__builtin_amdgcn_readlane((int)slab, 63);
        return (slab << 10) | (m16 & 0x3ffu);
    };
    const unsigned row_max = (unsigned)(n - 1);
    auto request_d = [&]() {          // d of Y's rows (row < 2^28: a 32-bit byte offset); slots behind the wave's stream
        // belong to another group (or the slack): their rebuilt row means nothing and is clamped into the array
        en_load_d<F>(min(my >> 4, row_max) * (unsigned)sizeof(F), d);
    };
    auto wait_loads = [&](auto nc) { en_take<decltype(nc)::value>(vx, mx, dy); };
    int psince = 0;                    // copy pieces issued since the last requests
    int lsince = 0;                    // stream / d loads requested since the last copy pieces
    (void)lsince;
    // boundary k: fold superbatch k (= Y, dy) into the scratch, Y <- X = superbatch k + 1, requests
    auto boundary = [&](int k) {
        // (everything requested before the last `psince` pieces has arrived once at most that many operations
        // are outstanding; a smaller count only waits longer)
        if (psince == 0) wait_loads(std::integral_constant<int, 0>{});
        else if (psince < NV) wait_loads(std::integral_constant<int, NV / 2>{});
        else if (psince < 2 * NV) wait_loads(std::integral_constant<int, NV>{});
        else wait_loads(std::integral_constant<int, 2 * NV>{});
        const unsigned rowg = my >> 4;
        const unsigned srel = (rowg >> 6) - s0u;
        const bool ok = vy != F(0) && dy != F(0);
        const F a = ok ? vy * dy : F(0);
        const unsigned kq = ok ? lds_base + (srel & 1u) * (unsigned)SLABB + (((rowg & 63u) + 1u) << RSH) : lds_base;
        // accumulator offsets of a quad of slots packed into the dword of its first lane (byte q = slot 4 k + q);
        // the other lanes of the quad carry the slab -- a batch reads both back with five v_readlane
        const unsigned jv = (my & 15u) << JSH;
        const unsigned j1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)jv, 0x101, 0xf, 0xf, true);   // row_shl:1
        const unsigned j2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)jv, 0x102, 0xf, 0xf, true);
        const unsigned j3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)jv, 0x103, 0xf, 0xf, true);
        const unsigned jvs = (lane & 3) == 0 ? (jv | (j1 << 8) | (j2 << 16) | (j3 << 24)) : srel;
        u4 e;
        if constexpr (sizeof(F) == 8) {
            e = u4{(unsigned)__double2loint(a), (unsigned)__double2hiint(a), kq, jvs};
        } else {
            e = u4{__float_as_uint(a), 0u, kq, jvs};
        }
        scratch[lane] = e;
        if constexpr (CSUM) {
            if (4 * k * EN_B + lane < nbw * EN_B) atomic_add(cs_lds + (jv >> JSH), (double)a);
        }
        vy = vx;
        my = expand(mx);
        request_d();
        request_stream(k + 2);
        psince = 0;
        lsince += 3;
    };
    // prologue: slab 0 of the range into buffer 0; superbatches 0 (arrived) and 1 (requested), d of 0
    issue_pieces(0, 0, NV);
    bnext += bstride;
    bleft -= bstride;
    request_stream(0);
    wait_loads(std::integral_constant<int, 0>{});
    vy = vx;
    my = expand(mx);
    request_d();
    request_stream(1);
    __syncthreads();

#if defined(EN_PROF)                  // cycles per section (tm_tune_set("ent_prof", device pointer))
    long long pt_a = 0, pt_p = 0, pt_x = 0, pt_b = 0, pt_bar = 0, pt_n = 0;
    long long pt0 = clock64();
    const long long pt_start = pt0;
#define EN_TICK(acc) { const long long t_ = clock64(); acc += t_ - pt0; pt0 = t_; }
#else
#define EN_TICK(acc)
#endif
    int w = 0;                         // slab being worked on (relative)
    int ncopied = 0;                   // pieces of slab w + 1 issued so far
    auto end_slab = [&]() {
        const bool more = w + 1 < ns;
        if (more) {
            if (ncopied < NV) {
                issue_pieces((w & 1) ^ 1, ncopied, NV);
                psince += NV - ncopied;
                lsince = 0;
            }
            bnext += bstride;
            bleft -= bstride;
        }
        EN_TICK(pt_p)
        // the copy pieces of the next slab must have landed; the stream / d requests issued AFTER them may
        // stay in flight (a plain __syncthreads() waits with vmcnt(0): for a request of a moment ago that is
        // a full HBM round trip in front of the barrier, once per slab on average)
#if defined(EN_ABL_NOBARRIER)         // timing only (wrong results): the waves of a workgroup run free
#define EN_BAR ""
#else
#define EN_BAR "\n\ts_barrier"
#endif
#if defined(__HIP_DEVICE_COMPILE__)
        if (lsince == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" EN_BAR ::: "memory");
        else if (lsince == 3) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" EN_BAR ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" EN_BAR ::: "memory");
#endif
        EN_TICK(pt_bar)
        ncopied = 0;
        ++w;
    };
    u4 enext = u4{0u, 0u, 0u, 0u};
    for (int b = 0; b < nbw; ++b) {
        const int q = b & 3;
        u4 e;
        if (q == 0) {
            boundary(b >> 2);
            e = scratch[l16];
        } else {
            e = enext;
        }
        // (the folded slots of the next batch are fetched behind this batch's LDS reads; batch 0 of a
        // superbatch has to wait for the fold)
        if (q != 3) enext = scratch[(q + 1) * EN_B + l16];
        EN_TICK(pt_a)
        const int slab = __builtin_amdgcn_readlane((int)e[3], 1);
        while (w < slab) end_slab();
        // the copy of the next slab starts in front of the slab's first batch (with the second / third batch, or
        // half and half, measured no different: profiles/r4_k3_ent.txt)
        if (ncopied == 0 && w + 1 < ns) {
            issue_pieces((w & 1) ^ 1, 0, NV);

            ncopied = NV;
            psince += NV;
            lsince = 0;
        }
        EN_TICK(pt_p)
        F a;
        if constexpr (sizeof(F) == 8) a = __hiloint2double((int)e[1], (int)e[0]);
        else a = __uint_as_float(e[0]);
#if defined(EN_ABL_NOBATCH)           // timing only: the memory side alone
        asm volatile("" ::"v"(a), "v"(e[2]), "v"(e[3]));
#else
        // Slots behind the batch's last one with a != 0 add nothing (the padding of a block sits at its end:
        // 7.5 of 59 slots per block at BASELINE configs[3]): the batch is worked on in quads of 4 slots.
        const unsigned live = (unsigned)__builtin_amdgcn_ballot_w64(a != F(0)) & 0xffffu;
        const int nq = live ? ((31 - __builtin_clz(live)) >> 2) + 1 : 0;
        unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)e[3], 0);
        unsigned p1 = (unsigned)__builtin_amdgcn_readlane((int)e[3], 4);
        unsigned p2 = (unsigned)__builtin_amdgcn_readlane((int)e[3], 8);
        unsigned p3 = (unsigned)__builtin_amdgcn_readlane((int)e[3], 12);
        if (nq == 4) en_batch_asm<16, F>(p0, p1, p2, p3, a, e[2], lane_off);
        else if (nq == 3) en_batch_asm<12, F>(p0, p1, p2, p3, a, e[2], lane_off);
        else if (nq == 2) en_batch_asm<8, F>(p0, p1, p2, p3, a, e[2], lane_off);
        else if (nq == 1) en_batch_asm<4, F>(p0, p1, p2, p3, a, e[2], lane_off);
#endif
        EN_TICK(pt_x)
#if defined(EN_PROF)
        ++pt_n;
#endif
    }
    while (w < ns) end_slab();
    // (requests still in flight must land before the wave ends: their registers are this wave's)
    wait_loads(std::integral_constant<int, 0>{});
#if defined(EN_PROF)
    if (prof != nullptr && lane == 0) {
        long long *o = prof + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (EN_NW * 8) + wave * 8;
        o[0] = pt_a; o[1] = pt_p; o[2] = pt_x; o[3] = pt_b; o[4] = pt_bar; o[5] = pt_n; o[6] = clock64() - pt_start; o[7] = ns;
    }
#endif

    if (active) {
        // ws layout: [part][block][n_groups * EN_C kernel columns][128]
        F *dst = ws + (((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * n_groups + group) * (EN_C * EN_W);
        static_for<EN_C>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            if constexpr (sizeof(F) == 8) {
                const u4 o = u4{en_read_acc<4 * j>(), en_read_acc<4 * j + 1>(), en_read_acc<4 * j + 2>(),
                                en_read_acc<4 * j + 3>()};
                *reinterpret_cast<u4 *>(dst + j * EN_W + 2 * lane) = o;
            } else {
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 o = u2{en_read_acc<2 * j>(), en_read_acc<2 * j + 1>()};
                *reinterpret_cast<u2 *>(dst + j * EN_W + 2 * lane) = o;
            }
        });
        if constexpr (CSUM) {
            if (blockIdx.y == 0 && lane < EN_C)
                ws_csum[(int64_t)blockIdx.x * (n_groups * EN_C) + group * EN_C + lane] = (F)cs_lds[lane];
        }
    }
}

// tmp [part][m][128] -> out[m][nB]
template <typename F>
__global__ void en_untile_kernel(const F *__restrict__ tmp, int64_t m, int64_t nB, F *__restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * nB) return;
    const int64_t i = e / nB, j = e % nB;
    out[e] = tmp[((j / EN_W) * m + i) * EN_W + (j % EN_W)];
}

// column sums: csum[c] = sum over the workgroups (fixed order) of their partial sums
template <typename F>
__global__ void en_csum_kernel(const F *__restrict__ part, int nblk, int64_t m, F *__restrict__ csum) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m) return;
    double a = 0.0;
    for (int b = 0; b < nblk; ++b) a += (double)part[(int64_t)b * m + c];
    csum[c] = (F)a;
}

template <typename F>
static int run_csr_dense_ent(const F *vals, const uint16_t *meta, const uint32_t *bstart, int64_t n, int64_t m,
                             const F *B, int64_t r, const F *d, F *out, F *colsum, hipStream_t st) {
    const int64_t nB = r;
    const int64_t total = m * nB;
    if (total == 0) return TM_OK;
    constexpr int VEC = 16 / (int)sizeof(F);
    if ((reinterpret_cast<uintptr_t>(B) & 15) != 0 || r % VEC != 0 || nB < VEC) {
        set_error("tm_csr_dense_sandwich_ent: B must be C-ordered with 16-byte aligned rows");
        return TM_EUNSUPPORTED;
    }
    if (m % EN_C != 0) {
        set_error("tm_csr_dense_sandwich_ent: m must be a multiple of tm_ent_group_cols()");
        return TM_EINVAL;
    }
    if (n >= (1ll << 28)) {      // meta = row << 4 | column
        set_error("tm_csr_dense_sandwich_ent: at most 2^28 - 1 rows per block (work in row parts)");
        return TM_EUNSUPPORTED;
    }
    const int64_t n_slabs = ceil_div(n, EN_R);
    const int n_groups = (int)(m / EN_C);
    const int n_parts = (int)ceil_div(nB, EN_W);
    const int nz = (int)ceil_div(n_groups, EN_NW);
    const bool want_csum = colsum != nullptr;
    if (n_slabs == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(F) * (size_t)total, st));
        if (want_csum) TM_HIP(hipMemsetAsync(colsum, 0, sizeof(F) * (size_t)m, st));
        return TM_OK;
    }
    int64_t nblk = std::max<int64_t>(1, tune("ent_rounds", 1) * NUM_CU / ((int64_t)n_parts * nz));
    nblk = std::min<int64_t>(nblk, n_slabs);
    const int64_t spb = ceil_div(n_slabs, nblk);
    nblk = ceil_div(n_slabs, spb);
    const int64_t stride = m * EN_W;  // per (part, block)
    const size_t tmp_bytes = (sizeof(F) * (size_t)(n_parts * stride) + 255) / 256 * 256;
    const size_t part_bytes = (sizeof(F) * (size_t)((int64_t)n_parts * nblk * stride) + 255) / 256 * 256;
    const size_t csum_bytes = ((want_csum ? sizeof(F) * (size_t)(nblk * m) : 0) + 255) / 256 * 256 + 256;
    void *wsv = nullptr;
    int rc = get_workspace(tmp_bytes + part_bytes + csum_bytes + 256, &wsv, st);
    if (rc) return rc;
    F *tmp = reinterpret_cast<F *>(wsv);
    F *ws = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes);
    F *ws_csum = reinterpret_cast<F *>(reinterpret_cast<char *>(wsv) + tmp_bytes + part_bytes);
    const size_t lds = (size_t)EnLds<F>::TOTAL;
    auto kern = want_csum ? &csr_dense_ent_kernel<F, true> : &csr_dense_ent_kernel<F, false>;
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    prof_begin(st);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)n_parts, (unsigned)nz), dim3(EN_THREADS), lds, st,
                       vals, meta, bstart, n_groups, n_slabs, spb, B, n, r, (int)nB, d, ws, ws_csum,
                       reinterpret_cast<long long *>((uintptr_t)tune("ent_prof", 0)));
    prof_end(st);
    TM_LAUNCH_CHECK();
    rc = launch_reduce_partials<F>(ws, stride, (int)nblk, n_parts, tmp, n_parts * stride, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL((en_untile_kernel<F>), dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, tmp, m,
                       nB, out);
    TM_LAUNCH_CHECK();
    if (want_csum) {
        hipLaunchKernelGGL((en_csum_kernel<F>), dim3((unsigned)ceil_div(m, 256)), dim3(256), 0, st, ws_csum,
                           (int)nblk, m, colsum);
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

}  // namespace tmh

extern "C" {
int tm_ent_rows(void) { return tmh::EN_R; }
int tm_ent_group_cols(void) { return tmh::EN_C; }
int tm_ent_batch_slots(void) { return tmh::EN_B; }

int tm_csr_dense_sandwich_ent_f32(const float *vals, const uint16_t *meta, const uint32_t *bstart, int64_t n,
                                  int64_t m, const float *B, int64_t r, const float *d, float *out,
                                  float *colsum, void *stream) {
    return tmh::run_csr_dense_ent<float>(vals, meta, bstart, n, m, B, r, d, out, colsum, tmh::as_stream(stream));
}
int tm_csr_dense_sandwich_ent_f64(const double *vals, const uint16_t *meta, const uint32_t *bstart, int64_t n,
                                  int64_t m, const double *B, int64_t r, const double *d, double *out,
                                  double *colsum, void *stream) {
    return tmh::run_csr_dense_ent<double>(vals, meta, bstart, n, m, B, r, d, out, colsum, tmh::as_stream(stream));
}
}  // extern "C"
