// K1d  Dense self sandwich  out = X' diag(d) X  for a C-ordered FLOAT32 block of 129 .. 256 columns on the
// bf16 matrix cores (reference: ext/dense_helpers-tmpl.cpp:266-311; BASELINE configs[1]: 10M x 256).
//
// v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate on gfx950 (157 TF; MI355X_MICROARCH.md), so the
// f32 syrk of dense.hip is compute-bound at 5.8 ms for 10M x 256 (0.72 of that rate) while the block
// itself streams in 1.3 ms.  The bf16 MFMA runs 16x faster.  Every element of the scaled block
// Y = diag(sqrt|d|) X is therefore split into THREE bf16 pieces,
//     y = h + m + l,   h = bf16(y),  m = bf16(y - h),  l = bf16(y - h - m)     (3 x 8 = 24 mantissa bits),
// and  Y' S Y  (S = diag(sign d)) is accumulated in f32 from the six piece products that matter,
//     h h' + h m' + m h' + h l' + l h' + m m'          (dropped: m l', l m', l l' <= 2^-24 relative),
// with v_mfma_f32_16x16x32_bf16.  The result has the accuracy of an f32 product accumulated in f32
// (measured against the f64 oracle in tests/test_gpu_syrk_bf16.py, next to the f32-MFMA path).
// Negative weights cost nothing: the A operand of row r is the B operand with the bf16 sign bits of
// that row flipped (one v_xor per fragment register), so no branch on the sign of d is needed.
//
// Layout (second version, round 3): a workgroup of 8 waves (two per SIMD) streams 32-row chunks (K = 32 of the
// MFMA).  The raw f32 rows come in by LDS-DMA (buffer_load ... lds, no registers) into a ring of THREE raw
// slots, three chunks ahead.  Per chunk a lane converts two units of 2 columns x 4 rows: 4 ds_read_b64 of the
// raw rows (lanes of a half wave = 4 row quads x 8 column pairs: conflict-free), scale by sqrt|d|, split
// into three bf16 pieces, 2 x 3 ds_write_b64 into the planes [piece][column][32 rows = 64 bytes] whose four
// 16-byte row groups are stored at (group ^ (column >> 1)) & 3 -- fragment reads (ds_read_b128) are then
// free of bank conflicts (4 instead of 8 LDS cycles with the padded stride of the first version, whose
// raw reads were 4-way and whose plane writes 4-way conflicted: ~4100 LDS cycles per chunk next to 3500
// cycles of MFMA).  The planes are single-buffered: the fragments of a chunk are read into registers, the
// first two piece products (l h', h l': pieces 0 and 2) start while piece 1 is still being read, then
// barrier B frees the planes and the conversion of the next chunk is issued in four slices behind the
// remaining product groups.  The 136 lower-triangular 16 x 16 tiles are dealt as 4 x 4 patches of
// column-block groups G0 .. G3: waves 0 .. 5 own the six off-diagonal patches (16 tiles), waves 6 / 7
// the triangles of (G0, G1) / (G2, G3) (20 tiles) -- 24 fragment reads per wave and chunk.  Work items of
// 32 chunks come from an atomic counter (as in syrk_co.hip).
// The sign work exists only in the NEG instantiation: d is screened on the device (one reduction
// kernel, no host synchronisation), both instantiations are launched and the one the flag does not
// select returns at once.
#include <algorithm>

#include "common.hpp"

namespace tmh {

constexpr int BX_W = 256;                     // padded columns
constexpr int BX_T = 136;                     // lower-triangular tiles
constexpr int BX_RS = 32;                     // rows per chunk = K of the MFMA
constexpr int BX_WAVES = 8;
constexpr int BX_THREADS = BX_WAVES * 64;
constexpr int BX_CSTR = 64;                   // bytes per column of a plane (32 rows x bf16, groups swizzled)
constexpr int BX_PLANE = BX_W * BX_CSTR;      // one piece: 16 384 B
constexpr int BX_PLANES = 3 * BX_PLANE;       // one chunk: 49 152 B
constexpr int BX_CPI = 32;                    // chunks per work item
constexpr int BX_ITEM_ROWS = BX_CPI * BX_RS;
constexpr int BX_RAWSTR = 1024 + 16;          // bytes per raw f32 row in LDS (256 columns + bank shift)
constexpr int BX_RAWBUF = BX_RS * BX_RAWSTR;  // one raw chunk: 33 280 B
constexpr int BX_NSLOT = 3;                   // raw ring
static_assert(BX_PLANES + BX_NSLOT * BX_RAWBUF + BX_NSLOT * BX_RS * 4 + 16 <= 160 * 1024, "LDS");

typedef __bf16 bx_frag __attribute__((ext_vector_type(8)));
typedef __bf16 bx_bf2 __attribute__((ext_vector_type(2)));
typedef float bx_f2 __attribute__((ext_vector_type(2)));
typedef float bx_f4 __attribute__((ext_vector_type(4)));
typedef unsigned bx_u4 __attribute__((ext_vector_type(4)));
typedef unsigned bx_u2 __attribute__((ext_vector_type(2)));

// tile t of the lower triangle of 16 blocks: (bi, bj <= bi) -> bi (bi + 1) / 2 + bj
constexpr int bx_tile(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }

// The patch of wave WID: row group GA, column group GB (off-diagonal waves), or the two diagonal
// groups D0, D1 of a diagonal wave.
template <int WID>
struct BxPatch {
    static constexpr bool DIAG = WID >= 6;
    static constexpr int GA = WID == 0 ? 1 : WID <= 2 ? 2 : 3;
    static constexpr int GB = WID == 0 ? 0 : WID == 1 ? 0 : WID == 2 ? 1 : WID - 3;
    static constexpr int D0 = WID == 6 ? 0 : 2, D1 = D0 + 1;
    static constexpr int NT = DIAG ? 20 : 16;
};

__device__ __forceinline__ unsigned bx_pack(float a, float b) {
    const bx_f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bx_bf2));     // v_cvt_pk_bf16_f32
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t bx_rsrc_t;
__device__ __forceinline__ bx_rsrc_t bx_rsrc(const void *base, int64_t bytes) {
    const unsigned nb = bytes <= 0 ? 0u : (bytes > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)nb, 0x00020000);
}
__device__ __forceinline__ void bx_dma16(bx_rsrc_t rs, void *lds, int voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)lds, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ void bx_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#else
struct bx_rsrc_t {};
inline bx_rsrc_t bx_rsrc(const void *, int64_t) { return {}; }
inline void bx_dma16(bx_rsrc_t, void *, int) {}
inline void bx_lds_barrier() {}
#endif

__device__ __forceinline__ void bx_sched_fence() { __builtin_amdgcn_sched_barrier(0); }

// flag != 0: a negative weight somewhere (the NEG instantiation runs)
__global__ void bx_screen_kernel(const float *__restrict__ d, int64_t n, unsigned *__restrict__ flag) {
    bool neg = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        neg |= d[i] < 0.0f;
    if (__builtin_amdgcn_ballot_w64(neg) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

template <bool NEG>
__global__ __launch_bounds__(BX_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
void syrk_bf16x3_kernel(const float *__restrict__ X, int64_t n, int64_t m, const float *__restrict__ d,
                        int n_items, unsigned *__restrict__ counter, float *__restrict__ part,
                        const unsigned *__restrict__ flag) {
    if ((*flag != 0u) != NEG) return;
    // separate static LDS objects (see syrk_i8.hip: LDS stores the wait-count pass cannot tell apart from the
    // target of an LDS-DMA copy wait for every copy in flight)
    __shared__ __attribute__((aligned(16))) unsigned char planes[BX_PLANES];        // [3][256][64 B]: ONE chunk
    __shared__ __attribute__((aligned(16))) unsigned char raw[BX_NSLOT * BX_RAWBUF]; // ring: [32 rows][BX_RAWSTR] f32
    __shared__ __attribute__((aligned(16))) float dl[BX_NSLOT * BX_RS];             // d of the ring slots
    __shared__ unsigned slot_mem[4];
    unsigned *slot = slot_mem;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // conversion role: column pair cp (of 8) and row quad r4 (of 8) -- the lanes of a half wave hold 4 row
    // quads x 8 column pairs: their ds_read_b64 of the raw rows (row stride 260 dwords) hit 64 distinct banks
    const int cp = lane & 7, r4 = ((lane >> 3) & 3) + 4 * (lane >> 5);

    // ---- the workgroup's chunk stream (see syrk_co.hip)
    unsigned idL = blockIdx.x;
    int oL = 0;
    unsigned idNext;
    if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
    __syncthreads();
    idNext = __builtin_amdgcn_readfirstlane(*slot);
    bool pending = false;

    // Wave w copies rows 4 w .. 4 w + 3 of a chunk: one instruction = one row = 64 lanes x 16 bytes.
    float dreg = 0.0f;
    const int dma_voff = lane * 4 < m ? lane * 16 : 0x7ffffff0;           // columns >= m read as 0
    bx_rsrc_t cur_rs;                                                     // descriptor of the chunk being requested
    auto issue_piece = [&](int rb, int j) {
#if !defined(BX_ABLATE_NO_DMA)         // (timing only)
        bx_dma16(cur_rs, raw + rb * BX_RAWBUF + (4 * wave + j) * BX_RAWSTR, (int)((4 * wave + j) * m * 4) + dma_voff);
#endif
    };
    auto issue_begin = [&](int rb) -> unsigned {                          // rb: ring slot
        const unsigned id = idL;
        const int64_t tb = (int64_t)id * BX_ITEM_ROWS + (int64_t)oL * BX_RS;
        // the chunk's rows that exist: everything beyond reads as 0 (rows >= n, columns >= m)
        const int64_t left = min((n - tb) * m * 4, (int64_t)BX_RS * m * 4);
        // (uniform by construction, but the 64-bit products are VALU work: without the readfirstlane the
        // descriptor counts as divergent and every copy sits in a waterfall loop)
        const uint64_t xb = (uint64_t)(uintptr_t)(X + tb * m);
        const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xb);
        const unsigned xhi = __builtin_amdgcn_readfirstlane((unsigned)(xb >> 32));
        const unsigned nbytes = __builtin_amdgcn_readfirstlane(
            left <= 0 ? 0u : (left > 0xFFFFFFFFll ? 0xFFFFFFFFu : (unsigned)left));
        cur_rs = bx_rsrc(reinterpret_cast<const void *>((uintptr_t)(((uint64_t)xhi << 32) | xlo)), nbytes);
        // (d first: the wave that loads it then waits for it with the 4 younger copies still in flight)
        if (wave == 0 && lane < BX_RS) dreg = tb + lane < n ? d[tb + lane] : 0.0f;
        if (++oL == BX_CPI) {
            oL = 0;
            idL = idNext;
            if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
            pending = true;
        }
        return id;
    };
    auto issue_chunk = [&](int rb) -> unsigned {                          // the whole chunk at once (prologue)
        const unsigned id = issue_begin(rb);
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_piece(rb, j);
        return id;
    };
    auto publish_d = [&](int rb) {          // d of the chunk just requested (wave 0, once its load has landed)
        if (wave == 0) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (lane < BX_RS) dl[rb * BX_RS + lane] = dreg;
        }
    };

    // ---- conversion: unit u (0 / 1) = columns 32 wave + 16 u + 2 cp, + 1; rows 4 r4 .. 4 r4 + 3
    float sd[4];
    struct Unit { bx_f2 x[4]; };
    auto convert_begin = [&](int rb) {
        const bx_f4 dv = *reinterpret_cast<const bx_f4 *>(dl + rb * BX_RS + 4 * r4);
#pragma unroll
        for (int j = 0; j < 4; ++j) sd[j] = __builtin_amdgcn_sqrtf(__builtin_fabsf(dv[j]));
    };
    // (inline asm: the compiler pairs these loads into ds_read2_b64, whose merged memory operand loses the LDS
    // variable -- and an LDS access the wait-count pass cannot attribute waits for EVERY LDS-DMA copy in
    // flight, s_waitcnt vmcnt(0) in every chunk; complete after unit_wait)
    const unsigned raw_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)raw;
    const unsigned planes_lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)planes;
    auto unit_read = [&](int rb, int u) {
        Unit q;
        const unsigned ra = raw_lds + (unsigned)(rb * BX_RAWBUF + (4 * r4) * BX_RAWSTR + (32 * wave + 16 * u + 2 * cp) * 4);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1040\n\tds_read_b64 %2, %4 offset:2080\n\t"
                     "ds_read_b64 %3, %4 offset:3120"
                     : "=&v"(q.x[0]), "=&v"(q.x[1]), "=&v"(q.x[2]), "=&v"(q.x[3])
                     : "v"(ra)
                     : "memory");
#else
        (void)ra;
        q.x[0] = q.x[1] = q.x[2] = q.x[3] = bx_f2{0.0f, 0.0f};
#endif
        static_assert(BX_RAWSTR == 1040, "the ds_read offsets above are multiples of the raw row stride");
        return q;
    };
    auto unit_wait = [&](Unit &q) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q.x[0]), "+v"(q.x[1]), "+v"(q.x[2]), "+v"(q.x[3])::"memory");
#endif
    };
    // one column (e = 0 / 1) of a unit: scale by sqrt|d|, split into three bf16 pieces, write the planes
    auto unit_col = [&](const Unit &q, int u, auto ec) {
        constexpr int e = decltype(ec)::value;
#if defined(BX_ABLATE_NO_CONVERT)      // (timing only)
        return;
#endif
        const int col = 32 * wave + 16 * u + 2 * cp + e;
        // 16-byte row group r4 >> 1 of the column, stored at (group ^ (col >> 1)) & 3 (see frag below)
        const unsigned pa = planes_lds + (unsigned)(col * BX_CSTR + ((((r4 >> 1) ^ cp) & 3) << 4) + (r4 & 1) * 8);
        // (round-to-nearest pieces; splitting by truncation -- h = the top 16 bits, one v_perm_b32 per pair -- is
        // exact as a sum too and 25 % cheaper, but its pieces are twice as large and biased: the dropped
        // products (m, l) (l, m) then cost 2e-5 instead of 2e-6 against the f32 path)
        float uu[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) uu[j] = q.x[j][e] * sd[j];
        bx_u2 h, mm, l;
        float r[4];
        h[0] = bx_pack(uu[0], uu[1]);
        h[1] = bx_pack(uu[2], uu[3]);
        r[0] = uu[0] - __uint_as_float(h[0] << 16);
        r[1] = uu[1] - __uint_as_float(h[0] & 0xffff0000u);
        r[2] = uu[2] - __uint_as_float(h[1] << 16);
        r[3] = uu[3] - __uint_as_float(h[1] & 0xffff0000u);
        mm[0] = bx_pack(r[0], r[1]);
        mm[1] = bx_pack(r[2], r[3]);
        r[0] -= __uint_as_float(mm[0] << 16);
        r[1] -= __uint_as_float(mm[0] & 0xffff0000u);
        r[2] -= __uint_as_float(mm[1] << 16);
        r[3] -= __uint_as_float(mm[1] & 0xffff0000u);
        l[0] = bx_pack(r[0], r[1]);
        l[1] = bx_pack(r[2], r[3]);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:16384\n\tds_write_b64 %0, %3 offset:32768"
                     :
                     : "v"(pa), "v"(h), "v"(mm), "v"(l)
                     : "memory");
#else
        (void)pa;
#endif
        static_assert(BX_PLANE == 16384, "the ds_write offsets above are the plane strides");
    };
    // fragment of block b, piece p: lane (i = lane & 15, kg = lane >> 4) -> rows 8 kg .. 8 kg + 7 of column
    // 16 b + i, found in the 16-byte group (kg ^ (column >> 1)) & 3 of the column's 64 bytes
    const int foff = (lane & 15) * BX_CSTR + ((((lane >> 4) ^ (lane >> 1)) & 3) << 4);
    auto frag = [&](int b, int p) {
        return *reinterpret_cast<const bx_u4 *>(planes + p * BX_PLANE + b * 16 * BX_CSTR + foff);
    };
    auto mma = [](bx_u4 a, bx_u4 b, bx_f4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bx_frag, a), __builtin_bit_cast(bx_frag, b),
                                                      c, 0, 0, 0);
    };

    auto run = [&](auto wid) {
        constexpr int WID = decltype(wid)::value;
        using P = BxPatch<WID>;
        bx_f4 acc[P::NT];
#pragma unroll
        for (int t = 0; t < P::NT; ++t) acc[t] = bx_f4{0, 0, 0, 0};
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using TT = std::true_type;
        using FF = std::false_type;

        // prologue: chunks 0, 1, 2 -> ring slots 0, 1, 2; chunk 0 -> planes
        unsigned id_c = issue_chunk(0);
        publish_d(0);
        unsigned id_c1 = issue_chunk(1);
        publish_d(1);
        unsigned id_c2 = issue_chunk(2);
        publish_d(2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");           // chunk 0 has landed (this wave's rows)
        bx_lds_barrier();
        if (pending) { idNext = __builtin_amdgcn_readfirstlane(*slot); pending = false; }
        convert_begin(0);
        {
            Unit q0 = unit_read(0, 0), q1 = unit_read(0, 1);
            unit_wait(q0);
            unit_wait(q1);
            unit_col(q0, 0, I0{});
            unit_col(q0, 0, I1{});
            unit_col(q1, 1, I0{});
            unit_col(q1, 1, I1{});
        }
        int s0 = 0;                                  // ring slot of chunk c (planes), then c + 1, c + 2
        while (id_c < (unsigned)n_items) {
            const int s1 = s0 == 2 ? 0 : s0 + 1;
            // ---- A: planes hold chunk c, slot s0 is converted by everybody, dl is settled
            bx_lds_barrier();
            if (pending) { idNext = __builtin_amdgcn_readfirstlane(*slot); pending = false; }
            // chunk c + 3 -> the slot of chunk c: its four copies are issued one behind each product group
            // after barrier B (an LDS-DMA piece costs ~60 cycles of issue among bare MFMAs and 100-185 inside
            // a phase full of ds_read_b128 -- MI355X_MICROARCH.md -- all four in the fragment phase right
            // behind barrier A made the copies cost more than the conversion)
            const unsigned id_c3 = issue_begin(s0);
#if defined(BX_DMA_BURST)              // (timing only: the first form, all four copies behind barrier A)
            for (int j = 0; j < 4; ++j) issue_piece(s0, j);
#endif
            // sign masks of this lane's 8 rows (pairs of rows per register, as the bf16 are packed)
            bx_u4 sm = bx_u4{0u, 0u, 0u, 0u};
            if constexpr (NEG) {
                const bx_f4 da = *reinterpret_cast<const bx_f4 *>(dl + s0 * BX_RS + 8 * (lane >> 4));
                const bx_f4 db = *reinterpret_cast<const bx_f4 *>(dl + s0 * BX_RS + 8 * (lane >> 4) + 4);
                sm[0] = ((__float_as_uint(da[0]) >> 16) & 0x8000u) | (__float_as_uint(da[1]) & 0x80000000u);
                sm[1] = ((__float_as_uint(da[2]) >> 16) & 0x8000u) | (__float_as_uint(da[3]) & 0x80000000u);
                sm[2] = ((__float_as_uint(db[0]) >> 16) & 0x8000u) | (__float_as_uint(db[1]) & 0x80000000u);
                sm[3] = ((__float_as_uint(db[2]) >> 16) & 0x8000u) | (__float_as_uint(db[3]) & 0x80000000u);
            }
            // the fragments of the wave's patch(es): pieces 0 and 2 first (the first two products need them)
            bx_u4 FA[4][3], FB[4][3];               // off-diagonal: A side (row group GA) / B side (GB);
                                                     // diagonal: group D0 / group D1 (both sides each)
            constexpr int ga = P::DIAG ? P::D0 : P::GA, gb = P::DIAG ? P::D1 : P::GB;
            static_for<3>([&](auto oc) {
                constexpr int order[3] = {0, 2, 1};
                constexpr int p = order[decltype(oc)::value];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    FA[q][p] = frag(4 * ga + q, p);
                    FB[q][p] = frag(4 * gb + q, p);
                    if constexpr (NEG && !P::DIAG) FA[q][p] ^= sm;
                }
            });
            // one piece product over a patch; consecutive MFMAs go to different tiles
            auto prod = [&](auto tri_c, auto t0_c, bx_u4 (&A)[4][3], bx_u4 (&B)[4][3], auto pa_c, auto pb_c) {
                constexpr bool TRI = decltype(tri_c)::value;
                constexpr int pa = decltype(pa_c)::value, pbb = decltype(pb_c)::value;
                int t = decltype(t0_c)::value;
#pragma unroll
                for (int qi = 0; qi < 4; ++qi)
#pragma unroll
                    for (int qj = 0; qj < 4; ++qj) {
                        if (TRI && qj > qi) continue;
                        bx_u4 a = TRI ? B[qi][pa] : A[qi][pa];
                        if constexpr (NEG && TRI) a ^= sm;
                        acc[t] = mma(a, B[qj][pbb], acc[t]);
                        ++t;
                    }
            };
            using T0 = std::integral_constant<int, 0>;
            using T10 = std::integral_constant<int, 10>;
            // the six piece products, small terms first: (l, h) (h, l) (m, m) (m, h) (h, m) (h, h)
            auto group = [&](auto pa_c, auto pb_c) {
                if constexpr (P::DIAG) {
                    prod(TT{}, T0{}, FA, FA, pa_c, pb_c);
                    prod(TT{}, T10{}, FB, FB, pa_c, pb_c);
                } else {
                    prod(FF{}, T0{}, FA, FB, pa_c, pb_c);
                }
            };
            group(I2{}, I0{});
            group(I0{}, I2{});
            bx_sched_fence();
            // ---- B: every wave holds its fragments (the planes may be rewritten) and its rows of chunk
            // c + 1 have landed (the 4 copies of chunk c + 2 -- and wave 0's load of d for chunk c + 3 -- stay
            // in flight; the copies of chunk c + 3 follow behind this barrier)
#if defined(BX_DMA_BURST)
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#else
            if (wave == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#endif
            bx_lds_barrier();
            convert_begin(s1);
            Unit q0 = unit_read(s1, 0);
            Unit q1 = unit_read(s1, 1);
            bx_sched_fence();
            group(I1{}, I1{});
            unit_wait(q0);
            unit_wait(q1);
            unit_col(q0, 0, I0{});
#if !defined(BX_DMA_BURST)
            issue_piece(s0, 0);
#endif
            bx_sched_fence();
            group(I1{}, I0{});
            unit_col(q0, 0, I1{});
#if !defined(BX_DMA_BURST)
            issue_piece(s0, 1);
#endif
            bx_sched_fence();
            group(I0{}, I1{});
            unit_col(q1, 1, I0{});
#if !defined(BX_DMA_BURST)
            issue_piece(s0, 2);
#endif
            bx_sched_fence();
            group(I0{}, I0{});
            unit_col(q1, 1, I1{});
#if !defined(BX_DMA_BURST)
            issue_piece(s0, 3);
#endif
            bx_sched_fence();
            // d of chunk c + 3 replaces d of chunk c (its sign masks are in registers, its rows converted)
            publish_d(s0);
            id_c = id_c1;
            id_c1 = id_c2;
            id_c2 = id_c3;
            s0 = s1;
        }
        // ---- partial tiles [t][16][16]: C layout col = lane & 15 (B side), row = 4 (lane >> 4) + reg (A side)
        float *dst = part + (int64_t)blockIdx.x * (BX_T * 256);
        auto store_patch = [&](int ga, int gb, bool tri, int t0) {
            int t = t0;
            for (int qi = 0; qi < 4; ++qi)
                for (int qj = 0; qj < 4; ++qj) {
                    if (tri && qj > qi) continue;
                    const int tile = bx_tile(4 * ga + qi, 4 * gb + qj);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[tile * 256 + (4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[t][r];
                    ++t;
                }
        };
        if constexpr (P::DIAG) {
            store_patch(P::D0, P::D0, true, 0);
            store_patch(P::D1, P::D1, true, 10);
        } else {
            store_patch(P::GA, P::GB, false, 0);
        }
    };
    switch (wave) {
        case 0: run(std::integral_constant<int, 0>{}); break;
        case 1: run(std::integral_constant<int, 1>{}); break;
        case 2: run(std::integral_constant<int, 2>{}); break;
        case 3: run(std::integral_constant<int, 3>{}); break;
        case 4: run(std::integral_constant<int, 4>{}); break;
        case 5: run(std::integral_constant<int, 5>{}); break;
        case 6: run(std::integral_constant<int, 6>{}); break;
        default: run(std::integral_constant<int, 7>{}); break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (copies of chunks beyond the end are still in flight)
}

// Sum of the partial tiles in a fixed order (double), mirrored into out; a quarter tile per block
__global__ __launch_bounds__(1024) void syrk_bf16x3_finish_kernel(const float *__restrict__ part, int nblk,
                                                                  int n_cols, float *__restrict__ out,
                                                                  int64_t ldo) {
    __shared__ double red[16][64];
    const int e = blockIdx.y * 64 + threadIdx.x, s = threadIdx.y, t = blockIdx.x;
    double a = 0.0;
    for (int b = s; b < nblk; b += 16) a += (double)part[((int64_t)b * BX_T + t) * 256 + e];
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
        // (a diagonal tile holds (i, j) and (j, i) as separately rounded sums: the lower one is mirrored)
        if (ci < n_cols && cj < n_cols && (bi != bj || (e >> 4) >= (e & 15))) {
            out[(int64_t)ci * ldo + cj] = (float)v;
            if (ci != cj) out[(int64_t)cj * ldo + ci] = (float)v;
        }
    }
}

int run_syrk_bf16x3(const float *X, int64_t n, int64_t m, const float *d, float *out, hipStream_t st) {
    TM_REQUIRE(n >= 0 && m >= 0, "negative shape");
    TM_REQUIRE(m == 0 || syrk_bf16x3_ok(X, m), "the bf16x3 syrk takes a 16-byte aligned C-ordered f32 block "
                                                "of 4 k <= 256 columns");
    if (m == 0) return TM_OK;
    if (n == 0) {
        TM_HIP(hipMemsetAsync(out, 0, sizeof(float) * (size_t)(m * m), st));
        return TM_OK;
    }
    const int64_t n_items64 = ceil_div(n, BX_ITEM_ROWS);
    TM_REQUIRE(n_items64 < (1ll << 31), "too many rows");
    const int n_items = (int)n_items64;
    const int grid = (int)std::min<int64_t>(n_items, tune("bx_grid", 2 * NUM_CU));
    void *wsv = nullptr;
    int rc = get_workspace(256 + sizeof(float) * (size_t)grid * BX_T * 256, &wsv, st);
    if (rc) return rc;
    unsigned *counter = reinterpret_cast<unsigned *>(wsv);
    unsigned *flag = counter + 32;                 // != 0: a negative weight (the NEG instantiation runs)
    float *part = reinterpret_cast<float *>(reinterpret_cast<char *>(wsv) + 256);
    TM_HIP(hipMemsetAsync(counter, 0, 256, st));
    const int sgrid = (int)std::min<int64_t>(2 * NUM_CU, ceil_div(n, 1024));
    hipLaunchKernelGGL(bx_screen_kernel, dim3((unsigned)sgrid), dim3(256), 0, st, d, n, flag);
    TM_LAUNCH_CHECK();
    prof_begin(st);
    hipLaunchKernelGGL(syrk_bf16x3_kernel<false>, dim3((unsigned)grid), dim3(BX_THREADS), 0, st, X, n, m, d,
                       n_items, counter, part, flag);
    prof_end(st);
    TM_LAUNCH_CHECK();
    prof_hold(true);               // (the event pair stays on the first launch: one of the two returns at once)
    hipLaunchKernelGGL(syrk_bf16x3_kernel<true>, dim3((unsigned)grid), dim3(BX_THREADS), 0, st, X, n, m, d,
                       n_items, counter, part, flag);
    prof_hold(false);
    TM_LAUNCH_CHECK();
    hipLaunchKernelGGL(syrk_bf16x3_finish_kernel, dim3(BX_T, 4), dim3(64, 16), 0, st, part, grid, (int)m, out, m);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

}  // namespace tmh

extern "C" {

int tm_dense_sandwich_bf16x3_f32(const float *X, int64_t n, int64_t m, const float *d, float *out,
                                 void *stream) {
    return tmh::run_syrk_bf16x3(X, n, m, d, out, tmh::as_stream(stream));
}

}  // extern "C"
