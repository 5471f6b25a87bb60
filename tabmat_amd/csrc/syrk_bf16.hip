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
// Layout: a workgroup of 8 waves streams 32-row chunks (K = 32 of the MFMA).  The raw f32 rows come in
// by LDS-DMA (buffer_load ... lds, no registers) TWO chunks ahead; per chunk every lane scales,
// splits and writes 4 columns x 4 rows of the three bf16 planes [piece][column][32 rows] (column
// stride 80 B: conflict-free 16-byte fragment reads) -- one column at a time behind the MFMA groups.
// The planes are single-buffered: all fragments of a chunk are read into registers between two
// barriers, then the planes are rewritten while the MFMAs run from registers.  The 136 lower-
// triangular 16 x 16 tiles are dealt as 4 x 4 patches of column-block groups G0 .. G3: waves 0 .. 5
// own the six off-diagonal patches (16 tiles), waves 6 / 7 the triangles of (G0, G1) / (G2, G3)
// (20 tiles) -- 24 fragment reads per wave and chunk.  Work items of 32 chunks come from an atomic
// counter (as in syrk_co.hip).
// Measured at 10M x 256 (profiles/r3_cfg2_bf16x3.txt): 4.04 ms against 5.75 ms for the f32-input MFMA;
// the conversion (~105 vector instructions per lane and chunk) and the MFMAs share the SIMD's issue
// slots, and the fragment phase between the two barriers is not overlapped.
#include <algorithm>

#include "common.hpp"

namespace tmh {

constexpr int BX_W = 256;                     // padded columns
constexpr int BX_T = 136;                     // lower-triangular tiles
constexpr int BX_RS = 32;                     // rows per chunk = K of the MFMA
constexpr int BX_WAVES = 8;
constexpr int BX_THREADS = BX_WAVES * 64;
constexpr int BX_CSTR = 80;                   // bytes per column of a plane (64 + 16: bank spread)
constexpr int BX_PLANE = BX_W * BX_CSTR;      // one piece
constexpr int BX_BUF = 3 * BX_PLANE;          // one chunk
constexpr int BX_CPI = 32;                    // chunks per work item
constexpr int BX_ITEM_ROWS = BX_CPI * BX_RS;
constexpr int BX_RAWSTR = 1024 + 16;          // bytes per raw f32 row in LDS (256 columns + bank shift)
constexpr int BX_RAWBUF = BX_RS * BX_RAWSTR;  // one raw chunk
constexpr size_t BX_LDS = (size_t)BX_BUF + 2 * (size_t)BX_RAWBUF + 2 * BX_RS * sizeof(float) + 16;

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

__global__ __launch_bounds__(BX_THREADS) __attribute__((amdgpu_waves_per_eu(2)))
void syrk_bf16x3_kernel(const float *__restrict__ X, int64_t n, int64_t m, const float *__restrict__ d,
                        int n_items, unsigned *__restrict__ counter, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *planes = smem;                                         // [3][256][BX_CSTR]: ONE chunk
    unsigned char *raw = smem + BX_BUF;                                   // [2][32 rows][BX_RAWSTR] f32
    float *dl = reinterpret_cast<float *>(raw + 2 * BX_RAWBUF);           // [2][BX_RS]
    unsigned *slot = reinterpret_cast<unsigned *>(dl + 2 * BX_RS);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // conversion role: 4 columns x 4 rows of the chunk
    const int sq = lane & 7, sg = lane >> 3;
    const int scol = 32 * wave + 4 * sq;

    // ---- the workgroup's chunk stream (see syrk_co.hip)
    unsigned idL = blockIdx.x;
    int oL = 0;
    unsigned idNext;
    if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
    __syncthreads();
    idNext = __builtin_amdgcn_readfirstlane(*slot);
    bool pending = false;

    // The raw f32 rows of a chunk come in by LDS-DMA (buffer_load ... lds: no registers, nobody waits for
    // its own loads), TWO chunks ahead of the MFMAs: 64 KB in flight per CU.  With the chunk staged in
    // registers one iteration ahead (32 KB per CU, 8 MB over the chip) the stream was capped at ~2 TB/s:
    // loads + conversion took 4.9 ms where the loads alone (never waited for) took 2.1 ms.
    // Wave w copies rows 4 w .. 4 w + 3: one instruction = one row = 64 lanes x 16 bytes.
    float dreg = 0.0f;
    const int dma_voff = lane * 4 < m ? lane * 16 : 0x7ffffff0;           // columns >= m read as 0
    auto issue_chunk = [&](int rb) -> unsigned {                          // rb: raw buffer 0 / 1
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
        const bx_rsrc_t rs = bx_rsrc(reinterpret_cast<const void *>((uintptr_t)(((uint64_t)xhi << 32) | xlo)), nbytes);
        // (d first: the wave that loads it then waits for it with the 4 younger copies still in flight)
        if (wave == 0 && lane < BX_RS) dreg = tb + lane < n ? d[tb + lane] : 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            bx_dma16(rs, raw + rb * BX_RAWBUF + (4 * wave + j) * BX_RAWSTR,
                     (int)((4 * wave + j) * m * 4) + dma_voff);
        if (++oL == BX_CPI) {
            oL = 0;
            idL = idNext;
            if (tid == 0) *slot = gridDim.x + atomicAdd(counter, 1u);
            pending = true;
        }
        return id;
    };
    // one of the lane's four columns of the raw chunk rb: scale by sqrt|d|, split, write the planes
    float sd[4];
    auto convert_begin = [&](int rb) {
        const bx_f4 dv = *reinterpret_cast<const bx_f4 *>(dl + rb * BX_RS + 4 * sg);
#pragma unroll
        for (int j = 0; j < 4; ++j) sd[j] = __builtin_amdgcn_sqrtf(__builtin_fabsf(dv[j]));
    };
    auto convert_col = [&](int rb, auto kc) {
        constexpr int k = decltype(kc)::value;
        const unsigned char *rp = raw + rb * BX_RAWBUF + (4 * sg) * BX_RAWSTR + (scol + k) * 4;
        unsigned char *pb = planes + (scol + k) * BX_CSTR + sg * 8;
        float u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = *reinterpret_cast<const float *>(rp + j * BX_RAWSTR) * sd[j];
        bx_u2 h, mm, l;
        float r[4];
        h[0] = bx_pack(u[0], u[1]);
        h[1] = bx_pack(u[2], u[3]);
        r[0] = u[0] - __uint_as_float(h[0] << 16);
        r[1] = u[1] - __uint_as_float(h[0] & 0xffff0000u);
        r[2] = u[2] - __uint_as_float(h[1] << 16);
        r[3] = u[3] - __uint_as_float(h[1] & 0xffff0000u);
        mm[0] = bx_pack(r[0], r[1]);
        mm[1] = bx_pack(r[2], r[3]);
        r[0] -= __uint_as_float(mm[0] << 16);
        r[1] -= __uint_as_float(mm[0] & 0xffff0000u);
        r[2] -= __uint_as_float(mm[1] << 16);
        r[3] -= __uint_as_float(mm[1] & 0xffff0000u);
        l[0] = bx_pack(r[0], r[1]);
        l[1] = bx_pack(r[2], r[3]);
        *reinterpret_cast<bx_u2 *>(pb) = h;
        *reinterpret_cast<bx_u2 *>(pb + BX_PLANE) = mm;
        *reinterpret_cast<bx_u2 *>(pb + 2 * BX_PLANE) = l;
    };
    // fragment of block b, piece p: lane (i = lane & 15, kg = lane >> 4) -> rows 8 kg .. 8 kg + 7 of column 16 b + i
    const int foff = (lane & 15) * BX_CSTR + (lane >> 4) * 16;
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
        using I3 = std::integral_constant<int, 3>;

        // prologue: chunk 0 -> raw 0 -> planes; chunk 1 -> raw 1
        unsigned id_c = issue_chunk(0);
        if (wave == 0 && lane < BX_RS) dl[lane] = dreg;                   // d of chunk 0
        unsigned id_c1 = issue_chunk(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bx_lds_barrier();
        if (pending) { idNext = __builtin_amdgcn_readfirstlane(*slot); pending = false; }
        convert_begin(0);
        static_for<4>([&](auto kc) { convert_col(0, kc); });
        if (wave == 0 && lane < BX_RS) dl[BX_RS + lane] = dreg;           // d of chunk 1
        int par = 0;                                                     // chunk c lives in dl[par]; raw[par] is free
        while (id_c < (unsigned)n_items) {
            // ---- top: chunk c + 2 -> raw[par] (its rows were converted one iteration ago)
            bx_lds_barrier();                       // A: planes hold chunk c, raw[par] and dl are settled
            if (pending) { idNext = __builtin_amdgcn_readfirstlane(*slot); pending = false; }
            const unsigned id_c2 = issue_chunk(par);
            // sign masks of this lane's 8 rows (pairs of rows per register, as the bf16 are packed)
            const bx_f4 da = *reinterpret_cast<const bx_f4 *>(dl + par * BX_RS + 8 * (lane >> 4));
            const bx_f4 db = *reinterpret_cast<const bx_f4 *>(dl + par * BX_RS + 8 * (lane >> 4) + 4);
            bx_u4 sm;
            sm[0] = ((__float_as_uint(da[0]) >> 16) & 0x8000u) | (__float_as_uint(da[1]) & 0x80000000u);
            sm[1] = ((__float_as_uint(da[2]) >> 16) & 0x8000u) | (__float_as_uint(da[3]) & 0x80000000u);
            sm[2] = ((__float_as_uint(db[0]) >> 16) & 0x8000u) | (__float_as_uint(db[1]) & 0x80000000u);
            sm[3] = ((__float_as_uint(db[2]) >> 16) & 0x8000u) | (__float_as_uint(db[3]) & 0x80000000u);

            // one 4 x 4 patch: row blocks 4 ga .. 4 ga + 3 (A side, sign-flipped), column blocks 4 gb ..
            // (B side); TRI: only bj <= bi (a diagonal group, same blocks on both sides).  All fragments
            // are read first; after barrier B the single planes buffer is free and the conversion of
            // the next chunk is issued column by column behind the product groups of the first patch.
            bx_u4 B0[4][3], A0[4][3], B1[4][3];
            auto read_patch = [&](auto ga_c, auto gb_c, auto tri_c, bx_u4 (&A)[4][3], bx_u4 (&B)[4][3]) {
                constexpr int ga = decltype(ga_c)::value, gb = decltype(gb_c)::value;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        B[q][p] = frag(4 * gb + q, p);
                        if constexpr (!decltype(tri_c)::value) A[q][p] = frag(4 * ga + q, p) ^ sm;
                    }
            };
            using TT = std::true_type;
            using FF = std::false_type;
            if constexpr (P::DIAG) {
                read_patch(std::integral_constant<int, P::D0>{}, std::integral_constant<int, P::D0>{}, TT{}, A0, B0);
                read_patch(std::integral_constant<int, P::D1>{}, std::integral_constant<int, P::D1>{}, TT{}, A0, B1);
            } else {
                read_patch(std::integral_constant<int, P::GA>{}, std::integral_constant<int, P::GB>{}, FF{}, A0, B0);
            }
            // B: every wave holds its fragments (the planes may be overwritten) and its share of chunk
            // c + 1 has landed in raw[par ^ 1] (the 4 copies of chunk c + 2 just issued stay in flight)
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            bx_lds_barrier();
            const int rb = par ^ 1;
            // (a branch on "no negative weight in this chunk" around the sign work was tried: MFMA
            // accumulators that meet at a control-flow join are kept twice -- 187 spills, 6.0 ms)
            auto prods = [&](auto tri_c, auto t0_c, bx_u4 (&A)[4][3], bx_u4 (&B)[4][3], auto stage_c) {
                constexpr bool TRI = decltype(tri_c)::value;
                constexpr int T0 = decltype(t0_c)::value;
                constexpr bool ST = decltype(stage_c)::value;
                // the six piece products, small terms first; consecutive MFMAs go to different tiles
                auto prod = [&](auto pa_c, auto pb_c) {
                    constexpr int pa = decltype(pa_c)::value, pbb = decltype(pb_c)::value;
                    int t = T0;
#pragma unroll
                    for (int qi = 0; qi < 4; ++qi)
#pragma unroll
                        for (int qj = 0; qj < 4; ++qj) {
                            if (TRI && qj > qi) continue;
                            acc[t] = mma(TRI ? (B[qi][pa] ^ sm) : A[qi][pa], B[qj][pbb], acc[t]);
                            ++t;
                        }
                };
                if constexpr (ST) convert_begin(rb);
                prod(I2{}, I0{});
                if constexpr (ST) convert_col(rb, I0{});
                prod(I0{}, I2{});
                if constexpr (ST) convert_col(rb, I1{});
                prod(I1{}, I1{});
                if constexpr (ST) convert_col(rb, I2{});
                prod(I1{}, I0{});
                if constexpr (ST) convert_col(rb, I3{});
                prod(I0{}, I1{});
                prod(I0{}, I0{});
            };
            if constexpr (P::DIAG) {
                prods(TT{}, std::integral_constant<int, 0>{}, A0, B0, TT{});
                prods(TT{}, std::integral_constant<int, 10>{}, A0, B1, FF{});
            } else {
                prods(FF{}, std::integral_constant<int, 0>{}, A0, B0, TT{});
            }
            // d of chunk c + 2 (loaded at the top) replaces d of chunk c, whose sign masks are in registers
            if (wave == 0 && lane < BX_RS) dl[par * BX_RS + lane] = dreg;
            id_c = id_c1;
            id_c1 = id_c2;
            par ^= 1;
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
    float *part = reinterpret_cast<float *>(reinterpret_cast<char *>(wsv) + 256);
    TM_HIP(hipMemsetAsync(counter, 0, 256, st));
    TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(syrk_bf16x3_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)BX_LDS));
    prof_begin(st);
    hipLaunchKernelGGL(syrk_bf16x3_kernel, dim3((unsigned)grid), dim3(BX_THREADS), BX_LDS, st, X, n, m, d,
                       n_items, counter, part);
    prof_end(st);
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
