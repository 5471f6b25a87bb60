// Shared host/device helpers for libtabmat_hip.so (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tabmat_hip.h"

namespace tmh {

constexpr int WAVE = 64;           // CDNA wavefront
constexpr int NUM_CU = 256;        // MI355X
constexpr int NUM_XCD = 8;
constexpr size_t LDS_BYTES = 160 * 1024;

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define TM_HIP(expr)                                                  \
    do {                                                              \
        hipError_t _e = (expr);                                       \
        if (_e != hipSuccess) return tmh::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define TM_LAUNCH_CHECK() TM_HIP(hipGetLastError())

#define TM_REQUIRE(cond, msg)                 \
    do {                                      \
        if (!(cond)) {                        \
            tmh::set_error("%s: %s", __func__, msg); \
            return TM_EINVAL;                 \
        }                                     \
    } while (0)

// Scratch per (device, stream): get_workspace() returns at least `bytes` of device memory that
// stays valid until the next get_workspace() call for the same device and stream asks for more.
// Ops in flight on different streams therefore never share partial-sum buffers.  Growing waits
// for that stream only and is refused while the stream is being captured into a graph (the
// pointer baked into the graph would dangle): warm the op up once before capturing.  A
// caller-provided workspace (tm_set_workspace) serves every stream of its device -- single-stream
// use is then the caller's contract.
int get_workspace(size_t bytes, void **ptr, hipStream_t st);

// Optional in-library kernel timing (tm_profile_enable): HIP events recorded on the launch
// stream immediately around the MAIN kernel of an op, so bench.py can quote the dominant
// kernel's duration without a profiler attached.
void prof_begin(hipStream_t st);
void prof_end(hipStream_t st);
void prof_hold(bool on);      // keep the recorded pair on the current kernel while a nested op launches

// integer tuning knob set with tm_tune_set (default when unset)
int64_t tune(const char *key, int64_t dflt);

// K1c (syrk_co.hip): X' diag(d) X of an unrestricted C-ordered f64 block of <= 128 columns (any parity since
// round 5: the 16-byte loads of an odd-width block start at 8-byte aligned addresses, which the hardware takes);
// colsum (may be NULL) receives X' d.  syrk_co_ok() says whether a block qualifies.
// center (may be NULL): per-column centres c -- the product (and column sums) of X - 1 c'.
int run_syrk_co(const double *X, int64_t n, int64_t m, const double *d, double *out,
                double *colsum, hipStream_t st, const double *center = nullptr);
inline bool syrk_co_ok(const void *X, int64_t m) {
    return m > 0 && m <= 128 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
}
// ... and pays: the kernel always works on the 36 tiles of a 128-column panel, so narrower blocks
// stay with the syrk_kernel instantiations of 16 / 32 / 64 columns
inline bool syrk_co_pays(int64_t m) { return m > 64; }

// K1e (syrk_i8.hip): the float64 syrk on the int8 matrix cores, for a block of <= 128 even columns or -- through
// the row strides ldx / ldo of X / out -- for a 128-column panel of a wider block in place.
int run_syrk_i8_panel(const double *X, int64_t ldx, int64_t n, int64_t m, const double *d, const double *colmax,
                      double *out, int64_t ldo, double *colsum, int *history, hipStream_t st,
                      const double *center = nullptr);

// K1d (syrk_bf16.hip): X' diag(d) X of an unrestricted C-ordered f32 block of 4 k <= 256 columns on the
// bf16 matrix cores (three-piece split, f32 accumulation); it pays above 128 columns.
int run_syrk_bf16x3(const float *X, int64_t n, int64_t m, const float *d, float *out, hipStream_t st);
inline bool syrk_bf16x3_ok(const void *X, int64_t m) {
    return m > 0 && m <= 256 && m % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0;
}

// K1n (syrk_narrow.hip): unrestricted blocks of at most 11 columns, any order / alignment
bool syrk_narrow_ok(int64_t m);
template <typename F>
int run_syrk_narrow(const F *X, int64_t n, int64_t m, int order_f, const F *d, F *out, hipStream_t st);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------
// Accumulator type of LDS tiles that are updated with atomics: ALWAYS double.  On gfx950
// ds_add_f32 runs lane-serially -- 169 cycles per 64-lane instruction against 7.6 for ds_add_f64
// (scripts/ubench/atomics_f32.hip) -- so float data is accumulated in double tiles (twice the
// LDS bytes, 20x the atomic rate, and no single-precision loss inside a workgroup).
typedef double lds_acc_t;

template <typename F>
__device__ __forceinline__ void atomic_add(F *p, F v) {
    // -munsafe-fp-atomics: lowers to ds_add_f32/f64 (LDS) or global_atomic_add_f32/f64.
    atomicAdd(p, v);
}

template <typename F>
__device__ __forceinline__ F wave_sum(F v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// value of lane (lane ^ S) for S in {1, 2, 3, 4, 8} with DPP moves on the VALU (no LDS crossbar)
template <int S>
__device__ __forceinline__ int dpp_xor_i32(int v) {
    // mov_dpp: no `old` operand to initialise (every lane has a valid source for these controls)
    if constexpr (S == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    else if constexpr (S == 3) return __builtin_amdgcn_mov_dpp(v, 0x1B, 0xF, 0xF, true);  // quad_perm [3,2,1,0]
    else if constexpr (S == 4) {
        const int m = __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);                 // row_half_mirror: i ^ 7
        return __builtin_amdgcn_mov_dpp(m, 0x1B, 0xF, 0xF, true);                         // quad_perm [3,2,1,0]: ^ 3
    } else return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);                     // row_ror:8: i ^ 8
}
template <int S>
__device__ __forceinline__ double dpp_xor(double v) {
    return __hiloint2double(dpp_xor_i32<S>(__double2hiint(v)), dpp_xor_i32<S>(__double2loint(v)));
}
template <int S>
__device__ __forceinline__ float dpp_xor(float v) {
    return __int_as_float(dpp_xor_i32<S>(__float_as_int(v)));
}


__device__ __forceinline__ int64_t readfirstlane_i64(int64_t v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
    return (int64_t)(((unsigned long long)hi << 32) | lo);
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
template <int N, int K0 = 0, typename Fn>
__device__ __forceinline__ void static_for(Fn &&f) {
    if constexpr (K0 < N) {
        f(std::integral_constant<int, K0>{});
        static_for<N, K0 + 1>(f);
    }
}

// value of slot K (0..7) of the lane's own 8-lane group, broadcast to the 8 lanes; x4 must be
// dpp_xor_i32<4>(v).  Two DPP moves: quad_perm [k,k,k,k] of x4 for the lanes of the other quad,
// then the same pattern of v written only to the quads (DPP banks) that hold slot K themselves.
template <int K>
__device__ __forceinline__ int dpp_bcast8_i32(int v, int x4) {
    constexpr int QP = (K & 3) * 0x55;
    const int t = __builtin_amdgcn_mov_dpp(x4, QP, 0xF, 0xF, true);
    return __builtin_amdgcn_update_dpp(t, v, QP, 0xF, K < 4 ? 0x5 : 0xA, false);
}
template <int K>
__device__ __forceinline__ double dpp_bcast8(double v, double x4) {
    return __hiloint2double(dpp_bcast8_i32<K>(__double2hiint(v), __double2hiint(x4)),
                            dpp_bcast8_i32<K>(__double2loint(v), __double2loint(x4)));
}
template <int K>
__device__ __forceinline__ float dpp_bcast8(float v, float x4) {
    return __int_as_float(dpp_bcast8_i32<K>(__float_as_int(v), __float_as_int(x4)));
}

// Placement log (tm_tune_set("wg_log", device pointer to a WgLogBuf; 0 = off): the instrumented
// kernels append one record per workgroup -- which compute unit it ran on and when.  This is the
// evidence that kernels launched on different streams shared compute units
// (scripts/dev/coresidency.py, profiles/r3_coresidency.txt).
struct WgLog {
    unsigned long long hw, t0, t1, tag;   // XCC_ID << 32 | HW_ID, s_memrealtime start / end, kernel tag
};
struct WgLogBuf {
    unsigned long long count, cap, pad0, pad1;
    WgLog e[1];
};
enum WgTag { WG_SYRK_CO = 1, WG_K2 = 2, WG_CAT_DENSE = 3, WG_CAT_SPARSE = 4, WG_K3 = 5 };
inline WgLogBuf *wg_log_ptr() { return reinterpret_cast<WgLogBuf *>((uintptr_t)tune("wg_log", 0)); }
__device__ __forceinline__ unsigned long long wg_log_begin(const WgLogBuf *b) {
    return b ? wall_clock64() : 0ull;
}
// (call from ONE thread of the workgroup)
__device__ __forceinline__ void wg_log_end(WgLogBuf *b, unsigned long long t_begin, int tag) {
    if (!b) return;
    const unsigned long long i = atomicAdd(&b->count, 1ull);
    if (i >= b->cap) return;
    // HW_ID (hwreg 4): wave / simd / cu / sh / se ids; XCC_ID (hwreg 20): the die
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    WgLog r;
    r.hw = ((unsigned long long)xcc << 32) | hw;
    r.t0 = t_begin;
    r.t1 = wall_clock64();
    r.tag = (unsigned long long)tag;
    b->e[i] = r;
}

// column of a categorical code: code - drop_first, negative = contributes nothing
__device__ __forceinline__ int cat_col(int code, int drop_first) { return code - drop_first; }

}  // namespace tmh
