// All categorical x categorical blocks of a SplitMatrix sandwich (and the categorical diagonals)
// in ONE pass over the codes (reference: ext/split.pyx:83-111 sandwich_cat_cat ->
// ext/cat_split_helpers-tmpl.cpp:44-94, and ext/categorical.pyx:183-218 for the diagonals).
//
// A design with k categoricals has k (k - 1) / 2 pair tables.  One launch per table (hist_lds_kernel)
// is ~30 us of launches for a few KB of work: 30 categoricals of 12 levels = 435 tables = 11.7 ms for
// 2M rows whose codes are 0.26 GB.  Here the tables are packed into BUNDLES that fit one LDS tile
// of doubles (16 384 bins).  A bundle is a RECTANGLE of tables: a list A and a list B of up to 16
// categoricals each, all tables (a, b); its tile is [sum of the A levels] x [sum of the B levels],
// so that a row's bin in table (a, b) is  ya[a] + xb[b]  with  ya[a] = (first tile row of a +
// code_a) * tile width,  xb[b] = first tile column of b + code_b:  |A| + |B| loads and conversions
// per row, then ONE add and one ds_add_f64 per table, everything statically indexed in registers.
// Modes: 0 = A x B;  1 = "triangle": B is A, tables a <= b (the a == a table's diagonal is the
// categorical's own diagonal);  2 = diagonals only (tile = one row of bins per categorical).
// Workgroups are dealt to bundles by the host plan in proportion to their table counts (wg_map);
// a second kernel sums the workgroup tiles, a third scatters every table (and its mirror) into the
// p x p result.
//
// (first version: arbitrary table lists per bundle, codes in a register vector indexed with the
// table's (i, j): v_movrels per operand, 2.3 ms for 20 categoricals x 50 levels at 2M rows, of
// which 1.3 ms remained with neither loads nor atomics.)
#include "common.hpp"
#include "reduce.hpp"

namespace tmh {

constexpr int CP_BINS = 16384;          // doubles per bundle tile (128 KB)
constexpr int CP_SLOTS = 16;            // categoricals per side of a bundle
constexpr int CP_ROW = 8 + 4 * CP_SLOTS;   // int32 words per bundle row
constexpr int CP_NEG = -(1 << 28);      // "no level here": any sum with it stays negative

// bundle row: [na, nb, mode, tile width, first workgroup, workgroups, bins, 0,
//              A: {categorical, ya offset} x 16,  B: {categorical, xb offset} x 16]
template <typename F, int NS>
__global__ __launch_bounds__(1024) void multi_cat_pairs_kernel(
    const int64_t *__restrict__ cat_tab, const F *__restrict__ d, const int32_t *__restrict__ rows,
    int64_t n_iter, const int32_t *__restrict__ bundles, const int32_t *__restrict__ wg_map,
    int max_bins, double *__restrict__ ws) {
    extern __shared__ double tile[];
    const int y = __builtin_amdgcn_readfirstlane(wg_map[blockIdx.x]);
    const int32_t *br = bundles + (int64_t)y * CP_ROW;
    const int na = __builtin_amdgcn_readfirstlane(br[0]), nb = __builtin_amdgcn_readfirstlane(br[1]);
    const int mode = __builtin_amdgcn_readfirstlane(br[2]), lbs = __builtin_amdgcn_readfirstlane(br[3]);
    const int wg0 = __builtin_amdgcn_readfirstlane(br[4]);
    const int bins = __builtin_amdgcn_readfirstlane(br[6]);
    // the bundle's row range is cut into at most br[5] parts of >= 8192 rows
    const int parts = (int)min((int64_t)__builtin_amdgcn_readfirstlane(br[5]),
                               max((int64_t)1, (n_iter + 8191) / 8192));
    const int part = (int)blockIdx.x - wg0;
    if (part >= parts) return;
    for (int b = threadIdx.x; b < bins; b += blockDim.x) tile[b] = 0.0;
    __syncthreads();
    const int32_t *pa[NS], *pb[NS];
    int dropa[NS], dropb[NS], offa[NS], offb[NS];
    static_for<NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        pa[s] = pb[s] = nullptr;
        dropa[s] = dropb[s] = offa[s] = offb[s] = 0;
        if (s < na) {
            const int id = __builtin_amdgcn_readfirstlane(br[8 + 2 * s]);
            offa[s] = __builtin_amdgcn_readfirstlane(br[9 + 2 * s]);
            pa[s] = reinterpret_cast<const int32_t *>(cat_tab[4 * id]);
            dropa[s] = (int)cat_tab[4 * id + 1];
        }
        if (s < nb) {
            const int id = __builtin_amdgcn_readfirstlane(br[8 + 2 * CP_SLOTS + 2 * s]);
            offb[s] = __builtin_amdgcn_readfirstlane(br[9 + 2 * CP_SLOTS + 2 * s]);
            pb[s] = reinterpret_cast<const int32_t *>(cat_tab[4 * id]);
            dropb[s] = (int)cat_tab[4 * id + 1];
        }
    });
    constexpr int UNR = NS <= 8 ? 4 : 2;           // rows per thread and step (loads in flight)
    const int64_t rpp = (n_iter + parts - 1) / parts;
    const int64_t t0 = (int64_t)part * rpp, t1 = min(t0 + rpp, n_iter);
    for (int64_t tb = t0; tb < t1; tb += (int64_t)UNR * blockDim.x) {
        int64_t k[UNR], kc[UNR];
        double dk[UNR];
        int ya[UNR][NS], xb[UNR][NS];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t t = tb + (int64_t)u * blockDim.x + threadIdx.x;
            k[u] = t < t1 ? (rows ? (int64_t)rows[t] : t) : -1;
        }
        // ALL loads of the step are issued before the first one is used (row index clamped instead
        // of a lane branch around the load: a guarded load is waited for inside its branch, which
        // made the categoricals of a bundle that many dependent memory round trips)
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            kc[u] = k[u] >= 0 ? k[u] : 0;
            dk[u] = (double)d[kc[u]];
        }
        static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s < na) {
#pragma unroll
                for (int u = 0; u < UNR; ++u) ya[u][s] = __builtin_nontemporal_load(pa[s] + kc[u]);
            }
        });
        if (mode == 0) {
            static_for<NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if (s < nb) {
#pragma unroll
                    for (int u = 0; u < UNR; ++u) xb[u][s] = __builtin_nontemporal_load(pb[s] + kc[u]);
                }
            });
        }
        bool live[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (k[u] < 0) dk[u] = 0.0;
            live[u] = dk[u] != 0.0;
        }
        static_for<NS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ca = ya[u][s] - dropa[s];
                const int cb = (mode == 0 ? xb[u][s] : ya[u][s]) - dropb[s];
                ya[u][s] = (s < na && live[u] && ca >= 0) ? offa[s] + ca * lbs : CP_NEG;
                xb[u][s] = mode == 2 ? 0 : ((s < nb && cb >= 0) ? offb[s] + cb : CP_NEG);
            }
        });
        static_for<NS>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            if (a < na) {
                static_for<NS>([&](auto bc) {
                    constexpr int b = decltype(bc)::value;
                    // mode 0: every (a, b);  1: a <= b;  2: a == b
                    const bool on = b == a ? (mode != 0 || b < nb)
                                           : (b < nb && (mode == 0 || (mode == 1 && b > a)));
                    if (on) {
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            const int bin = ya[u][a] + xb[u][b];
                            if (bin >= 0) atomicAdd(tile + bin, dk[u]);
                        }
                    }
                });
            }
        });
    }
    __syncthreads();
    double *dst = ws + (int64_t)blockIdx.x * max_bins;
    for (int b = threadIdx.x; b < bins; b += blockDim.x) dst[b] = tile[b];
}

// tables[y * max_bins + e] = sum of the bundle's workgroup tiles (64 bins x 4 slices of the
// workgroup list per block)
__global__ __launch_bounds__(256) void multi_cat_pairs_reduce_kernel(
    const double *__restrict__ ws, const int32_t *__restrict__ bundles, int64_t n_iter, int max_bins,
    double *__restrict__ tables) {
    __shared__ double part_sum[4][64];
    const int32_t *br = bundles + (int64_t)blockIdx.x * CP_ROW;
    const int wg0 = br[4], bins = br[6];
    const int parts = (int)min((int64_t)br[5], max((int64_t)1, (n_iter + 8191) / 8192));
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    for (int e0 = blockIdx.y * 64; e0 < bins; e0 += gridDim.y * 64) {
        const int e = e0 + lane;
        double s0 = 0.0, s1 = 0.0;
        if (e < bins) {
            int q = sl;
            for (; q + 4 < parts; q += 8) {
                s0 += ws[(int64_t)(wg0 + q) * max_bins + e];
                s1 += ws[(int64_t)(wg0 + q + 4) * max_bins + e];
            }
            if (q < parts) s0 += ws[(int64_t)(wg0 + q) * max_bins + e];
        }
        part_sum[sl][lane] = s0 + s1;
        __syncthreads();
        if (sl == 0 && e < bins)
            tables[(int64_t)blockIdx.x * max_bins + e] =
                (part_sum[0][lane] + part_sum[1][lane]) + (part_sum[2][lane] + part_sum[3][lane]);
        __syncthreads();
    }
}

// desc[table] = {offset in `tables`, L_i, L_j, tile width (diagonal: element stride), first position
//                of block i, of block j, diagonal flag, 0}
__global__ __launch_bounds__(256) void multi_cat_pairs_scatter_kernel(
    const double *__restrict__ tables, const int64_t *__restrict__ desc,
    const int64_t *__restrict__ pos, double *__restrict__ out, int64_t p) {
    const int64_t *q = desc + (int64_t)blockIdx.x * 8;
    const int64_t toff = q[0], li = q[1], lj = q[2], w = q[3], pi = q[4], pj = q[5];
    const bool diag = q[6] != 0;
    const int64_t total = diag ? li : li * lj;
    for (int64_t e = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.y * blockDim.x) {
        if (diag) {
            const int64_t r = pos[pi + e];
            if (r >= 0) out[r * p + r] = tables[toff + e * w];
        } else {
            const int64_t ei = e / lj, ej = e % lj;
            const double v = tables[toff + ei * w + ej];
            const int64_t r = pos[pi + ei], cc = pos[pj + ej];
            if (r >= 0 && cc >= 0) {
                out[r * p + cc] = v;
                out[cc * p + r] = v;
            }
        }
    }
}

// out[k] += sum over the categoricals of v[pos[first position of c + code_c(k) - first kept code]]
// (SplitMatrix.matvec over all categorical blocks at once; reference: split_matrix.py:373-420 calling
// ext/categorical.pyx:110-136 matvec_fast block by block).  Lane <-> row, 4 rows per thread; the
// categoricals' descriptors sit in registers (lane c of a page = categorical c) and are read back
// with v_readlane; two categoricals per step = 8 code loads, then 8 + 8 dependent gathers in flight.
template <typename F>
__global__ __launch_bounds__(256) void multi_cat_matvec_kernel(
    const int64_t *__restrict__ cat_tab, int n_cats, const int64_t *__restrict__ pos,
    const F *__restrict__ v, int64_t n, F *__restrict__ out) {
    constexpr int UNR = 4;
    const int lane = threadIdx.x & 63;
    int64_t k[UNR];
    double acc[UNR];
    const int64_t base = (int64_t)blockIdx.x * blockDim.x * UNR + threadIdx.x;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int64_t t = base + (int64_t)u * blockDim.x;
        k[u] = t < n ? t : n - 1;
        acc[u] = 0.0;
    }
    for (int c0 = 0; c0 < n_cats; c0 += 64) {
        const int cnt = min(n_cats - c0, 64);
        const int cl = min(c0 + lane, n_cats - 1);
        const int64_t ptr = cat_tab[4 * cl], poff = cat_tab[4 * cl + 2];
        const int plo = (int)(uint32_t)ptr, phi = (int)(uint32_t)((uint64_t)ptr >> 32);
        const int drop = (int)cat_tab[4 * cl + 1];
        const int olo = (int)(uint32_t)poff, ohi = (int)(uint32_t)((uint64_t)poff >> 32);
        auto desc = [&](int c, const int32_t *&codes, int &dr, int64_t &po) {
            const uint64_t p = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(plo, c) |
                               ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(phi, c) << 32);
            codes = reinterpret_cast<const int32_t *>(p);
            dr = __builtin_amdgcn_readlane(drop, c);
            po = (int64_t)((uint64_t)(uint32_t)__builtin_amdgcn_readlane(olo, c) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(ohi, c) << 32));
        };
        int c = 0;
        for (; c + 2 <= cnt; c += 2) {
            const int32_t *ca, *cb;
            int da, db;
            int64_t pa, pb;
            desc(c, ca, da, pa);
            desc(c + 1, cb, db, pb);
            int ia[UNR], ib[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                ia[u] = __builtin_nontemporal_load(ca + k[u]) - da;
                ib[u] = __builtin_nontemporal_load(cb + k[u]) - db;
            }
            int64_t qa[UNR], qb[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                qa[u] = pos[pa + max(ia[u], 0)];
                qb[u] = pos[pb + max(ib[u], 0)];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const double xa = (double)v[qa[u]], xb = (double)v[qb[u]];
                acc[u] += (ia[u] >= 0 ? xa : 0.0) + (ib[u] >= 0 ? xb : 0.0);
            }
        }
        if (c < cnt) {
            const int32_t *ca;
            int da;
            int64_t pa;
            desc(c, ca, da, pa);
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int ia = __builtin_nontemporal_load(ca + k[u]) - da;
                const double xa = (double)v[pos[pa + max(ia, 0)]];
                acc[u] += ia >= 0 ? xa : 0.0;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int64_t t = base + (int64_t)u * blockDim.x;
        if (t < n) out[t] += (F)acc[u];
    }
}

template <typename F>
static int run_multi_cat_matvec(const int64_t *cat_tab, int n_cats, const int64_t *pos, const F *v,
                                int64_t n, F *out, hipStream_t st) {
    if (n <= 0 || n_cats <= 0) return TM_OK;
    const int64_t nblk = ceil_div(n, 256 * 4);
    prof_begin(st);
    hipLaunchKernelGGL((multi_cat_matvec_kernel<F>), dim3((unsigned)nblk), dim3(256), 0, st, cat_tab,
                       n_cats, pos, v, n, out);
    prof_end(st);
    TM_LAUNCH_CHECK();
    return TM_OK;
}

template <typename F>
static int run_multi_cat_pairs(const int64_t *cat_tab, int64_t n, const F *d, const int32_t *rows,
                               int64_t n_rows, const int32_t *bundles, int n_bundles,
                               const int32_t *wg_map, int n_wg, int slots, int64_t bins,
                               const int64_t *desc, int64_t n_pairs, const int64_t *pos,
                               double *tables, double *out, int64_t p, hipStream_t st) {
    if (bins < 1 || bins > CP_BINS || n_bundles < 1 || n_wg < n_bundles || slots < 1 ||
        slots > CP_SLOTS) {
        set_error("tm_multi_cat_pairs: 1..%d bins per bundle, 1..%d categoricals per side, one "
                  "workgroup per bundle at least", CP_BINS, CP_SLOTS);
        return TM_EINVAL;
    }
    const int64_t n_iter = rows ? n_rows : n;
    if (n_iter > 0) {
        void *wsv = nullptr;
        int rc = get_workspace(sizeof(double) * (size_t)(n_wg * bins) + 256, &wsv, st);
        if (rc) return rc;
        double *ws = reinterpret_cast<double *>(wsv);
        const size_t lds = sizeof(double) * (size_t)bins;
        auto go = [&](auto kern) -> int {
            if (lds > 48 * 1024)
                TM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            prof_begin(st);
            hipLaunchKernelGGL(kern, dim3((unsigned)n_wg), dim3(1024), lds, st, cat_tab, d, rows,
                               n_iter, bundles, wg_map, (int)bins, ws);
            prof_end(st);
            TM_LAUNCH_CHECK();
            return TM_OK;
        };
        if (slots <= 4) rc = go(&multi_cat_pairs_kernel<F, 4>);
        else if (slots <= 8) rc = go(&multi_cat_pairs_kernel<F, 8>);
        else rc = go(&multi_cat_pairs_kernel<F, 16>);
        if (rc) return rc;
        const unsigned ry = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(bins, 64), 2048 / n_bundles));
        hipLaunchKernelGGL(multi_cat_pairs_reduce_kernel, dim3((unsigned)n_bundles, ry), dim3(256), 0, st,
                           ws, bundles, n_iter, (int)bins, tables);
        TM_LAUNCH_CHECK();
    } else {
        TM_HIP(hipMemsetAsync(tables, 0, sizeof(double) * (size_t)(n_bundles * bins), st));
    }
    if (out != nullptr && n_pairs > 0) {
        hipLaunchKernelGGL(multi_cat_pairs_scatter_kernel, dim3((unsigned)n_pairs, 4), dim3(256), 0, st,
                           tables, desc, pos, out, p);
        TM_LAUNCH_CHECK();
    }
    return TM_OK;
}

}  // namespace tmh

extern "C" {
int tm_multi_cat_pairs_max_bins(void) { return tmh::CP_BINS; }
int tm_multi_cat_pairs_max_slots(void) { return tmh::CP_SLOTS; }
int tm_multi_cat_pairs_row_words(void) { return tmh::CP_ROW; }
int tm_multi_cat_pairs_f32(const int64_t *cat_tab, int64_t n, const float *d, const int32_t *rows,
                           int64_t n_rows, const int32_t *bundles, int n_bundles,
                           const int32_t *wg_map, int n_wg, int slots, int64_t bins,
                           const int64_t *desc, int64_t n_pairs, const int64_t *pos, double *tables,
                           double *out, int64_t p, void *stream) {
    return tmh::run_multi_cat_pairs<float>(cat_tab, n, d, rows, n_rows, bundles, n_bundles, wg_map, n_wg,
                                           slots, bins, desc, n_pairs, pos, tables, out, p,
                                           tmh::as_stream(stream));
}
int tm_multi_cat_pairs_f64(const int64_t *cat_tab, int64_t n, const double *d, const int32_t *rows,
                           int64_t n_rows, const int32_t *bundles, int n_bundles,
                           const int32_t *wg_map, int n_wg, int slots, int64_t bins,
                           const int64_t *desc, int64_t n_pairs, const int64_t *pos, double *tables,
                           double *out, int64_t p, void *stream) {
    return tmh::run_multi_cat_pairs<double>(cat_tab, n, d, rows, n_rows, bundles, n_bundles, wg_map, n_wg,
                                            slots, bins, desc, n_pairs, pos, tables, out, p,
                                            tmh::as_stream(stream));
}
int tm_multi_cat_matvec_f32(const int64_t *cat_tab, int n_cats, const int64_t *pos, const float *v,
                            int64_t n, float *out, void *stream) {
    return tmh::run_multi_cat_matvec<float>(cat_tab, n_cats, pos, v, n, out, tmh::as_stream(stream));
}
int tm_multi_cat_matvec_f64(const int64_t *cat_tab, int n_cats, const int64_t *pos, const double *v,
                            int64_t n, double *out, void *stream) {
    return tmh::run_multi_cat_matvec<double>(cat_tab, n_cats, pos, v, n, out, tmh::as_stream(stream));
}
}  // extern "C"
