// Deterministic second stage shared by all "row-slab -> LDS tile -> partial" kernels:
// every workgroup writes its privatised tile to the workspace, this kernel sums the
// partials of one tile in fixed order and writes / accumulates the result.
#pragma once
#include "common.hpp"

namespace tmh {

// ws layout: [n_parts][nb][stride]; out element = part * stride + e (bounded by out_elems).
// ACC is the accumulation type (double even for float data: partial sums of up to ~1e4
// rows each are combined without further single-precision loss).
// NW waves share the partials of 64 elements (wave w takes b = w, w + NW, ...; fixed tree at the
// end): 4 for large outputs, 16 for small ones -- a 256-bin histogram reduced by 4 blocks of 4
// waves was a serial chain of nb / 4 dependent loads (32 us for nothing).
template <typename F, bool ACCUMULATE, int NW>
__global__ __launch_bounds__(NW * 64) void reduce_partials_kernel(const F *__restrict__ ws,
                                                                  int64_t stride, int nb,
                                                                  F *__restrict__ out,
                                                                  int64_t out_elems) {
    __shared__ double red[NW][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t e = (int64_t)blockIdx.x * 64 + lane;
    const int part = blockIdx.y;
    double acc = 0.0;
    if (e < stride) {
        const F *p = ws + ((int64_t)part * nb) * stride + e;
        for (int b = wave; b < nb; b += NW) acc += (double)p[(int64_t)b * stride];
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < stride) {
        const int64_t o = (int64_t)part * stride + e;
        if (o < out_elems) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w += 4)
                s += (red[w][lane] + red[w + 1][lane]) + (red[w + 2][lane] + red[w + 3][lane]);
            if (ACCUMULATE)
                out[o] += (F)s;
            else
                out[o] = (F)s;
        }
    }
}

template <typename F>
inline int launch_reduce_partials(const F *ws, int64_t stride, int nb, int n_parts, F *out,
                                  int64_t out_elems, bool accumulate, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(stride, 64), (unsigned)n_parts);
    const bool small = (int64_t)grid.x * grid.y < 1024 && nb > 16;
    if (small) {
        if (accumulate)
            hipLaunchKernelGGL((reduce_partials_kernel<F, true, 16>), grid, dim3(1024), 0, st, ws,
                               stride, nb, out, out_elems);
        else
            hipLaunchKernelGGL((reduce_partials_kernel<F, false, 16>), grid, dim3(1024), 0, st, ws,
                               stride, nb, out, out_elems);
    } else if (accumulate) {
        hipLaunchKernelGGL((reduce_partials_kernel<F, true, 4>), grid, dim3(256), 0, st, ws, stride,
                           nb, out, out_elems);
    } else {
        hipLaunchKernelGGL((reduce_partials_kernel<F, false, 4>), grid, dim3(256), 0, st, ws, stride,
                           nb, out, out_elems);
    }
    TM_LAUNCH_CHECK();
    return TM_OK;
}

// tiny helpers used to prepare column maps / masks in the workspace
__global__ void fill_i32_kernel(int32_t *p, int64_t n, int32_t v);
__global__ void scatter_iota_i32_kernel(int32_t *map, const int32_t *cols, int64_t n_cols);

// map[c] = position of c in cols, or -1.  `map` must hold m entries.
int build_col_map(int32_t *map, int64_t m, const int32_t *cols, int64_t n_cols, hipStream_t st);

}  // namespace tmh
