// Calibration of the FETCH_SIZE counter (rocprofv3 --pmc FETCH_SIZE) against access patterns whose HBM bytes are
// known: every kernel reads each of its bytes exactly once from a 4 GB buffer (16x the 256 MB Infinity Cache, so
// nothing is served from a cache), in segments of SEG bytes at a stride of 256 bytes.
//
//   calib_stream   whole 128-byte lines, consecutive (the pattern of K1e / cat x dense / the K3 slab copies)
//   calib_seg<64>  64-byte segments  (K2b: 8 entries x 8 bytes of one row in one chunk)
//   calib_seg<32>  32-byte segments  (K2b: 8 column indices x 4 bytes)
//   calib_seg<16>  16-byte segments  (K2b: block descriptors, when not consecutive)
//
// useful bytes per launch are printed; FETCH_SIZE (KiB) / useful KiB is the factor the counter has to be multiplied
// by for that pattern.  build + run (on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib scripts/ubench/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE -d /tmp/fc -o k --output-format csv -- /tmp/fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#define CHECK(x)                                                                   \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));           \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void calib_stream(const double *__restrict__ buf, int64_t n_words, double *__restrict__ sink) {
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x)
        acc += buf[i];
    if (acc == 12345.678) sink[0] = acc;
}

// segment s = bytes [256 s, 256 s + SEG): SEG / 8 lanes each read 8 bytes of it
template <int SEG>
__global__ void calib_seg(const double *__restrict__ buf, int64_t n_seg, double *__restrict__ sink) {
    constexpr int LPS = SEG / 8;     // lanes per segment
    double acc = 0.0;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t t = t0; t < n_seg * LPS; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = t / LPS;
        const int k = (int)(t % LPS);
        acc += buf[s * 32 + k];
    }
    if (acc == 12345.678) sink[0] = acc;
}

int main() {
    const int64_t bytes = 4ll << 30;
    const int64_t n_words = bytes / 8, n_seg = bytes / 256;
    double *buf = nullptr, *sink = nullptr;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 0, bytes));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 8, block = 256;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(calib_stream, dim3(grid), dim3(block), 0, 0, buf, n_words, sink);
        hipLaunchKernelGGL(calib_seg<64>, dim3(grid), dim3(block), 0, 0, buf, n_seg, sink);
        hipLaunchKernelGGL(calib_seg<32>, dim3(grid), dim3(block), 0, 0, buf, n_seg, sink);
        hipLaunchKernelGGL(calib_seg<16>, dim3(grid), dim3(block), 0, 0, buf, n_seg, sink);
        CHECK(hipDeviceSynchronize());
    }
    std::printf("useful KiB per launch: calib_stream %lld  calib_seg<64> %lld  calib_seg<32> %lld  calib_seg<16> %lld\n",
                (long long)(bytes >> 10), (long long)(n_seg * 64 >> 10), (long long)(n_seg * 32 >> 10),
                (long long)(n_seg * 16 >> 10));
    CHECK(hipFree(buf));
    CHECK(hipFree(sink));
    return 0;
}
