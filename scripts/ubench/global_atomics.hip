// Global (L2) float atomics on gfx950: f32 vs f64 rate, random addresses in a 1M-element array and
// the conflict-heavy case of 64 addresses.  hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
template <typename T>
__global__ void k(T *out, int iters, unsigned mask) {
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        atomicAdd(&out[(s >> 8) & mask], (T)1);
    }
}
template <typename T>
void run(const char *name, unsigned mask) {
    T *out; hipMalloc(&out, sizeof(T) * (mask + 1)); hipMemset(out, 0, sizeof(T) * (mask + 1));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 2048, threads = 256, iters = 256;
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, out, 8, mask);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(threads), 0, 0, out, iters, mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %8.2f G atomics/s\n", name, (double)blocks * threads * iters / ms / 1e6);
    hipFree(out);
}
int main() {
    run<float>("f32, 1M addresses", (1u << 20) - 1);
    run<double>("f64, 1M addresses", (1u << 20) - 1);
    run<float>("f32, 4096 addresses", 4095);
    run<double>("f64, 4096 addresses", 4095);
    run<float>("f32, 64 addresses", 63);
    run<double>("f64, 64 addresses", 63);
    return 0;
}
