// Probe (round 4): VGPR index mode with 32-bit operands -- does every index value 0..31 select v[64 + idx]?
// Each lane adds (1000 * idx + lane) to v[64 + idx] under s_set_gpr_idx_on with v_fmac_f32_dpp / v_add_f32 /
// v_mov_b32; the host checks which register received it.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __attribute__((amdgpu_num_vgpr(40))) void probe(float *out, int mode) {
    const int lane = threadIdx.x;
    asm volatile("v_mov_b32 v95, 0" ::: "v95");
    for (int k = 0; k < 32; ++k) asm volatile("s_set_gpr_idx_on %0, 0x8\n\tv_mov_b32 v64, 0\n\ts_set_gpr_idx_off" :: "s"(k) : "m0");
    for (int idx = 0; idx < 32; ++idx) {
        float a = 1.0f, x = 1000.0f * idx + lane;
        if (mode == 0)
            asm volatile("s_nop 1\n\ts_set_gpr_idx_on %0, 0xc\n\tv_fmac_f32_dpp v64, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\ts_set_gpr_idx_off"
                         :: "s"(idx), "v"(a), "v"(x) : "m0");
        else if (mode == 1)
            asm volatile("s_set_gpr_idx_on %0, 0xc\n\tv_fmac_f32 v64, %1, %2\n\ts_set_gpr_idx_off" :: "s"(idx), "v"(a), "v"(x) : "m0");
        else if (mode == 2)
            asm volatile("s_set_gpr_idx_on %0, 0x8\n\tv_mov_b32 v64, %2\n\ts_set_gpr_idx_off" :: "s"(idx), "v"(a), "v"(x) : "m0");
        else if (mode == 3)      // mode switched on with index 0, then s_set_gpr_idx_idx
            asm volatile("s_set_gpr_idx_on %3, 0xc\n\ts_set_gpr_idx_idx %0\n\tv_fmac_f32 v64, %1, %2\n\ts_set_gpr_idx_off" :: "s"(idx), "v"(a), "v"(x), "s"(0) : "m0");
        else                     // ... with an LDS read between the index change and the FMA
            asm volatile("s_set_gpr_idx_on %3, 0xc\n\ts_set_gpr_idx_idx %0\n\tds_read_b32 v40, %4\n\ts_waitcnt lgkmcnt(0)\n\tv_fmac_f32 v64, %1, %2\n\tv_add_f32 v64, v64, v40\n\ts_set_gpr_idx_off" :: "s"(idx), "v"(a), "v"(x), "s"(0), "v"(0) : "m0", "v40");
    }
    float r[32];
#define RD(K) asm volatile("v_mov_b32 %0, v[64+" #K "]" : "=v"(r[K]));
    RD(0) RD(1) RD(2) RD(3) RD(4) RD(5) RD(6) RD(7) RD(8) RD(9) RD(10) RD(11) RD(12) RD(13) RD(14) RD(15)
    RD(16) RD(17) RD(18) RD(19) RD(20) RD(21) RD(22) RD(23) RD(24) RD(25) RD(26) RD(27) RD(28) RD(29) RD(30) RD(31)
    for (int k = 0; k < 32; ++k) out[k * 64 + lane] = r[k];
}
int main() {
    float *out; hipMalloc(&out, 4 * 32 * 64);
    float h[32 * 64];
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, mode);
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (%s): register k holds (lane 5): ", mode, mode == 0 ? "fmac_f32_dpp" : mode == 1 ? "fmac_f32" : mode == 2 ? "mov_b32" : mode == 3 ? "idx_idx + fmac" : "idx_idx + ds_read + fmac + add");
        for (int k = 0; k < 32; ++k) printf("%g ", h[k * 64 + 5]);
        printf("\n");
    }
    return 0;
}
