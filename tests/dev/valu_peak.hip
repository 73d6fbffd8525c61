// Development aid: integer VALU issue rate of one SIMD on gfx950 as a function of waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned *out, int iters)
{
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) {
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));
            } else if (KIND == 1) {
                asm volatile("v_pk_add_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_add_u16 %3, %3, %8\n"
                             "v_pk_add_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n v_pk_add_u16 %6, %6, %8\n v_pk_add_u16 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0));
            } else {
                asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_u32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_max_u32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_u32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_max_u32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_u32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                             "v_max_u32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_max_u32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) ((unsigned long long *)(out + 1048576))[blockIdx.x] = t1 - t0;
}
template <int KIND> void run(const char *name, unsigned *d)
{
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * 4 * wps;     // one-wave workgroups: wps waves on every SIMD when the chip is full
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> t(blocks);
        hipMemcpy(t.data(), d + 1048576, blocks * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto x : t) avg += (double)x; avg /= blocks;
        double instr = (double)iters * 64.0;
        printf("%-14s waves/SIMD %d: %.2f cycles per instruction per wave -> SIMD issues one every %.2f cycles\n", name, wps, avg / instr, avg / instr / wps);
    }
}
int main()
{
    unsigned *d; hipMalloc(&d, (1048576 + 2 * 8192 * 2) * 4 + 1048576);
    run<0>("v_add_u32", d); run<1>("v_pk_add_u16", d); run<2>("v_max_u32_dpp", d);
    return 0;
}
