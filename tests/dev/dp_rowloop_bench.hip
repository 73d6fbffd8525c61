// Development aid: the row loop of the two-parameter forward pass (dp_forward2p) alone, at a chosen occupancy, to see what
// the chip sustains per row and what each part of the row costs (build variants of dp_wrap.hip.inc with parts removed).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<dir of the csrc copy> [-DWPS=8] -o dp_rowloop_bench dp_rowloop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dp_wrap.hip.inc"
#ifndef WPS
#define WPS 8
#endif
#ifndef NCHB
#define NCHB 2
#endif
__global__ __launch_bounds__(64, WPS) void k_rows(const uint32_t *pk, const uint8_t *units, int U, int rows, int reps, uint8_t *cells, size_t cells_per_wave, int *out)
{
    int best[2][3];
    int acc = 0;
    uint8_t *codes = cells + (size_t)blockIdx.x * cells_per_wave;
    const uint32_t *p = pk + (size_t)(blockIdx.x & 1023) * 256;              // 1024 different reads of 4096 bases
    const uint8_t *u = units + (size_t)(blockIdx.x & 1023) * 512;
    for (int r = 0; r < reps; r++) {
        dp_forward2p<NCHB>(p, 100 + r, rows, u, U, 1, 1, 3, 1, 3, 1, codes, best);
        acc += best[0][0] + best[1][0] + best[0][1] + best[1][2];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane_id() == 0) out[blockIdx.x] = acc;
}
int main(int argc, char **argv)
{
    const int U = argc > 1 ? atoi(argv[1]) : 100, rows = argc > 2 ? atoi(argv[2]) : 1000, reps = argc > 3 ? atoi(argv[3]) : 8;
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int waves = pr.multiProcessorCount * 4 * WPS;
    std::vector<uint32_t> pk(1024 * 256); std::vector<uint8_t> un(1024 * 512);
    srand(7);
    for (int r = 0; r < 1024; r++) {
        // a read that repeats a random unit of U bases with ~10 % substitutions, and that unit
        std::vector<int> unit(U); for (int j = 0; j < U; j++) unit[j] = rand() & 3;
        for (int j = 0; j < U; j++) un[(size_t)r * 512 + j] = (uint8_t)unit[j];
        for (int w = 0; w < 256; w++) { uint32_t v = 0; for (int b = 0; b < 16; b++) { int pos = w * 16 + b; int c = unit[pos % U]; if (rand() % 10 == 0) c = rand() & 3; v |= (uint32_t)c << (30 - 2 * b); } pk[(size_t)r * 256 + w] = v; }
    }
    uint32_t *dpk; uint8_t *dun, *dcells; int *dout;
    const size_t cpw = (size_t)(rows + 8) * (size_t)U + 4096;
    (void)hipMalloc(&dpk, pk.size() * 4); (void)hipMalloc(&dun, un.size()); (void)hipMalloc(&dcells, cpw * waves); (void)hipMalloc(&dout, waves * 4);
    (void)hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dun, un.data(), un.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rows, dim3(waves), dim3(64), 0, 0, dpk, dun, U, rows, reps, dcells, cpw, dout);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    int chk = 0; std::vector<int> o(waves); (void)hipMemcpy(o.data(), dout, waves * 4, hipMemcpyDeviceToHost); for (int v : o) chk ^= v;
    const double rows_total = (double)waves * rows * reps;
    printf("U %d rows %d, %d waves/SIMD (%d waves): %.3f ms, %.2f ns per row and SIMD (= %.1f cycles at 2.1 GHz), %.1f G cell pairs/s, check %08x\n",
           U, rows, WPS, waves, best, best * 1e6 / (rows_total / (pr.multiProcessorCount * 4)), best * 1e6 / (rows_total / (pr.multiProcessorCount * 4)) * 2.1, rows_total * U / best / 1e6, chk);
    return 0;
}
