// Development aid: two alignments over the same window as ONE pass (dp_pair_proto.hip.inc, -DPAIR) against two passes of
// dp_forward2p_2c; best cells and every cell byte are compared through checksums.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mtr_amd/csrc -I tests/dev [-DPAIR] -o dp_pair_bench tests/dev/dp_pair_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dp_wrap.hip.inc"
#include "dp_pair_proto.hip.inc"
#ifndef WPS
#define WPS 8
#endif
// pairs of alignments over the same window: unit A = the read's own, unit B = another read's; PAIR = one pass for both
__global__ __launch_bounds__(64, WPS) void k_rows(const uint32_t *pk, const uint8_t *units, int UA, int UB, int rows, int reps, uint8_t *cells, size_t cells_per_wave, int *out)
{
    int acc = 0;
    uint8_t *codes = cells + (size_t)blockIdx.x * cells_per_wave;
    const uint32_t *p = pk + (size_t)(blockIdx.x & 1023) * 256;
    const uint8_t *uA = units + (size_t)(blockIdx.x & 1023) * 512, *uB = units + (size_t)((blockIdx.x + 7) & 1023) * 512;
    for (int r = 0; r < reps; r++) {
#ifdef PAIR
        int best[2][2][3];
        Dp2 q; q.U[0] = UA; q.U[1] = UB; q.unit[0] = uA; q.unit[1] = uB; q.rstride[0] = (UA + 3) & ~3; q.rstride[1] = (UB + 3) & ~3;
        q.off[0] = 0; q.off[1] = (size_t)rows * q.rstride[0];
        dp_forward2p_g2(p, 100 + r, rows, q, 1, 1, 3, 1, 3, 1, codes, best);
        for (int g = 0; g < 2; g++) acc += (best[g][0][0] * 3 + best[g][1][0] * 5 + best[g][0][1] * 7 + best[g][1][2] * 11 + best[g][0][2] * 13 + best[g][1][1] * 17) * (g + 1);
#else
        int bA[2][3], bB[2][3];
        dp_forward2p_2c(p, 100 + r, rows, uA, UA, 1, 1, 3, 1, 3, 1, codes, (UA + 1) & ~1, bA);
        dp_forward2p_2c(p, 100 + r, rows, uB, UB, 1, 1, 3, 1, 3, 1, codes + (size_t)rows * 132, (UB + 1) & ~1, bB);
        acc += (bA[0][0] * 3 + bA[1][0] * 5 + bA[0][1] * 7 + bA[1][2] * 11 + bA[0][2] * 13 + bA[1][1] * 17) * 1;
        acc += (bB[0][0] * 3 + bB[1][0] * 5 + bB[0][1] * 7 + bB[1][2] * 11 + bB[0][2] * 13 + bB[1][1] * 17) * 2;
#endif
        __builtin_amdgcn_wave_barrier();
    }
    if (lane_id() == 0) out[blockIdx.x] = acc * (int)(blockIdx.x / 1024 + 1);
}
// flags of the cells: a checksum over the matrices of wave 0..63 (same layout per alignment: compare per cell value)
int main(int argc, char **argv)
{
    const int UA = argc > 1 ? atoi(argv[1]) : 100, UB = argc > 2 ? atoi(argv[2]) : 97, rows = argc > 3 ? atoi(argv[3]) : 1000, reps = argc > 4 ? atoi(argv[4]) : 4;
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int waves = pr.multiProcessorCount * 4 * WPS;
    std::vector<uint32_t> pk(1024 * 256); std::vector<uint8_t> un(1024 * 512);
    srand(7);
    for (int r = 0; r < 1024; r++) {
        const int U = (r & 1) ? UB : UA;
        std::vector<int> unit(128); for (int j = 0; j < 128; j++) unit[j] = rand() & 3;
        for (int j = 0; j < 128; j++) un[(size_t)r * 512 + j] = (uint8_t)unit[j];
        for (int w = 0; w < 256; w++) { uint32_t v = 0; for (int b = 0; b < 16; b++) { int pos = w * 16 + b; int c = unit[pos % U]; if (rand() % 10 == 0) c = rand() & 3; v |= (uint32_t)c << (30 - 2 * b); } pk[(size_t)r * 256 + w] = v; }
    }
    uint32_t *dpk; uint8_t *dun, *dcells; int *dout;
    const size_t cpw = (size_t)(rows + 8) * 264 + 8192;
    (void)hipMalloc(&dpk, pk.size() * 4); (void)hipMalloc(&dun, un.size()); (void)hipMalloc(&dcells, cpw * waves); (void)hipMalloc(&dout, waves * 4);
    (void)hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dun, un.data(), un.size(), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_rows, dim3(waves), dim3(64), 0, 0, dpk, dun, UA, UB, rows, reps, dcells, cpw, dout);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < best) best = ms;
    }
    int chk = 0; std::vector<int> o(waves); (void)hipMemcpy(o.data(), dout, waves * 4, hipMemcpyDeviceToHost); for (int v : o) chk ^= v;
    // the cell bytes of the first waves' LAST repetition: per alignment a sum over (cell value x position), row strides removed
    std::vector<uint8_t> hc(cpw * 8); (void)hipMemcpy(hc.data(), dcells, cpw * 8, hipMemcpyDeviceToHost);
    unsigned long long csum = 0;
    for (int w = 0; w < 8; w++)
        for (int g = 0; g < 2; g++) {
            const int U = g ? UB : UA;
#ifdef PAIR
            const int rs = (U + 3) & ~3; const size_t off = g ? (size_t)rows * ((UA + 3) & ~3) : 0;
#else
            const int rs = (U + 1) & ~1; const size_t off = g ? (size_t)rows * 132 : 0;
#endif
            for (int i = 0; i < rows; i++) for (int j = 0; j < U; j++) csum = csum * 1000003ull + hc[(size_t)w * cpw + off + (size_t)i * rs + j];
        }
    const double rows_total = (double)waves * rows * reps * 2;            // alignment rows
    printf("U %d+%d rows %d, %d waves/SIMD: %.3f ms, %.1f cycles per ALIGNMENT row and SIMD at 2.1 GHz, best-cell check %08x, cells check %016llx\n",
           UA, UB, rows, WPS, best, best * 1e6 / (rows_total / (pr.multiProcessorCount * 4)) * 2.1, chk, csum);
    return 0;
}
