// Development aid (round 5): dp_forward1p_h8 (EIGHT one-parameter DPs per wavefront, two in the halves of every register) against
// dp_forward1p_g16<., 16, 4, true> (four per wavefront): identical cells (by the traceback's reading of the flags) and best cells, and the time per DP row.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../../mtr_amd/csrc -I../../../include [-DWPS=4] -o dp_oct_bench dp_oct_bench.hip
//   ./dp_oct_bench U rows reps [spreadU] [spreadRows] [mixed scores 0/1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dp_wrap.hip.inc"
#include "dp_quad.hip.inc"
#ifndef WPS
#define WPS 4
#endif
struct Job { int read, base, rows, U, G, MM, D; };
template <int C>
__device__ void quad_pass(const uint32_t *pk, const uint8_t *units, const Job *jobs, int first, uint8_t *cb, int maxrows, int (&best)[4][3])
{
    DpQuad q; q.n = 4;
    for (int g = 0; g < 4; g++) {
        const Job jb = jobs[first + g];
        const int rd = uni(jb.read);
        q.pk[g] = pk + (size_t)rd * 256; q.wlim[g] = 256; q.base[g] = uni(jb.base); q.rows[g] = uni(jb.rows); q.U[g] = uni(jb.U);
        q.unit[g] = units + (size_t)rd * 512; q.G[g] = uni(jb.G); q.MM[g] = uni(jb.MM); q.D[g] = uni(jb.D);
    }
    dp_forward1p_g16<C, 16, DPQ_SCALE, true>(q, cb, maxrows, best);
}
// wavefront w: jobs 8w .. 8w+7 as two four-per-wavefront passes, every DP's matrix at cb + slot * DPQP_DP_BYTES(C, maxrows) (slot = 0..7)
template <int C>
__global__ __launch_bounds__(64, WPS) void k_quad(const uint32_t *pk, const uint8_t *units, const Job *jobs, int reps, uint8_t *cells, size_t cells_per_wave, int *out)
{
    int acc = 0;
    int maxrows = 0;
    for (int g = 0; g < 8; g++) { const int r = uni(jobs[blockIdx.x * 8 + g].rows); maxrows = r > maxrows ? r : maxrows; }
    for (int r = 0; r < reps; r++) {
        uint8_t *cb = cells + (size_t)blockIdx.x * cells_per_wave;
        for (int half = 0; half < 2; half++) {
            int best[4][3];
            quad_pass<C>(pk, units, jobs, blockIdx.x * 8 + 4 * half, cb + (size_t)(4 * half) * DPQP_DP_BYTES(C, maxrows), maxrows, best);
            if (lane_id() == 0 && r == 0) for (int g = 0; g < 4; g++) for (int k = 0; k < 3; k++) out[(blockIdx.x * 8 + 4 * half + g) * 4 + k] = best[g][k];
            acc += best[0][0];
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane_id() == 0) out[blockIdx.x * 32 + 3] = acc;
}
template <int C>
__global__ __launch_bounds__(64, WPS) void k_oct(const uint32_t *pk, const uint8_t *units, const Job *jobs, int reps, uint8_t *cells, size_t cells_per_wave, int *out)
{
    int acc = 0;
    for (int r = 0; r < reps; r++) {
        DpOct q; q.n = 8;
        int maxrows = 0;
        for (int g = 0; g < 8; g++) {
            const Job jb = jobs[blockIdx.x * 8 + g];
            const int rd = uni(jb.read);
            q.pk[g] = pk + (size_t)rd * 256; q.wlim[g] = 256; q.base[g] = uni(jb.base); q.rows[g] = uni(jb.rows); q.U[g] = uni(jb.U);
            q.unit[g] = units + (size_t)rd * 512; q.G[g] = uni(jb.G); q.MM[g] = uni(jb.MM); q.D[g] = uni(jb.D);
            maxrows = q.rows[g] > maxrows ? q.rows[g] : maxrows;
        }
        int best[8][3];
        dp_forward1p_h8<C>(q, cells + (size_t)blockIdx.x * cells_per_wave, maxrows, best);
        if (lane_id() == 0 && r == 0) for (int g = 0; g < 8; g++) for (int k = 0; k < 3; k++) out[(blockIdx.x * 8 + g) * 4 + k] = best[g][k];
        acc += best[0][0];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane_id() == 0) out[blockIdx.x * 32 + 3] = acc;
}
template <int C> void launch(int which, int waves, const uint32_t *pk, const uint8_t *un, const Job *j, int reps, uint8_t *cells, size_t cpw, int *out)
{
    if (which == 0) hipLaunchKernelGGL(k_quad<C>, dim3(waves), dim3(64), 0, 0, pk, un, j, reps, cells, cpw, out);
    else hipLaunchKernelGGL(k_oct<C>, dim3(waves), dim3(64), 0, 0, pk, un, j, reps, cells, cpw, out);
}
static void launch_c(int cq, int which, int waves, const uint32_t *pk, const uint8_t *un, const Job *j, int reps, uint8_t *cells, size_t cpw, int *out)
{
    switch (cq) {
    case 1: launch<1>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    case 2: launch<2>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    case 3: launch<3>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    case 4: launch<4>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    case 5: launch<5>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    case 6: launch<6>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    case 7: launch<7>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    default: launch<8>(which, waves, pk, un, j, reps, cells, cpw, out); break;
    }
}
static int canon(int f) { return (f & 8) ? 8 : !(f & 1) ? 0 : !(f & 2) ? 1 : !(f & 4) ? 3 : 7; }
int main(int argc, char **argv)
{
    const int U = argc > 1 ? atoi(argv[1]) : 100, rows = argc > 2 ? atoi(argv[2]) : 1000, reps = argc > 3 ? atoi(argv[3]) : 4;
    const int spreadU = argc > 4 ? atoi(argv[4]) : 0, spreadR = argc > 5 ? atoi(argv[5]) : 0, mixed = argc > 6 ? atoi(argv[6]) : 1;
    const int cq = (U + 15) / 16;
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int waves = pr.multiProcessorCount * 4 * WPS, njobs = waves * 8;
    std::vector<uint32_t> pk(1024 * 256); std::vector<uint8_t> un(1024 * 512);
    std::vector<int> runit(1024);
    srand(7);
    for (int r = 0; r < 1024; r++) {
        int Ur = U - (spreadU ? rand() % (spreadU + 1) : 0); if (Ur < 2) Ur = 2;
        if (cq > 1 && Ur <= 16 * (cq - 1)) Ur = 16 * (cq - 1) + 1;          // (one columns-per-lane class per run)
        runit[r] = Ur;
        std::vector<int> unit(Ur); for (int j = 0; j < Ur; j++) unit[j] = rand() & 3;
        for (int j = 0; j < Ur; j++) un[(size_t)r * 512 + j] = (uint8_t)unit[j];
        for (int w = 0; w < 256; w++) { uint32_t v = 0; for (int b = 0; b < 16; b++) { int pos = w * 16 + b; int c = unit[pos % Ur]; if (rand() % 10 == 0) c = rand() & 3; v |= (uint32_t)c << (30 - 2 * b); } pk[(size_t)r * 256 + w] = v; }
    }
    std::vector<Job> jobs(njobs);
    for (int j = 0; j < njobs; j++) {
        Job &b = jobs[j]; b.read = rand() & 1023; b.base = 90 + rand() % 40; b.rows = rows - (spreadR ? rand() % (spreadR + 1) : 0); if (b.rows < 1) b.rows = 1; b.U = runit[b.read];
        if (b.base + b.rows + 20 > 4096) b.rows = 4096 - 20 - b.base;
        const int p = mixed ? rand() & 1 : 0; b.G = p ? 1 : 5; b.MM = 1; b.D = p ? 3 : 1;          // the revision's two parameter sets (consensus.c:1055, :1070)
    }
    uint32_t *dpk; uint8_t *dun, *dcA, *dcB; int *doA, *doB; Job *dj;
    const size_t cpw = 8 * DPQP_DP_BYTES(8, rows + 8) + 4096;
    (void)hipMalloc(&dpk, pk.size() * 4 + 4096); (void)hipMalloc(&dun, un.size()); (void)hipMalloc(&dcA, cpw * waves); (void)hipMalloc(&dcB, cpw * waves);
    (void)hipMalloc(&doA, njobs * 4 * 4 + 4096); (void)hipMalloc(&doB, njobs * 4 * 4 + 4096); (void)hipMalloc(&dj, njobs * sizeof(Job));
    (void)hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dun, un.data(), un.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dj, jobs.data(), njobs * sizeof(Job), hipMemcpyHostToDevice);
    (void)hipMemset(dcA, 0xEE, cpw * waves); (void)hipMemset(dcB, 0xEE, cpw * waves);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double rows_total = 0; for (auto &b : jobs) rows_total += b.rows; rows_total *= reps;
    const double per = (double)pr.multiProcessorCount * 4;
    float ms_of[2] = { 0, 0 };
    for (int which = 0; which < 2; which++) {
        float bestms = 1e9f;
        for (int it = 0; it < 4; it++) {
            (void)hipEventRecord(e0, 0);
            launch_c(cq, which, waves, dpk, dun, dj, reps, which == 0 ? dcA : dcB, cpw, which == 0 ? doA : doB);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < bestms) bestms = ms;
        }
        ms_of[which] = bestms;
        printf("%-34s U %d(-%d) C %d rows %d(-%d) %s %d waves/SIMD: %.3f ms, %.2f ns per DP row and SIMD\n", which == 0 ? "four per wavefront (1p_g16, x 2)" : "eight per wavefront (1p_h8)", U, spreadU, cq, rows, spreadR,
               mixed ? "mixed scores" : "(5,1,1)", WPS, bestms, bestms * 1e6 / (rows_total / per));
    }
    printf("eight / four = %.3f\n", ms_of[1] / ms_of[0]);
    if (hipDeviceSynchronize() != hipSuccess) { printf("a kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    std::vector<uint8_t> cA(cpw * waves), cB(cpw * waves); std::vector<int> oA(njobs * 4), oB(njobs * 4);
    (void)hipMemcpy(cA.data(), dcA, cA.size(), hipMemcpyDeviceToHost); (void)hipMemcpy(cB.data(), dcB, cB.size(), hipMemcpyDeviceToHost);
    (void)hipMemcpy(oA.data(), doA, oA.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(oB.data(), doB, oB.size() * 4, hipMemcpyDeviceToHost);
    long bad_cells = 0, bad_best = 0, checked = 0;
    for (int j = 0; j < njobs; j++) {
        const Job &b = jobs[j];
        const int w = j >> 3, slot = j & 7;
        int wmax = 0; for (int g = 0; g < 8; g++) wmax = jobs[w * 8 + g].rows > wmax ? jobs[w * 8 + g].rows : wmax;
        for (int k = 0; k < 3; k++) if (oA[j * 4 + k] != oB[j * 4 + k]) { if (bad_best < 5) printf("best differs job %d field %d: %d vs %d (U %d rows %d)\n", j, k, oA[j * 4 + k], oB[j * 4 + k], b.U, b.rows); bad_best++; }
        if (j % 37 != 0 && j > 64) continue;            // cells of a sample of the jobs
        const size_t m0 = (size_t)w * cpw + (size_t)slot * DPQP_DP_BYTES(cq, wmax);
        for (int i = 1; i <= b.rows; i++)
            for (int c = 0; c < b.U; c++) {
                const size_t at = m0 + (size_t)((i - 1) >> 1) * 16 * cq + c;
                const int a = (cA[at] >> (4 * ((i - 1) & 1))) & 15, q = (cB[at] >> (4 * ((i - 1) & 1))) & 15;
                checked++;
                if (canon(a) != canon(q)) { if (bad_cells < 10) printf("cell differs job %d (slot %d U %d rows %d G %d) row %d col %d: %x vs %x\n", j, slot, b.U, b.rows, b.G, i, c + 1, a, q); bad_cells++; }
            }
    }
    printf("checked %ld cells: %ld differ; best cells differing: %ld\n", checked, bad_cells, bad_best);
    return bad_cells || bad_best ? 1 : 0;
}
