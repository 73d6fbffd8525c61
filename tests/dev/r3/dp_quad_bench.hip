// Development aid (round 3): dp_forward2p_g16 (four alignments per wavefront) against dp_forward2p / dp_forward2p_2c (one per
// wavefront): identical cells and best cells, and the time per alignment row.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../../mtr_amd/csrc -I../../../include [-DWPS=8] [-DCQ=8] -o dp_quad_bench dp_quad_bench.hip
//   ./dp_quad_bench U rows reps [spreadU] [spreadRows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dp_wrap.hip.inc"
#include "dp_quad.hip.inc"
#ifndef WPS
#define WPS 8
#endif
#ifndef CQ
#define CQ 8
#endif
struct Job { int read, base, rows, U; };
// one alignment per wavefront: wavefront w does jobs 4w .. 4w+3 one after the other (the same work as a quad wavefront)
__global__ __launch_bounds__(64, WPS) void k_single(const uint32_t *pk, const uint8_t *units, const Job *jobs, int reps, uint8_t *cells, size_t cells_per_job, int *out)
{
    int acc = 0;
    for (int r = 0; r < reps; r++) {
        for (int g = 0; g < 4; g++) {
            const int ji = blockIdx.x * 4 + g;
            const Job jb = jobs[ji];
            const int rd = uni(jb.read), base = uni(jb.base), rows = uni(jb.rows), U = uni(jb.U);
            int best[2][3];
            uint8_t *codes = cells + (size_t)ji * cells_per_job;
            const uint32_t *p = pk + (size_t)rd * 256; const uint8_t *u = units + (size_t)rd * 512;
            if (U <= 64) dp_forward2p<1>(p, base, rows, u, U, 1, 1, 3, 1, 3, 1, codes, best);
            else dp_forward2p<2>(p, base, rows, u, U, 1, 1, 3, 1, 3, 1, codes, best);
            if (lane_id() == 0 && r == 0) for (int d = 0; d < 2; d++) for (int k = 0; k < 3; k++) out[ji * 8 + d * 3 + k] = best[d][k];
            acc += best[0][0];
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane_id() == 0) out[blockIdx.x * 32 + 7] = acc;
}
__global__ __launch_bounds__(64, WPS) void k_single2c(const uint32_t *pk, const uint8_t *units, const Job *jobs, int reps, uint8_t *cells, size_t cells_per_job, int *out)
{
    int acc = 0;
    for (int r = 0; r < reps; r++) {
        for (int g = 0; g < 4; g++) {
            const int ji = blockIdx.x * 4 + g;
            const Job jb = jobs[ji];
            const int rd = uni(jb.read), base = uni(jb.base), rows = uni(jb.rows), U = uni(jb.U);
            int best[2][3];
            uint8_t *codes = cells + (size_t)ji * cells_per_job;
            const uint32_t *p = pk + (size_t)rd * 256; const uint8_t *u = units + (size_t)rd * 512;
            if (U <= 64) dp_forward2p<1>(p, base, rows, u, U, 1, 1, 3, 1, 3, 1, codes, best);
            else dp_forward2p_2c(p, base, rows, u, U, 1, 1, 3, 1, 3, 1, codes, (U + 1) & ~1, best);
            acc += best[0][0];
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane_id() == 0) out[blockIdx.x * 32 + 7] = acc;
}
__global__ __launch_bounds__(64, WPS) void k_quad(const uint32_t *pk, const uint8_t *units, const Job *jobs, int reps, uint8_t *cells, size_t cells_per_job, int *out, int cq)
{
    int acc = 0;
    for (int r = 0; r < reps; r++) {
        DpQuad q; q.n = 4;
        int maxrows = 0;
        for (int g = 0; g < 4; g++) {
            const Job jb = jobs[blockIdx.x * 4 + g];
            const int rd = uni(jb.read);
            q.pk[g] = pk + (size_t)rd * 256; q.wlim[g] = 256; q.base[g] = uni(jb.base); q.rows[g] = uni(jb.rows); q.U[g] = uni(jb.U);
            q.unit[g] = units + (size_t)rd * 512;
            maxrows = q.rows[g] > maxrows ? q.rows[g] : maxrows;
        }
        int best[4][2][3];
        uint8_t *cb = cells + (size_t)blockIdx.x * 4 * cells_per_job;
#ifdef ONLYC
        dp_forward2p_g16<ONLYC>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best);
#else
        switch (cq) {
        case 2: dp_forward2p_g16<2>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best); break;
        case 3: dp_forward2p_g16<3>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best); break;
        case 4: dp_forward2p_g16<4>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best); break;
        case 5: dp_forward2p_g16<5>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best); break;
        case 6: dp_forward2p_g16<6>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best); break;
        case 7: dp_forward2p_g16<7>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best); break;
        default: dp_forward2p_g16<8>(q, 1, 1, 3, 1, 3, 1, cb, maxrows, best); break;
        }
#endif
        if (lane_id() == 0 && r == 0) for (int g = 0; g < 4; g++) for (int d = 0; d < 2; d++) for (int k = 0; k < 3; k++) out[(blockIdx.x * 4 + g) * 8 + d * 3 + k] = best[g][d][k];
        acc += best[0][0][0];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane_id() == 0) out[blockIdx.x * 32 + 7] = acc;
}

#ifndef SG
#define SG 5
#define SMM 1
#define SD 1
#endif
__global__ __launch_bounds__(64, WPS) void k_single1p(const uint32_t *pk, const uint8_t *units, const Job *jobs, int reps, uint8_t *cells, size_t cells_per_job, int *out, int fast)
{
    int acc = 0;
    for (int r = 0; r < reps; r++) {
        for (int g = 0; g < 4; g++) {
            const int ji = blockIdx.x * 4 + g;
            const Job jb = jobs[ji];
            const int rd = uni(jb.read), base = uni(jb.base), rows = uni(jb.rows), U = uni(jb.U);
            int bv, bi, bj;
            uint8_t *codes = cells + (size_t)ji * cells_per_job;
            const uint32_t *p = pk + (size_t)rd * 256; const uint8_t *u = units + (size_t)rd * 512;
            if (fast && U > 64) dp_forward_2c(p, base, rows, u, U, SG, SMM, SD, codes, (U + 1) >> 1, bv, bi, bj);
            else if (U <= 64) dp_forward<1>(p, base, rows, u, U, SG, SMM, SD, codes, bv, bi, bj);
            else dp_forward<2>(p, base, rows, u, U, SG, SMM, SD, codes, bv, bi, bj);
            if (lane_id() == 0 && r == 0) { out[ji * 8 + 0] = bv; out[ji * 8 + 1] = bi; out[ji * 8 + 2] = bj; out[ji * 8 + 3] = 0; out[ji * 8 + 4] = 0; out[ji * 8 + 5] = 0; }
            acc += bv;
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (lane_id() == 0) out[blockIdx.x * 32 + 7] = acc;
}
__global__ __launch_bounds__(64, WPS) void k_quad1p(const uint32_t *pk, const uint8_t *units, const Job *jobs, int reps, uint8_t *cells, size_t cells_per_job, int *out, int cq)
{
    int acc = 0;
    for (int r = 0; r < reps; r++) {
        DpQuad q; q.n = 4;
        int maxrows = 0;
        for (int g = 0; g < 4; g++) {
            const Job jb = jobs[blockIdx.x * 4 + g];
            const int rd = uni(jb.read);
            q.pk[g] = pk + (size_t)rd * 256; q.wlim[g] = 256; q.base[g] = uni(jb.base); q.rows[g] = uni(jb.rows); q.U[g] = uni(jb.U);
            q.unit[g] = units + (size_t)rd * 512;
            maxrows = q.rows[g] > maxrows ? q.rows[g] : maxrows;
        }
        for (int g = 0; g < 4; g++) { q.G[g] = SG; q.MM[g] = SMM; q.D[g] = SD; }
        int best[4][3];
        uint8_t *cb = cells + (size_t)blockIdx.x * 4 * cells_per_job;
#ifdef ONLYC
        dp_forward1p_g16<ONLYC>(q, cb, maxrows, best);
#else
        switch (cq) {
        case 1: dp_forward1p_g16<1>(q, cb, maxrows, best); break;
        case 2: dp_forward1p_g16<2>(q, cb, maxrows, best); break;
        case 3: dp_forward1p_g16<3>(q, cb, maxrows, best); break;
        case 4: dp_forward1p_g16<4>(q, cb, maxrows, best); break;
        case 5: dp_forward1p_g16<5>(q, cb, maxrows, best); break;
        case 6: dp_forward1p_g16<6>(q, cb, maxrows, best); break;
        case 7: dp_forward1p_g16<7>(q, cb, maxrows, best); break;
        default: dp_forward1p_g16<8>(q, cb, maxrows, best); break;
        }
#endif
        if (lane_id() == 0 && r == 0) for (int g = 0; g < 4; g++) { for (int k = 0; k < 3; k++) out[(blockIdx.x * 4 + g) * 8 + k] = best[g][k]; for (int k = 3; k < 6; k++) out[(blockIdx.x * 4 + g) * 8 + k] = 0; }
        acc += best[0][0];
        __builtin_amdgcn_wave_barrier();
    }
    if (lane_id() == 0) out[blockIdx.x * 32 + 7] = acc;
}
static int canon(int f) { return (f & 8) ? 8 : !(f & 1) ? 0 : !(f & 2) ? 1 : !(f & 4) ? 3 : 7; }
int main(int argc, char **argv)
{
    const int U = argc > 1 ? atoi(argv[1]) : 100, rows = argc > 2 ? atoi(argv[2]) : 1000, reps = argc > 3 ? atoi(argv[3]) : 4;
    const int spreadU = argc > 4 ? atoi(argv[4]) : 0, spreadR = argc > 5 ? atoi(argv[5]) : 0;
    const int cq = argc > 6 && atoi(argv[6]) > 0 ? atoi(argv[6]) : (U + 15) / 16;
    const int single = argc > 7 ? atoi(argv[7]) : 0;
    hipDeviceProp_t pr; (void)hipGetDeviceProperties(&pr, 0);
    const int waves = pr.multiProcessorCount * 4 * WPS, njobs = waves * 4;
    std::vector<uint32_t> pk(1024 * 256); std::vector<uint8_t> un(1024 * 512);
    std::vector<int> runit(1024);
    srand(7);
    for (int r = 0; r < 1024; r++) {
        int Ur = U - (spreadU ? rand() % (spreadU + 1) : 0); if (Ur < 2) Ur = 2;
        runit[r] = Ur;
        std::vector<int> unit(Ur); for (int j = 0; j < Ur; j++) unit[j] = rand() & 3;
        for (int j = 0; j < Ur; j++) un[(size_t)r * 512 + j] = (uint8_t)unit[j];
        for (int w = 0; w < 256; w++) { uint32_t v = 0; for (int b = 0; b < 16; b++) { int pos = w * 16 + b; int c = unit[pos % Ur]; if (rand() % 10 == 0) c = rand() & 3; v |= (uint32_t)c << (30 - 2 * b); } pk[(size_t)r * 256 + w] = v; }
    }
    std::vector<Job> jobs(njobs);
    for (int j = 0; j < njobs; j++) {
        Job &b = jobs[j]; b.read = rand() & 1023; b.base = 90 + rand() % 40; b.rows = rows - (spreadR ? rand() % (spreadR + 1) : 0); if (b.rows < 1) b.rows = 1; b.U = runit[b.read];
        if (b.base + b.rows + 20 > 4096) b.rows = 4096 - 20 - b.base;
    }
    uint32_t *dpk; uint8_t *dun, *dcA, *dcB; int *doA, *doB; Job *dj;
    const size_t cpj = (size_t)(rows + 8) * 128 + 4096;
    (void)hipMalloc(&dpk, pk.size() * 4 + 4096); (void)hipMalloc(&dun, un.size()); (void)hipMalloc(&dcA, cpj * njobs); (void)hipMalloc(&dcB, cpj * njobs);
    (void)hipMalloc(&doA, njobs * 8 * 4 + 4096); (void)hipMalloc(&doB, njobs * 8 * 4 + 4096); (void)hipMalloc(&dj, njobs * sizeof(Job));
    (void)hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dun, un.data(), un.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(dj, jobs.data(), njobs * sizeof(Job), hipMemcpyHostToDevice);
    (void)hipMemset(dcA, 0xEE, cpj * njobs); (void)hipMemset(dcB, 0xEE, cpj * njobs);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double rows_total = 0; for (auto &b : jobs) rows_total += b.rows; rows_total *= reps;
    const double per = (double)pr.multiProcessorCount * 4;
    for (int which = 0; which < 3; which++) {
        float bestms = 1e9f;
        for (int it = 0; it < 3; it++) {
            (void)hipEventRecord(e0, 0);
            if (single) {
                if (which == 0) hipLaunchKernelGGL(k_single1p, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcA, cpj, doA, 0);
                else if (which == 1) hipLaunchKernelGGL(k_single1p, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcA, cpj, doB, 1);
                else hipLaunchKernelGGL(k_quad1p, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcB, cpj, doB, cq);
            }
            else if (which == 0) hipLaunchKernelGGL(k_single, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcA, cpj, doA);
            else if (which == 1) hipLaunchKernelGGL(k_single2c, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcA + 0, cpj, doB);
            else hipLaunchKernelGGL(k_quad, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcB, cpj, doB, cq);
            (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < bestms) bestms = ms;
            if (which == 1) break;      // (its cells overwrite the reference's in another stride: run once for the time only after the check below)
        }
        if (which == 1) {               // time it properly on a scratch copy: reuse dcB before the quad run
            bestms = 1e9f;
            for (int it = 0; it < 3; it++) {
                (void)hipEventRecord(e0, 0);
                if (single) hipLaunchKernelGGL(k_single1p, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcB, cpj, doB, 1);
                else hipLaunchKernelGGL(k_single2c, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, reps, dcB, cpj, doB);
                (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (it > 0 && ms < bestms) bestms = ms;
            }
            // restore the reference cells
            if (single) hipLaunchKernelGGL(k_single1p, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, 1, dcA, cpj, doA, 0);
            else hipLaunchKernelGGL(k_single, dim3(waves), dim3(64), 0, 0, dpk, dun, dj, 1, dcA, cpj, doA);
            (void)hipDeviceSynchronize();
        }
        const char *nm = which == 0 ? "one per wavefront (dp_forward2p)" : which == 1 ? "one per wavefront (2 columns/lane)" : "four per wavefront (g16)";
        printf("%-36s U %d(-%d) rows %d(-%d) %d waves/SIMD: %.3f ms, %.2f ns per alignment row and SIMD\n", nm, U, spreadU, rows, spreadR, WPS, bestms, bestms * 1e6 / (rows_total / per));
    }
    // ---- compare: cells of every job (reference: linear, row stride U; quad: row i of job g of wavefront w at ((i-1)*4+g)*16*CQ) and best cells
    std::vector<uint8_t> cA(cpj * njobs), cB(cpj * njobs); std::vector<int> oA(njobs * 8), oB(njobs * 8);
    (void)hipMemcpy(cA.data(), dcA, cA.size(), hipMemcpyDeviceToHost); (void)hipMemcpy(cB.data(), dcB, cB.size(), hipMemcpyDeviceToHost);
    (void)hipMemcpy(oA.data(), doA, oA.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(oB.data(), doB, oB.size() * 4, hipMemcpyDeviceToHost);
    std::vector<int> wmax(waves, 0);
    for (int j = 0; j < njobs; j++) wmax[j >> 2] = jobs[j].rows > wmax[j >> 2] ? jobs[j].rows : wmax[j >> 2];
    long bad_cells = 0, bad_best = 0, checked = 0;
    for (int j = 0; j < njobs; j++) {
        const Job &b = jobs[j];
        const int w = j >> 2, g = j & 3;
        for (int d = 0; d < 2; d++) for (int k = 0; k < 3; k++) if (oA[j * 8 + d * 3 + k] != oB[j * 8 + d * 3 + k]) { if (bad_best < 5) printf("best differs job %d set %d field %d: %d vs %d (U %d rows %d)\n", j, d, k, oA[j * 8 + d * 3 + k], oB[j * 8 + d * 3 + k], b.U, b.rows); bad_best++; }
        if (j % 37 != 0 && j > 64) continue;            // cells of a sample of the jobs
        for (int i = 1; i <= b.rows; i++)
            for (int c = 0; c < b.U; c++) {
                const uint8_t a = cA[(size_t)j * cpj + (size_t)(i - 1) * b.U + c];
                const uint8_t q = cB[(size_t)w * 4 * cpj + (size_t)g * DPQ_DP_BYTES(cq, wmax[w]) + (size_t)(i - 1) * 16 * cq + c];
                checked++;
                if (single ? (a != canon(q & 15)) : (a != q)) { if (bad_cells < 10) printf("cell differs job %d (U %d rows %d) row %d col %d: %02x vs %02x\n", j, b.U, b.rows, i, c + 1, a, q); bad_cells++; }
            }
    }
    printf("checked %ld cells: %ld differ; best cells differing: %ld\n", checked, bad_cells, bad_best);
    return bad_cells || bad_best ? 1 : 0;
}
