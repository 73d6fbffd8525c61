// development probe: the library's own mtr_k_dp_test kernel with hand-made arguments
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mtr_amd/csrc/k2_units.hip.inc"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1);} } while (0)
template <typename T> T *up(const std::vector<T> &v) { T *d; CK(hipMalloc(&d, v.size() * sizeof(T) + 16)); CK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
int main(int argc, char **argv)
{
    int ntask = argc > 1 ? atoi(argv[1]) : 1;
    const int L = 120, U = 3, qs = 20, qe = 100, rows = qe - qs + 1;
    std::vector<uint32_t> pk(L / 16 + 4, 0u);
    for (int p = 0; p < L; p++) { unsigned b = (p >= 30 && p < 90) ? (unsigned)((p - 30) % 3) : (unsigned)((p * 7 + 3) % 4); pk[p >> 4] |= b << (30 - 2 * (p & 15)); }
    std::vector<int64_t> woff{0}; std::vector<int32_t> lens{L}, order{0};
    std::vector<int32_t> rd(ntask, 0), vqs(ntask, qs), vqe(ntask, qe), g(ntask, 1), m(ntask, 1), d(ntask, 3), uo(ntask + 1);
    std::vector<uint8_t> units;
    for (int t = 0; t < ntask; t++) { uo[t] = (int)units.size(); units.push_back(0); units.push_back(1); units.push_back(2); }
    uo[ntask] = (int)units.size();
    DpTestArgs a;
    a.b.packed = up(pk); a.b.woff = up(woff); a.b.lens = up(lens); a.b.order = up(order); a.b.n_reads = 1;
    a.n_tasks = ntask; a.read_idx = up(rd); a.qs = up(vqs); a.qe = up(vqe); a.units = up(units); a.unit_off = up(uo);
    a.gain = up(g); a.mism = up(m); a.indel = up(d);
    int32_t *out8; CK(hipMalloc(&out8, ntask * 32)); a.out8 = out8;
    uint8_t *sc; CK(hipMalloc(&sc, 4096)); a.scratch = sc; a.scratch_per_wave = 256; a.cells_cap = rows * U;
    int32_t *st; CK(hipMalloc(&st, 4)); CK(hipMemset(st, 0, 4)); a.status = st;
    unsigned *wc; CK(hipMalloc(&wc, 4)); CK(hipMemset(wc, 0, 4)); a.work_counter = wc;
    unsigned long long *cnt; CK(hipMalloc(&cnt, 128)); CK(hipMemset(cnt, 0, 128)); a.counters = cnt;
    printf("launching\n"); fflush(stdout);
    hipLaunchKernelGGL(mtr_k_dp_test, dim3(1), dim3(64), 0, 0, a);
    CK(hipDeviceSynchronize());
    std::vector<int32_t> o(ntask * 8); CK(hipMemcpy(o.data(), out8, ntask * 32, hipMemcpyDeviceToHost));
    for (int t = 0; t < ntask; t++) printf("task %d: %d %d %d %d %d %d %d %d\n", t, o[t*8], o[t*8+1], o[t*8+2], o[t*8+3], o[t*8+4], o[t*8+5], o[t*8+6], o[t*8+7]);
    return 0;
}
