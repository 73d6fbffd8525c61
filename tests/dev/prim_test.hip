// development probe: wave primitives used by dp_wrap.hip.inc, one test per argv[1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mtr_amd/csrc/dp_wrap.hip.inc"

__global__ void k_scan(const unsigned *in, unsigned *out) { out[threadIdx.x] = scan_max_u32(in[threadIdx.x]); }
__global__ void k_shift(const int *in, int *out) { out[threadIdx.x] = shift_up1(in[threadIdx.x], -7); }
__global__ void k_fwd(const uint32_t *pk, int base, int rows, const uint8_t *unit, int U, uint8_t *codes, int *out)
{
    int bv, bi, bj;
    dp_forward<1>(pk, base, rows, unit, U, 1, 1, 3, codes, bv, bi, bj);
    if (threadIdx.x == 0) { out[0] = bv; out[1] = bi; out[2] = bj; }
}
__global__ void k_full(const uint32_t *pk, int base, int rows, const uint8_t *unit, int U, uint8_t *codes, int *out, unsigned long long *cnt)
{
    DpRes o;
    bool ok = dp_wrap(pk, base, rows, unit, U, 1, 1, 3, codes, (size_t)rows * U, 0, nullptr, nullptr, o, cnt);
    if (threadIdx.x == 0) { out[0] = ok; out[1] = o.stop_i; out[2] = o.end_i; out[3] = o.mat; out[4] = o.mis; out[5] = o.ins; out[6] = o.del; }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s\n", hipGetErrorString(e), #x); exit(1);} } while (0)
int main(int argc, char **argv)
{
    int which = argc > 1 ? atoi(argv[1]) : 0;
    if (which == 0) {
        std::vector<unsigned> h(64), r(64); for (int i = 0; i < 64; i++) h[i] = (unsigned)((i * 37) % 23 + (i == 40 ? 100 : 0));
        unsigned *d, *o; CK(hipMalloc(&d, 256)); CK(hipMalloc(&o, 256)); CK(hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice));
        k_scan<<<1, 64>>>(d, o); CK(hipDeviceSynchronize()); CK(hipMemcpy(r.data(), o, 256, hipMemcpyDeviceToHost));
        unsigned m = 0; int bad = 0; for (int i = 0; i < 64; i++) { m = h[i] > m ? h[i] : m; if (r[i] != m) bad++; }
        printf("scan_max_u32: %d mismatches\n", bad);
    } else if (which == 1) {
        std::vector<int> h(64), r(64); for (int i = 0; i < 64; i++) h[i] = i * 3 + 1;
        int *d, *o; CK(hipMalloc(&d, 256)); CK(hipMalloc(&o, 256)); CK(hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice));
        k_shift<<<1, 64>>>(d, o); CK(hipDeviceSynchronize()); CK(hipMemcpy(r.data(), o, 256, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 64; i++) { int want = i == 0 ? -7 : h[i - 1]; if (r[i] != want) bad++; }
        printf("shift_up1: %d mismatches (lane0=%d lane1=%d lane16=%d lane32=%d)\n", bad, r[0], r[1], r[16], r[32]);
    } else {
        const int L = 120, U = 3, qs = 20, qe = 100, rows = qe - qs + 1;
        std::vector<uint32_t> pk(L / 16 + 4, 0u);
        for (int p = 0; p < L; p++) { unsigned b = (p >= 30 && p < 90) ? (unsigned)((p - 30) % 3) : (unsigned)((p * 7 + 3) % 4); pk[p >> 4] |= b << (30 - 2 * (p & 15)); }
        uint8_t hu[3] = { 0, 1, 2 };
        uint32_t *dpk; uint8_t *du, *dc; int *dout; unsigned long long *dcnt;
        CK(hipMalloc(&dpk, pk.size() * 4)); CK(hipMalloc(&du, 16)); CK(hipMalloc(&dc, rows * U + 256)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dcnt, 16 * 8));
        CK(hipMemset(dcnt, 0, 128)); CK(hipMemset(dout, 0, 64));
        CK(hipMemcpy(dpk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(du, hu, 3, hipMemcpyHostToDevice));
        if (which == 2) k_fwd<<<1, 64>>>(dpk, qs, rows, du, U, dc, dout);
        else k_full<<<1, 64>>>(dpk, qs, rows, du, U, dc, dout, dcnt);
        CK(hipDeviceSynchronize());
        int out[8]; CK(hipMemcpy(out, dout, 32, hipMemcpyDeviceToHost));
        printf("test %d: %d %d %d %d %d %d %d\n", which, out[0], out[1], out[2], out[3], out[4], out[5], out[6]);
        if (which == 2) { std::vector<uint8_t> c(rows * U); CK(hipMemcpy(c.data(), dc, rows * U, hipMemcpyDeviceToHost)); printf("codes rows 1..12:"); for (int i = 0; i < 36; i++) printf(" %d", c[i]); printf("\n"); }
    }
    return 0;
}
