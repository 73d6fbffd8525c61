// Development aid: what a process pays before its first kernel has run on this box (runtime initialisation, first
// allocation, code-object load), and what large allocations cost.  hipcc --offload-arch=gfx950 -O2 -o /tmp/probe hip_init_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nothing(int *p) { if (p && threadIdx.x == 0) *p = 1; }
int main()
{
    const double t0 = now();
    double t = t0;
    auto lap = [&](const char *w) { const double n = now(); printf("%-40s %8.1f ms   (+%.1f)\n", w, (n - t0) * 1e3, (n - t) * 1e3); t = n; };
    (void)hipInit(0); lap("hipInit");
    (void)hipSetDevice(0); lap("hipSetDevice");
    int *d = nullptr; (void)hipMalloc(&d, 4); lap("first hipMalloc (4 B)");
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking); lap("hipStreamCreate");
    hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s, d); (void)hipStreamSynchronize(s); lap("first kernel (code-object load)");
    hipLaunchKernelGGL(k_nothing, dim3(1), dim3(64), 0, s, d); (void)hipStreamSynchronize(s); lap("second kernel");
    for (size_t gb : { (size_t)1, (size_t)4, (size_t)16 }) {
        void *p = nullptr; char w[64];
        (void)hipMalloc(&p, gb << 30); snprintf(w, sizeof w, "hipMalloc %zu GiB", gb); lap(w);
        (void)hipMemsetAsync(p, 0, 1 << 20, s); (void)hipStreamSynchronize(s); lap("  memset of its first MiB");
        (void)hipFree(p); snprintf(w, sizeof w, "hipFree %zu GiB", gb); lap(w);
    }
    void *h = nullptr; (void)hipHostMalloc(&h, 64 << 20, hipHostMallocDefault); lap("hipHostMalloc 64 MiB");
    (void)hipHostFree(h); lap("hipHostFree");
    (void)hipFree(d); (void)hipStreamDestroy(s); lap("free + stream destroy");
    return 0;
}
