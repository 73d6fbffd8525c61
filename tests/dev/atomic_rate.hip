// Development aid: how many returning atomicAdd on ONE address the chip completes per second when every wavefront of a
// persistent grid pulls its work from one counter (the staged mode's queues), against counters spread over addresses.
// hipcc --offload-arch=gfx950 -O2 -o atomic_rate atomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_queue(unsigned *counter, unsigned n_items, int chunk, int spread, unsigned long long *sink)
{
    unsigned long long acc = 0;
    unsigned *c = counter + (spread ? (blockIdx.x % spread) * 64 : 0);          // spread: that many counters, 256 B apart
    const unsigned lim = spread ? n_items / spread : n_items;
    for (;;) {
        unsigned t = 0;
        if (threadIdx.x == 0) t = atomicAdd(c, (unsigned)chunk);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= lim) break;
        acc += t;
    }
    if (threadIdx.x == 0 && acc == 0xdeadbeefull) *sink = acc;
}
int main()
{
    unsigned *d; unsigned long long *s;
    (void)hipMalloc(&d, 64 * 4 * 64); (void)hipMalloc(&s, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned n = 2000000;
    for (int waves : { 256, 1024, 4096, 8192 })
        for (int spread : { 0, 8, 64 })
            for (int chunk : { 1, 16 }) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; rep++) {
                    (void)hipMemset(d, 0, 64 * 4 * 64);
                    (void)hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(k_queue, dim3(waves), dim3(64), 0, 0, d, n, chunk, spread, s);
                    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
                    float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                const double atomics = (double)n / chunk + waves;
                printf("waves %5d  counters %2d  chunk %2d: %8.3f ms for %.0f atomics = %6.1f ns each (%.0f M/s)\n", waves, spread ? spread : 1, chunk, best, atomics, best * 1e6 / atomics, atomics / best / 1e3);
            }
    return 0;
}
