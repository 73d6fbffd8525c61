// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS path's store pattern (MI355X_MICROARCH.md, HBM section:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every wavefront owns one slice and writes it as the DP kernels do: row after row, lane = column, ONE BYTE per lane,
// rows of U bytes back to back (U = 100: byte-per-cell matrices of the 2 kb workload; U = 50: the 4-bit format), with a
// pause between rows so that the 16 waves of a CU interleave as they do in the real kernel.  The reference kernel writes
// the same byte count with 16-byte-per-lane streaming stores.  Build: hipcc --offload-arch=gfx950 -O2 -o pmc_calib pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

template <int U>
__global__ __launch_bounds__(64) void w_byte_rows(uint8_t* base, long slice, int rows, int pause)
{
    uint8_t* p = base + (long)blockIdx.x * slice;
    int lane = threadIdx.x;
    for (int r = 0; r < rows; ++r) {
        uint8_t* row = p + (long)r * U;
        row[lane < U ? lane : 0] = (uint8_t)(r + lane);
        if (U > 64 && lane + 64 < U) row[lane + 64] = (uint8_t)(r - lane);
        for (int s = 0; s < pause; ++s) __builtin_amdgcn_s_sleep(2);
    }
}

__global__ __launch_bounds__(64) void w_wide(uint4* base, long slice16, int iters)
{
    uint4* p = base + (long)blockIdx.x * slice16;
    for (int i = 0; i < iters; ++i) p[(long)i * 64 + threadIdx.x] = make_uint4(i, threadIdx.x, 3, 4);
}

// the read side: 16 cells of one row per lane group as the traceback window loader does (dword loads of a short span)
__global__ __launch_bounds__(64) void r_wide(const uint4* base, long slice16, int iters, uint32_t* sink)
{
    const uint4* p = base + (long)blockIdx.x * slice16;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) { uint4 v = p[(long)i * 64 + threadIdx.x]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    int waves = argc > 1 ? atoi(argv[1]) : 4096;
    long slice = (argc > 2 ? atol(argv[2]) : 2) << 20;       // MiB per wave
    int pause = argc > 3 ? atoi(argv[3]) : 8;
    uint8_t* buf; uint32_t* sink;
    CK(hipMalloc(&buf, (size_t)waves * slice)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, (size_t)waves * slice)); CK(hipDeviceSynchronize());
    int rows100 = (int)(slice / 100), rows50 = (int)(slice / 50);
    hipLaunchKernelGGL(w_byte_rows<100>, dim3(waves), dim3(64), 0, 0, buf, slice, rows100, pause); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(w_byte_rows<50>, dim3(waves), dim3(64), 0, 0, buf, slice, rows50, pause); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(w_byte_rows<100>, dim3(waves), dim3(64), 0, 0, buf, slice, rows100, 0); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(w_wide, dim3(waves), dim3(64), 0, 0, (uint4*)buf, slice / 16, (int)(slice / 1024)); CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(r_wide, dim3(waves), dim3(64), 0, 0, (const uint4*)buf, slice / 16, (int)(slice / 1024), sink); CK(hipDeviceSynchronize());
    printf("{\"waves\": %d, \"bytes_w_byte_rows_100\": %ld, \"bytes_w_byte_rows_50\": %ld, \"bytes_w_wide\": %ld, \"bytes_r_wide\": %ld, \"pause\": %d}\n",
           waves, (long)waves * rows100 * 100, (long)waves * rows50 * 50, (long)waves * (slice / 1024) * 1024, (long)waves * (slice / 1024) * 1024, pause);
    return 0;
}
