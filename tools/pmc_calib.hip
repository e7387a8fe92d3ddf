// Calibration of the FETCH_SIZE / WRITE_SIZE counters for the BA sweeps' access pattern on gfx950:
// coalesced 8-byte-per-lane streaming loads and stores of a KNOWN byte count (MI355X_MICROARCH.md §HBM
// says FETCH_SIZE under-reports wide streaming reads by 2x and that other widths must be calibrated).
// Run under  rocprofv3 --pmc FETCH_SIZE  and  rocprofv3 --pmc WRITE_SIZE  (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void calib_read8(const double* __restrict__ in, double* out, size_t n)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
    if (s == 12345.678) out[0] = s;
}
__global__ void calib_write8(double* out, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (double)i;
}
__global__ void calib_read16(const double2* __restrict__ in, double* out, size_t n)
{
    double s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = in[i]; s += v.x + v.y; }
    if (s == 12345.678) out[0] = s;
}
int main()
{
    const size_t n = (size_t)1 << 27; // 1 GiB of doubles: far beyond the 256 MiB Infinity Cache
    double *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(calib_read8, dim3(4096), dim3(256), 0, 0, a, b, n);
    hipLaunchKernelGGL(calib_write8, dim3(4096), dim3(256), 0, 0, b, n);
    hipLaunchKernelGGL(calib_read16, dim3(4096), dim3(256), 0, 0, (const double2*)a, b, n / 2);
    hipDeviceSynchronize();
    printf("calibration kernels moved %zu bytes each\n", n * 8);
    return 0;
}
