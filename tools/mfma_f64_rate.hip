// issue rate of v_mfma_f64_4x4x4_4b_f64 (and 16x16x4 for comparison) on one wave / all CUs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
__global__ void k444(double* out, int iters, long long* cyc)
{
    double acc[36];
    for (int i = 0; i < 36; ++i) acc[i] = 0.0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 36; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0; for (int i = 0; i < 36; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void k16(double* out, int iters, long long* cyc)
{
    v4 acc[9];
    for (int i = 0; i < 9; ++i) acc[i] = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0; for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double* o; long long* c; hipMalloc(&o, 8 << 20); hipMalloc(&c, 8);
    const int iters = 2000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        long long h = 0;
        hipLaunchKernelGGL(k444, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("4x4x4   %d waves in one workgroup: %.1f cycles per instruction per wave\n", waves, (double)h / (iters * 36));
        hipLaunchKernelGGL(k16, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("16x16x4 %d waves in one workgroup: %.1f cycles per instruction per wave\n", waves, (double)h / (iters * 9));
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k444, dim3(1024), dim3(256), 0, 0, o, iters, c);
    hipEventRecord(e0); hipLaunchKernelGGL(k444, dim3(2048), dim3(256), 0, 0, o, iters, c); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("4x4x4 chip: %.1f TFLOP/s\n", 2048.0 * 4 * iters * 36 * 256 * 2 / (ms * 1e-3) / 1e12);
    hipEventRecord(e0); hipLaunchKernelGGL(k16, dim3(2048), dim3(256), 0, 0, o, iters, c); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("16x16x4 chip: %.1f TFLOP/s\n", 2048.0 * 4 * iters * 9 * 1024 * 2 / (ms * 1e-3) / 1e12);
    return 0;
}
