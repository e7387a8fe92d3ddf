// Micro-benchmarks of FP64 VALU / MFMA issue on gfx950 (development aid, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int ILP>
__global__ void k_fma(double* out, int iters, double a, double b)
{
    double x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = fma(x[i], a, b);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}
template <int ILP>
__global__ void k_fma32(float* out, int iters, float a, float b)
{
    float x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x + i;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = fmaf(x[i], a, b);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((double*)out)[1 << 19] = (double)(t1 - t0);
}
template <int ILP>
__global__ void k_mfma(double* out, int iters, double a, double b)
{
    v4f64 acc[ILP];
    for (int i = 0; i < ILP; ++i) acc[i] = {0, 0, 0, 0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0);
}
int main()
{
    double* d;
    hipMalloc(&d, sizeof(double) * ((1 << 20) + 16));
    const int iters = 2000;
    double cyc;
    auto rd = [&]() { hipDeviceSynchronize(); hipMemcpy(&cyc, d + (1 << 20), 8, hipMemcpyDeviceToHost); return cyc; };
    for (int threads : {64, 256, 512, 1024}) {
        hipLaunchKernelGGL(k_fma<1>, dim3(1), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9);
        double c1 = rd();
        hipLaunchKernelGGL(k_fma<8>, dim3(1), dim3(threads), 0, 0, d, iters, 1.0000001, 1e-9);
        double c8 = rd();
        hipLaunchKernelGGL(k_fma32<8>, dim3(1), dim3(threads), 0, 0, (float*)d, iters, 1.0000001f, 1e-9f);
        hipDeviceSynchronize(); double c32; hipMemcpy(&c32, ((double*)d) + (1 << 19), 8, hipMemcpyDeviceToHost);
        hipLaunchKernelGGL(k_mfma<1>, dim3(1), dim3(threads), 0, 0, d, iters, 1.0, 1.0);
        double m1 = rd();
        hipLaunchKernelGGL(k_mfma<4>, dim3(1), dim3(threads), 0, 0, d, iters, 1.0, 1.0);
        double m4 = rd();
        printf("threads/WG %4d (waves/SIMD %.2f): f64 fma dependent %.1f cyc/op | 8 independent %.1f cyc/op/wave | f32 8-indep %.1f | mfma_f64_16x16x4 dependent %.1f cyc | 4 indep %.1f cyc/op/wave\n",
               threads, threads / 256.0, c1 / iters, c8 / (8.0 * iters), c32 / (8.0 * iters), m1 / iters, m4 / (4.0 * iters));
    }
    hipFree(d);
    return 0;
}
