// aggregate FP64 VALU throughput on the whole chip (hipEvents), for several waves per SIMD and ILP
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, bool MULADD>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b)
{
    double x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = MULADD ? (x[i] * a + ((it & 1) ? b : -b)) : fma(x[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int ILP>
void run(double* d, int wgs_per_cu)
{
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<ILP, false>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<ILP, false>), dim3(grid), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double inst = (double)grid * 4 * iters * ILP; // wave instructions
    printf("ILP %d, %d waves/SIMD: %.1f TFLOP/s, %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", ILP, wgs_per_cu,
           inst * 64 * 2 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (inst / 1024));
}
int main()
{
    double* d; hipMalloc(&d, 8 * 256 * 256 * 16);
    for (int w : {1, 2, 4, 8}) { run<1>(d, w); run<4>(d, w); run<8>(d, w); }
    return 0;
}
