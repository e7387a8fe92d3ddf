// latency of DEPENDENT FP64 vector operations on one wave (and with a second wave on the same SIMD): v_fma_f64 chain, and the
// pivot reciprocal of the factorisation (v_rcp_f64 + two Newton steps)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void kfma(double* out, int iters, long long* cyc)
{
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9, x = 0.5;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) x = fma(x, a, b);
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}
__global__ void krcp(double* out, int iters, long long* cyc)
{
    double x = 1.5 + threadIdx.x * 1e-9;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x = fast_rcp(x) + 1.0;
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double* o; long long* c; (void)hipMalloc(&o, 8 << 20); (void)hipMalloc(&c, 8);
    const int iters = 2000; long long h = 0;
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL(kfma, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%d waves in the workgroup: dependent v_fma_f64 %.1f cycles", waves, (double)h / (iters * 32));
        hipLaunchKernelGGL(krcp, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); (void)hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf(" | rcp + 2 Newton + add %.1f cycles\n", (double)h / (iters * 8));
    }
    return 0;
}
