// latency of DEPENDENT v_mfma_f64_16x16x4_f64 (the accumulator of one is the C operand of the next) against independent issue,
// and of the accumulator used as the B operand of the next product (the X = L^-1 off-diagonal blocks do that)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4 __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void kdep(double* out, int iters, long long* cyc)
{
    v4 acc[CHAINS];
    for (int i = 0; i < CHAINS; ++i) acc[i] = {0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 12 / CHAINS; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = clock64();
    double s = 0; for (int i = 0; i < CHAINS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void kchainB(double* out, int iters, long long* cyc)
{
    v4 acc = {1e-3, 2e-3, 3e-3, 4e-3};
    double a = threadIdx.x * 1e-3;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        v4 r = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) r = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[q], r, 0, 0, 0);
        acc = r;
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    double* o; long long* c; hipMalloc(&o, 8 << 20); hipMalloc(&c, 8);
    const int iters = 2000; long long h = 0;
    for (int waves = 1; waves <= 8; waves *= 2) {
        hipLaunchKernelGGL(kdep<1>, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%d waves: 1 chain  %.1f cycles per MFMA", waves, (double)h / (iters * 12));
        hipLaunchKernelGGL(kdep<2>, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf(" | 2 chains %.1f", (double)h / (iters * 12));
        hipLaunchKernelGGL(kdep<3>, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf(" | 3 chains %.1f", (double)h / (iters * 12));
        hipLaunchKernelGGL(kchainB, dim3(1), dim3(64 * waves), 0, 0, o, iters, c); hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf(" | D->B chain %.1f\n", (double)h / (iters * 4));
    }
    return 0;
}
