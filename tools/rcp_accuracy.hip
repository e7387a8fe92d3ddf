// accuracy of v_rcp_f64 / v_rsq_f64 seeds and Newton refinements on gfx950 (decides how many steps fast_rcp needs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double r = __builtin_amdgcn_rcp(v);
    out[i] = r;
    r = fma(r, fma(-v, r, 1.0), r); out[n + i] = r;
    r = fma(r, fma(-v, r, 1.0), r); out[2 * n + i] = r;
    double q = __builtin_amdgcn_rsq(v);
    out[3 * n + i] = q;
    q = fma(q * 0.5, fma(-v * q, q, 1.0), q); out[4 * n + i] = q;
    q = fma(q * 0.5, fma(-v * q, q, 1.0), q); out[5 * n + i] = q;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(6 * (size_t)n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-20, 20);
    for (auto& v : x) v = std::exp(u(g));
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
    const char* names[6] = {"rcp seed", "rcp 1 NR", "rcp 2 NR", "rsq seed", "rsq 1 NR", "rsq 2 NR"};
    for (int k = 0; k < 6; ++k) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double exact = k < 3 ? 1.0L / (long double)x[i] : 1.0L / sqrtl((long double)x[i]);
            const long double e = fabsl(((long double)o[(size_t)k * n + i] - exact) / exact);
            if (e > worst) worst = e;
        }
        printf("%-10s max rel err %.3Le = 2^%.1Lf\n", names[k], worst, log2l(worst));
    }
    return 0;
}
