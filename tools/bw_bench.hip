// Achievable HBM bandwidth on this box for the access shapes of the Jacobian sweep: NP input planes read with one
// coalesced 8 B load per lane, NW planes written, one-shot workgroups (lane = element), buffer sizes like config 4.
//   hipcc -O3 --offload-arch=gfx950 tools/bw_bench.hip -o /tmp/bw_bench && /tmp/bw_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NR, int NW>
__global__ __launch_bounds__(256) void k_planes(const double* __restrict__ in, double* __restrict__ out, size_t E)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= E) return;
    double v[NR > 0 ? NR : 1];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = in[k * E + e];
    double s = 1.0;
#pragma unroll
    for (int k = 0; k < NR; ++k) s += v[k];
    if (NW == 0) { if (s == 12345.678) out[e] = s; }
#pragma unroll
    for (int k = 0; k < NW; ++k) out[k * E + e] = s + k;
}

template <int NR, int NW>
static void run(const char* name, double* in, double* out, size_t E)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int grid = (int)((E + 255) / 256);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_planes<NR, NW>), dim3(grid), dim3(256), 0, 0, in, out, E);
    hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_planes<NR, NW>), dim3(grid), dim3(256), 0, 0, in, out, E);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = 8.0 * E * (NR + NW), us = 1e3 * ms / reps;
    printf("%-28s E=%9zu  %7.1f MB  %7.2f us  %6.2f TB/s\n", name, E, bytes / 1e6, us, bytes / us / 1e6);
}

int main()
{
    const size_t Emax = 16u << 20;
    double *in, *out;
    hipMalloc(&in, Emax * 8 * 12); hipMalloc(&out, Emax * 8 * 12);
    hipMemset(in, 0, Emax * 8 * 12); hipMemset(out, 0, Emax * 8 * 12);
    for (size_t E : {(size_t)800000, (size_t)3200000, (size_t)12800000}) {
        run<7, 0>("read 7 planes", in, out, E);
        run<12, 0>("read 12 planes", in, out, E);
        run<0, 12>("write 12 planes", in, out, E);
        run<7, 12>("read 7 write 12 (K2 shape)", in, out, E);
        run<6, 6>("copy 6 planes", in, out, E);
    }
    return 0;
}
