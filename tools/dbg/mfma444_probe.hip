// Finds the lane layout of v_mfma_f64_4x4x4_4b_f64 empirically: a = 1 in lane la only, b = 1 in lane lb only; which lanes of D are non-zero?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out)
{
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) out[la * 64 + lb] = lane + 1;
        }
}
int main()
{
    int* d; hipMalloc(&d, 4096 * sizeof(int)); hipMemset(d, 0, 4096 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    static int h[4096]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // for every A lane: the B lanes it pairs with and the D lane
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) printf(" (B %2d -> D %2d)", lb, h[la * 64 + lb] - 1);
        printf("\n");
    }
    return 0;
}
