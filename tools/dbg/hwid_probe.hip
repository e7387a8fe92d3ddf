// hwid_probe.hip — where do the workgroups of a launch land?  (xcc, se, sh, cu) of every workgroup from HW_REG_XCC_ID / HW_REG_HW_ID:
// how many distinct CUs the register fields tell apart, and how a launch of fewer workgroups than 2 per CU is spread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned* out, unsigned long long hold_ticks)
{
    extern __shared__ double sm[];
    sm[threadIdx.x] = 1.0;
    if (threadIdx.x == 0) {
        unsigned hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc;
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(8);   // stay resident so that the launch is one round
}
int main(int argc, char** argv)
{
    const int lds = 59 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned* d; (void)hipMalloc(&d, 8 * 4096);
    for (int n : {512, 448, 384, 256}) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), lds, 0, d, 2000ull);
        std::vector<unsigned> h(2 * n);
        (void)hipMemcpy(h.data(), d, 8 * n, hipMemcpyDeviceToHost);
        std::map<unsigned, int> per_cu;
        std::map<unsigned, int> by_blockmod;
        for (int b = 0; b < n; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
            const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
            if (b < 16) printf("  block %d: xcc %u se %u sh %u cu %u simd %u (hw %08x)\n", b, xcc, se, sh, cu, (hw >> 4) & 3, hw);
        }
        int hist[8] = {0};
        for (auto& kv : per_cu) hist[kv.second < 7 ? kv.second : 7]++;
        printf("grid %d: %zu distinct (xcc,se,sh,cu); CUs with 1 WG: %d, 2 WGs: %d, 3+: %d\n", n, per_cu.size(), hist[1], hist[2], hist[3] + hist[4] + hist[5] + hist[6] + hist[7]);
        std::map<unsigned, int> cu_ids, se_ids;
        for (auto& kv : per_cu) { cu_ids[kv.first & 0xF]++; se_ids[(kv.first >> 8) & 0xF]++; }
        printf("   cu field values:"); for (auto& kv : cu_ids) printf(" %u(x%d)", kv.first, kv.second); printf("\n   se field values:"); for (auto& kv : se_ids) printf(" %u(x%d)", kv.first, kv.second); printf("\n");
    }
    return 0;
}
