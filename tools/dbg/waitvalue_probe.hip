// waitvalue_probe.hip — how fast can a kernel on stream B start behind a value that a RUNNING kernel on stream A publishes?
//   (1) hipStreamWaitValue64 on signal memory written by the last-arriving workgroup of a producer that keeps running
//   (2) hipStreamWaitEvent behind a producer kernel's end (the plain cross-stream dependency)
// and does the consumer's workgroup (512 threads, ~100 KB LDS) get a CU while the producer occupies the chip?
// build: hipcc -O2 --offload-arch=gfx950 tools/dbg/waitvalue_probe.hip -o gpurun_out/waitvalue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); } // 100 MHz

// producer: every workgroup "works" for phase_ticks, then arrives at a counter; the last arriver publishes `*sig = 1` and stamps the time;
// then everybody goes on working for another phase_ticks (the kernel is still running when the consumer should start)
__global__ __launch_bounds__(256) void producer(unsigned long long phase_ticks, int* counter, unsigned long long* sig, unsigned long long* stamps, int lds_words)
{
    extern __shared__ double sm[];
    if (lds_words > 0) sm[threadIdx.x % lds_words] = 1.0;
    const unsigned long long t0 = now();
    while (now() - t0 < phase_ticks) __builtin_amdgcn_s_sleep(4);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const int k = atomicAdd(counter, 1);
        if (k == (int)gridDim.x - 1) {
            stamps[0] = now();
            __hip_atomic_store(sig, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const unsigned long long t1 = now();
    while (now() - t1 < phase_ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = now();
}

__global__ __launch_bounds__(512) void consumer(unsigned long long* stamps, int slot, int lds_words)
{
    extern __shared__ double sm[];
    if (lds_words > 0) sm[threadIdx.x % lds_words] = 1.0;
    if (threadIdx.x == 0) {
        stamps[slot + blockIdx.x] = now();
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        stamps[slot + 32 + blockIdx.x] = xcc & 0xF;
    }
}

int main(int argc, char** argv)
{
    const int n_wg = argc > 1 ? atoi(argv[1]) : 480;         // producer workgroups (two fit a CU by LDS)
    const unsigned long long phase = argc > 2 ? atoll(argv[2]) : 5000; // 50 us per phase
    int dev_ok = 0;
    CK(hipDeviceGetAttribute(&dev_ok, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("CanUseStreamWaitValue %d\n", dev_ok);
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    printf("priority range least %d greatest %d\n", lo, hi);
    CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
    unsigned long long* sig = nullptr;
    CK(hipExtMallocWithFlags(reinterpret_cast<void**>(&sig), 8, hipMallocSignalMemory));
    int* counter; unsigned long long* stamps;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&stamps, 8 * 128));
    const int lds_p = 59 * 1024, lds_c = 96 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(producer), hipFuncAttributeMaxDynamicSharedMemorySize, lds_p));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(consumer), hipFuncAttributeMaxDynamicSharedMemorySize, lds_c));
    std::vector<unsigned long long> h(128);
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4, A)); CK(hipMemsetAsync(stamps, 0, 8 * 128, A));
        *sig = 0;
        CK(hipStreamSynchronize(A));
        // (1) wait-value
        CK(hipStreamWaitValue64(B, sig, 1, hipStreamWaitValueGte, 0xFFFFFFFFFFFFFFFFull));
        hipLaunchKernelGGL(consumer, dim3(8), dim3(512), lds_c, B, stamps, 8, 64);
        hipLaunchKernelGGL(producer, dim3(n_wg), dim3(256), lds_p, A, phase, counter, sig, stamps, 64);
        CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        CK(hipMemcpy(h.data(), stamps, 8 * 128, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull, last = 0;
        for (int i = 0; i < 8; ++i) { first = h[8 + i] < first ? h[8 + i] : first; last = h[8 + i] > last ? h[8 + i] : last; }
        printf("waitvalue rep %d: publish -> first consumer WG %.2f us, -> last %.2f us; producer end - publish %.2f us; xcc", rep,
               ((double)first - (double)h[0]) / 100.0, ((double)last - (double)h[0]) / 100.0, ((double)h[1] - (double)h[0]) / 100.0);
        for (int i = 0; i < 8; ++i) printf(" %llu", h[8 + 32 + i]);
        printf("\n");
    }
    // (2) event
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4, A)); CK(hipMemsetAsync(stamps, 0, 8 * 128, A));
        *sig = 0;
        CK(hipStreamSynchronize(A));
        hipLaunchKernelGGL(producer, dim3(n_wg), dim3(256), lds_p, A, phase, counter, sig, stamps, 64);
        CK(hipEventRecord(ev, A));
        CK(hipStreamWaitEvent(B, ev, 0));
        hipLaunchKernelGGL(consumer, dim3(8), dim3(512), lds_c, B, stamps, 8, 64);
        CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
        CK(hipMemcpy(h.data(), stamps, 8 * 128, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull;
        for (int i = 0; i < 8; ++i) first = h[8 + i] < first ? h[8 + i] : first;
        printf("event rep %d: producer end -> first consumer WG %.2f us\n", rep, ((double)first - (double)h[1]) / 100.0);
    }
    // (3) same stream, for reference
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(counter, 0, 4, A)); CK(hipMemsetAsync(stamps, 0, 8 * 128, A));
        CK(hipStreamSynchronize(A));
        hipLaunchKernelGGL(producer, dim3(n_wg), dim3(256), lds_p, A, phase, counter, sig, stamps, 64);
        hipLaunchKernelGGL(consumer, dim3(8), dim3(512), lds_c, A, stamps, 8, 64);
        CK(hipStreamSynchronize(A));
        CK(hipMemcpy(h.data(), stamps, 8 * 128, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull;
        for (int i = 0; i < 8; ++i) first = h[8 + i] < first ? h[8 + i] : first;
        printf("same stream rep %d: producer end -> first consumer WG %.2f us\n", rep, ((double)first - (double)h[1]) / 100.0);
    }
    return 0;
}
