// hip_host_stub.cpp — a malloc-backed stand-in for the two dozen HIP runtime calls the HOST side of the C ABI makes
// (ba_host.cpp, ba_g2o_io.cpp, capi_common.cpp), so that graph construction, structure analysis (build_structure), the
// .g2o reader / writer and the write-back rules run under g++'s AddressSanitizer + UBSan with NO GPU runtime in the
// process (tools/host_san/README in the Makefile header).  "Device" buffers are plain heap blocks: every host-side
// size computation that feeds hipMalloc / hipMemcpyAsync is checked by the sanitizer.  Test infrastructure only -
// never linked into libsvi_hot.so.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = nullptr; return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = 0; return hipSuccess; }
hipError_t hipExtMallocWithFlags(void** p, size_t n, unsigned) { *p = std::calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipStreamWaitValue64(hipStream_t, void*, uint64_t, unsigned, uint64_t) { return hipSuccess; }
hipError_t hipStreamWriteValue64(hipStream_t, void* p, uint64_t v, unsigned) { *static_cast<uint64_t*>(p) = v; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
}
