// common.h — status plumbing shared by the C-ABI translation units (host side only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

#include "svi_hot.h"

namespace svi {

// Thread-local last-error text (svi_last_error()).
inline std::string& last_error()
{
    static thread_local std::string e;
    return e;
}

inline int fail(int status, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return status;
}

#define SVI_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return svi::fail(SVI_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// hipSetDevice at the head of an entry point.  The runtime's "last error" is per thread and sticky: a failure left behind by
// someone else in the process (another library probing a device that does not exist, an earlier call of ours that reported its
// error properly) must not be picked up by the hipGetLastError() that checks OUR launches further down - it is dropped here.
inline hipError_t enter_device(int device)
{
    (void)hipGetLastError();
    return hipSetDevice(device);
}

// Compute units of `device` (0 if the query fails).  One attribute instead of the whole property block: hipGetDeviceProperties
// called from several threads at once (the shards of a sharded solve initialise side by side) has been seen to fail with
// "invalid device ordinal" on a fresh box, and an unchecked failure stays behind as the thread's last error.
inline int device_compute_units(int device)
{
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// Select `device`, verifying that one exists. The library has no CPU fallback: no device => error.
inline int use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(SVI_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(SVI_ERR_INVALID, "device %d out of range (have %d)", device, n);
    SVI_HIP(enter_device(device));
    return SVI_OK;
}

// Growable device buffer owned by a handle.
struct DevBuf {
    void*  p   = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes)
    {
        if (bytes <= cap) return SVI_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        SVI_HIP(hipMalloc(&p, want));
        cap = want;
        return SVI_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

} // namespace svi
