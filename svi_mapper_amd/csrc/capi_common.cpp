// capi_common.cpp — status strings, version, device probing of the C ABI (include/svi_hot.h).
#include "common.h"

extern "C" {

const char* svi_status_string(int status)
{
    switch (status) {
    case SVI_OK: return "ok";
    case SVI_ERR_INVALID: return "invalid argument";
    case SVI_ERR_NO_DEVICE: return "no gfx950 HIP device";
    case SVI_ERR_HIP: return "HIP runtime error";
    case SVI_ERR_STATE: return "call out of order";
    case SVI_ERR_UNSUPPORTED: return "unsupported graph shape";
    case SVI_ERR_NOT_FOUND: return "vertex id not found";
    case SVI_ERR_IO: return "file i/o error";
    case SVI_ERR_COMM: return "all-reduce hook failed";
    case SVI_ERR_INTERNAL: return "internal error";
    default: return "unknown status";
    }
}

const char* svi_last_error(void) { return svi::last_error().c_str(); }

int svi_version(void) { return SVI_HOT_VERSION; }

int svi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} // extern "C"
