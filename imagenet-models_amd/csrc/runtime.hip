// Error plumbing, version and device query of libgaext (include/gaext.h).
#include <stdarg.h>
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void ga_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ga_check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ga_set_error("%s: %s", what, hipGetErrorString(e));
        return GA_ERR_HIP;
    }
    return GA_OK;
}

extern "C" int ga_version(void) { return 100; }

extern "C" int ga_last_error(char* buf, size_t n) {
    if (buf && n) {
        strncpy(buf, g_err, n - 1);
        buf[n - 1] = 0;
    }
    return (int)strlen(g_err);
}

extern "C" int ga_device_info(int* num_cu, int* lds_bytes, int* wave_size) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) {
        ga_set_error("ga_device_info: no HIP device");
        return GA_ERR_HIP;
    }
    if (num_cu) *num_cu = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (wave_size) *wave_size = p.warpSize;
    return GA_OK;
}

extern "C" int ga_memset(void* p, int value, size_t bytes, ga_stream_t stream) {
    if (!p || !bytes) {
        ga_set_error("ga_memset: null/empty");
        return GA_ERR_BAD_ARG;
    }
    const hipError_t e = hipMemsetAsync(p, value, bytes, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        ga_set_error("ga_memset: %s", hipGetErrorString(e));
        return GA_ERR_HIP;
    }
    return GA_OK;
}

