// Error plumbing and library-level entry points of the C ABI (include/mgf.h).
#include "mgf_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void mgf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* mgf_last_error(void) { return g_err; }

extern "C" int mgf_version(void) { return 100; }

extern "C" int mgf_device_ok(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        mgf_set_error("no HIP device visible");
        return 0;
    }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mgf_set_error("device %d is %s; this library is built for gfx950 only", dev, prop.gcnArchName);
        return 0;
    }
    return 1;
}
