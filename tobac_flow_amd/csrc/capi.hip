// Library-level entry points of include/tobac_flow_hip.h: version, error string, device count.
#include "tf_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void tf_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int tf_version(void) { return 100; }
extern "C" const char *tf_last_error(void) { return g_err; }
extern "C" int tf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { tf_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return 0; }
    return n;
}
