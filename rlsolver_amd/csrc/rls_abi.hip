// Error plumbing + trivial ABI entry points.
#include "rls_common.h"
#include <cstdarg>
#include <cstdio>

namespace rls {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char* kernel_name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RLS_ELAUNCH, "%s: %s", kernel_name, hipGetErrorString(e));
    return RLS_OK;
}

}  // namespace rls

extern "C" {

int rls_version(void) { return RLS_ABI_VERSION; }

const char* rls_last_error_string(void) { return rls::g_err; }

int rls_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
