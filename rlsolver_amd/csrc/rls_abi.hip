// The HIP-dependent remainder of the ABI plumbing: launch-error check and the device count.  Everything host-only lives
// in rls_host.cpp.
#include "rls_common.h"

namespace rls {

int check_launch(const char* kernel_name) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(RLS_ELAUNCH, "%s: %s", kernel_name, hipGetErrorString(e));
    return RLS_OK;
}

}  // namespace rls

extern "C" {

int rls_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

}  // extern "C"
