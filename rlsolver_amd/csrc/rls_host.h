// What the host-only translation unit (rls_host.cpp) and the HIP sources share: the public ABI and the error recorder.
#pragma once
#include <stdint.h>
#include "rlsolver_hip.h"

namespace rls {
int fail(int code, const char* fmt, ...);  // records the message (thread local), returns code
}
