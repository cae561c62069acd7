// What the host-only translation unit (rls_host.cpp) and the HIP sources share: the public ABI and the error recorder.
#pragma once
#include <stdint.h>
#include "rlsolver_hip.h"

#include <atomic>

namespace rls {
int fail(int code, const char* fmt, ...);  // records the message (thread local), returns code

// ---- tuning table --------------------------------------------------------------------------------------------------
// Launch policies (which tile form, how many waves, ...) are chosen per launch from the shapes; every choice can be FORCED
// through rls_tuning_set("RLS_<NAME>", value) of the C ABI -- an explicit call of the caller's code, which the forced-form
// parity tests (tests/test_gpu_tile32.py) and the A/B sweeps under tools/ use.  The production library reads NO environment
// variable; a development build (-DRLS_DEV: RLS_EXTRA_CFLAGS=-DRLS_DEV python -m rlsolver_amd.build --force) additionally seeds
// the table from the environment variables of the same names when it is loaded.  Forcing a form never changes a result.
#define RLS_KNOB_LIST(X)                                                                                                  \
    X(STEP_NTS) X(STEP_EPW) X(STEP_WPB) X(STEP_PERSIST) X(STEP_CHASE) X(STEP_ALIGN) X(STEP_NOSTAGE)                                       \
    X(ISCO_SEL_CAP) X(ISCO_WAVES) X(ISCO_GLOBAL_ROWS) X(ISCO_FORCE_WG)                                                                                         \
    X(LS_SD_GLOBAL) X(LS_SLICES) X(LS_WAVES) X(LS_PER_ROUND) X(LS_APPLY32) X(SWEEP_NO_LEVELS) X(SWEEP_WAVES) X(SWEEP_UNBATCHED) \
    X(NODE_STATS_MIN_B) X(NODE_STATS_LANE_ENV) X(NODE_STATS_NO_TILE) X(NS_TILE32) X(NS_WAVES) X(NS_PARK) X(NS_ROWS)                             \
    X(K1_TILE32) X(K1_LDS_KB) X(K5_TILE32) X(K6_TILE32) X(K7_WAVES) X(K7_PAIR) X(TILE_NOSTAGE) X(TILE_LINECUT) X(MCPG_SHIM) X(PLAN_FIXED) X(PLAN_BLOCK) X(PLAN_MERGE) X(METRO_QG) X(NARROW_TILE) X(QUBO_LEVELS)
enum Knob {
#define RLS_X(n) KN_##n,
    RLS_KNOB_LIST(RLS_X)
#undef RLS_X
    KN_COUNT
};
constexpr int64_t kKnobUnset = INT64_MIN;
extern std::atomic<int64_t> g_knobs[KN_COUNT];
inline int64_t knob(Knob k, int64_t dflt) {
    const int64_t v = g_knobs[k].load(std::memory_order_relaxed);
    return v == kKnobUnset ? dflt : v;
}
inline bool knob_on(Knob k) { return knob(k, 0) != 0; }
}
