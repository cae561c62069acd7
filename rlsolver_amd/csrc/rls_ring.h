// LDS ring over a shared int32 stream (CSR col entries, MCPG visit stream): the node-sequential
// kernels consume the stream strictly in order, so it is prefetched with direct global->LDS loads a
// quarter ring at a time, at least half a ring ahead of the consumer; nothing on the per-node path
// waits on global memory.
#pragma once
#include "rls_common.h"

namespace rls {

constexpr int kRing = 4096;            // entries in the ring (16 KB)
constexpr int kRefill = kRing / 4;     // entries requested per refill
constexpr int kRingMaxRun = kRing / 8; // longest run (CSR row + 64-entry lookahead block) a consumer may read at once

__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// request entries [F, F + kRefill) of the stream into the ring (entries past `len` are skipped)
__device__ __forceinline__ void ring_refill(const int32_t* __restrict__ stream, int64_t len, int64_t F,
                                            int32_t* ring, int lane) {
#pragma unroll
    for (int k = 0; k < kRefill / kWave; ++k) {
        const int64_t e0 = F + (int64_t)k * kWave;
        if (e0 + lane < len) glds4(stream + e0 + lane, ring + (e0 & (kRing - 1)));
    }
}

// Keep the ring at least half full ahead of `cursor` (wave-uniform call).  Entries the consumer is
// about to read (cursor .. cursor + kRingMaxRun) always belong to a refill older than the newest one,
// and every older refill has been waited for before the newest is issued.
// A wave that sat out earlier batches (fewer nodes than waves) may find the cursor far ahead of its own F:
// it skips what nobody will read and refills until the window is restored (the wait in front of every
// refill but the first covers the run at `cursor`, which lies in the first one or two of them).
__device__ __forceinline__ void ring_advance(const int32_t* __restrict__ stream, int64_t len, int64_t& F,
                                             int64_t cursor, int32_t* ring, int lane) {
    if (F < len && F - cursor < kRing / 2) {
        if (F < cursor) F = cursor & ~(int64_t)(kRefill - 1);
        do {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ring_refill(stream, len, F, ring, lane);
            F += kRefill;
        } while (F < len && F - cursor < kRing / 2);
        if (F - kRefill <= cursor + kRingMaxRun) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // end of stream
    }
}

__device__ __forceinline__ void ring_prime(const int32_t* __restrict__ stream, int64_t len, int64_t& F,
                                           int32_t* ring, int lane) {
    F = 0;
    while (F < len && F < kRing / 2 + kRefill) {
        ring_refill(stream, len, F, ring, lane);
        F += kRefill;
    }
}

}  // namespace rls
