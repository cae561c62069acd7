// Shared helpers for the gfx950 kernels.  Internal; the public ABI is include/rlsolver_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rlsolver_hip.h"
#include "rls_host.h"

namespace rls {

constexpr int kWave = 64;                 // CDNA wavefront
constexpr int kLdsBytes = 160 * 1024;     // per-CU LDS on MI355X
constexpr int kMaxDynLds = 64 * 1024 * 2; // what we are willing to ask for per workgroup

int check_launch(const char* kernel_name); // hipGetLastError -> RLS_ELAUNCH

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- counter-based RNG (Philox-4x32-10), keyed by (seed, global env id) so results
// do not depend on how envs are sharded over ranks/workgroups.
struct Philox {
    uint32_t key0, key1;
    __host__ __device__ Philox(uint64_t seed) : key0((uint32_t)seed), key1((uint32_t)(seed >> 32)) {}
    __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
        uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32);
        lo = (uint32_t)p;
    }
    __host__ __device__ inline void operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t out[4]) const {
        uint32_t k0 = key0, k1 = key1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t h0, l0, h1, l1;
            mulhilo(0xD2511F53u, c0, h0, l0);
            mulhilo(0xCD9E8D57u, c2, h1, l1);
            uint32_t n0 = h1 ^ c1 ^ k0, n1 = l1, n2 = h0 ^ c3 ^ k1, n3 = l0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

// uniform float in [0,1) with 24 random bits (same granularity as torch.rand for f32)
__host__ __device__ inline float u32_to_unit_float(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

}  // namespace rls

#define RLS_REQUIRE(cond, code, ...) \
    do { if (!(cond)) return ::rls::fail((code), __VA_ARGS__); } while (0)

namespace rls {

inline bool rows_vec_aligned(const void* p, int64_t N, int elt) {
    return (((uintptr_t)p) & 15) == 0 && ((N * elt) & 15) == 0;
}

// bit-tile kernels (rls_tile.h): base 16-byte aligned, every row 4-byte aligned
inline bool tile_rows_aligned(const void* p, int64_t N, int elt) {
    return (((uintptr_t)p) & 15) == 0 && ((N * elt) & 3) == 0 && (elt == 1 || ((N * elt) & 15) == 0);
}

// compute units of the current device (256 on MI355X); cached per process
inline int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// Dynamic LDS beyond 64 KB needs the function attribute raised -- ONCE per (kernel instantiation, device), not per launch
// (a host call on the hot path: every G70-sized step).  A lock-free table of the pairs already raised to the whole LDS.
inline void ensure_dyn_lds(const void* kern, size_t lds) {
    if (lds <= 64 * 1024) return;
    static std::atomic<uintptr_t> seen[1024];
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uintptr_t key = (uintptr_t)kern ^ ((uintptr_t)(dev + 1) << 56);
    size_t h = (size_t)((key >> 4) * 0x9E3779B97F4A7C15ull >> 54);
    for (int probe = 0; probe < 1024; ++probe, h = (h + 1) & 1023) {
        const uintptr_t cur = seen[h].load(std::memory_order_acquire);
        if (cur == key) return;
        if (cur == 0) {
            // recorded only when the attribute really was raised: a failed first attempt is retried by the next launch (whose
            // own launch error then names the kernel), never remembered as done
            if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes) != hipSuccess) {
                (void)hipGetLastError();
                return;
            }
            uintptr_t expect = 0;
            seen[h].compare_exchange_strong(expect, key, std::memory_order_release);
            return;
        }
    }
    (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);   // table full: as before
}

inline int grid_for(int64_t total, int block) {
    int64_t g = ceil_div(total, block);
    const int64_t cap = 256 * 8 * 4;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

inline int check_graph(const rls_graph* g) {
    RLS_REQUIRE(g != nullptr, RLS_EINVAL, "graph is NULL");
    RLS_REQUIRE(g->num_nodes > 0 && g->num_nodes < (1ll << 31), RLS_EINVAL, "bad num_nodes %lld",
                (long long)g->num_nodes);
    RLS_REQUIRE(g->num_stored_edges >= 0 && g->nnz >= 0, RLS_EINVAL, "negative edge count");
    RLS_REQUIRE(g->num_stored_edges == 0 || (g->eu && g->ev), RLS_EINVAL, "edge list pointers are NULL");
    RLS_REQUIRE(g->rowptr != nullptr && g->erowptr != nullptr, RLS_EINVAL, "rowptr/erowptr is NULL");
    RLS_REQUIRE(g->nnz == 0 || g->col, RLS_EINVAL, "col is NULL");
    return RLS_OK;
}

}  // namespace rls
