// 64-env "bit tile" primitives shared by the MaxCut kernels.
//
// Layout idea: the API boundary is env-major bytes
// x[B, N] (torch.bool).  On chip a wavefront owns 64 consecutive envs and keeps
// their spins transposed and bit-packed in LDS:  words[n] is a 64-bit word whose
// bit e is the spin of node n in env (b0 + e).  One XOR of two words compares an
// edge in 64 envs at once; a broadcast LDS read + v_bfe gives a lane its own env's
// spin.  N * 8 bytes of LDS per wave (16 KB for G22, 80 KB for N = 10^4).
#pragma once
#include "rls_common.h"

namespace rls {

__device__ __forceinline__ bool spin_is_set(uint8_t v) { return v != 0; }
__device__ __forceinline__ bool spin_is_set(float v) { return v > 0.0f; }  // env_PPO: xs > 0

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// widest aligned vector of spins + "logical_not of element idx" on it
template <typename T> struct SpinVec;
template <> struct SpinVec<uint8_t> {
    using type = u32x4;
    static constexpr int n = 16;
    __device__ static __forceinline__ u32x4 flip_at(u32x4 v, int idx) {
        const int sh = (idx & 3) * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t d = v[q];
            const uint32_t nb = (((d >> sh) & 0xffu) == 0) ? 1u : 0u;
            const uint32_t nd = (d & ~(0xffu << sh)) | (nb << sh);
            v[q] = (q == (idx >> 2)) ? nd : d;
        }
        return v;
    }
};
template <> struct SpinVec<float> {
    using type = f32x4;
    static constexpr int n = 4;
    __device__ static __forceinline__ f32x4 flip_at(f32x4 v, int idx) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (q == idx) ? (v[q] == 0.0f ? 1.0f : 0.0f) : v[q];
        return v;
    }
};

// 64-bit ballot as two dwords
__device__ __forceinline__ uint64_t ballot64(bool p) { return __ballot(p); }

// Load the tile of envs [b0, b0+64) x nodes [0, N) into words[0..N).
// VEC = true requires row starts to be 16-byte aligned (x aligned and N * sizeof(T) % 16 == 0).
// W waves of one workgroup may share the job (wave w of W takes every W-th batch of columns); every
// wave sees all 64 envs, so each ballot still yields a complete word.  Callers sync afterwards.
template <typename T, bool VEC>
__device__ __forceinline__ void tile_load_bits(const T* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                               uint64_t* __restrict__ words, int lane, int w = 0, int W = 1) {
    const int64_t b = b0 + lane;
    const bool valid = b < B;
    const T* row = x + (valid ? b : 0) * N;
    if constexpr (VEC && sizeof(T) == 1) {
        const u32x4* rv = reinterpret_cast<const u32x4*>(row);
        const int64_t nv = N >> 4;
        constexpr int DEPTH = 8;  // row loads in flight per lane (the loop is latency-bound otherwise)
        for (int64_t i0 = (int64_t)w * DEPTH; i0 < nv; i0 += (int64_t)W * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q)
                v[q] = (valid && i0 + q < nv) ? rv[i0 + q] : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                if (i0 + q < nv) {
                    const uint32_t d[4] = {v[q][0], v[q][1], v[q][2], v[q][3]};
                    uint64_t mine = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        uint64_t w = ballot64(((d[k >> 2] >> ((k & 3) * 8)) & 0xffu) != 0);
                        if (lane == k) mine = w;
                    }
                    if (lane < 16) words[((i0 + q) << 4) + lane] = mine;
                }
            }
        }
    } else if constexpr (VEC && sizeof(T) == 4) {
        const f32x4* rv = reinterpret_cast<const f32x4*>(row);
        const int64_t nv = N >> 2;
#pragma unroll 4
        for (int64_t i = w; i < nv; i += W) {
            f32x4 v = valid ? rv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            const float d[4] = {v[0], v[1], v[2], v[3]};
            uint64_t mine = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint64_t w = ballot64(spin_is_set(d[k]));
                if (lane == k) mine = w;
            }
            if (lane < 4) words[(i << 2) + lane] = mine;
        }
    } else {
        for (int64_t n0 = (int64_t)w * 64; n0 < N; n0 += (int64_t)W * 64) {
            uint64_t mine = 0;
            const int lim = (int)((N - n0) < 64 ? (N - n0) : 64);
            for (int k = 0; k < lim; ++k) {
                T v = valid ? row[n0 + k] : T(0);
                uint64_t w = ballot64(spin_is_set(v));
                if (lane == k) mine = w;
            }
            if (lane < lim) words[n0 + lane] = mine;
        }
    }
}

// Write the tile back as env-major bytes (uint8 0|1).
template <bool VEC>
__device__ __forceinline__ void tile_store_bytes(uint8_t* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                                 const uint64_t* __restrict__ words, int lane, int w = 0, int W = 1) {
    const int64_t b = b0 + lane;
    if (b >= B) return;
    uint8_t* row = x + b * N;
    const int half = lane >> 5, sh = lane & 31;
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(words);
    if constexpr (VEC) {
        u32x4* rv = reinterpret_cast<u32x4*>(row);
        const int64_t nv = N >> 4;
        for (int64_t i = w; i < nv; i += W) {
            uint32_t d[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                uint32_t bit = (w32[(((i << 4) + k) << 1) + half] >> sh) & 1u;
                d[k >> 2] |= bit << ((k & 3) * 8);
            }
            rv[i] = u32x4{d[0], d[1], d[2], d[3]};
        }
    } else {
        for (int64_t n = w; n < N; n += W) row[n] = (uint8_t)((w32[(n << 1) + half] >> sh) & 1u);
    }
}

// ---- bit-sliced (vertical) counters: plane p holds bit p of 64 independent counts.
__device__ __forceinline__ void csa(uint64_t& hi, uint64_t& lo, uint64_t a, uint64_t b, uint64_t c) {
    const uint64_t u = a ^ b;
    hi = (a & b) | (u & c);
    lo = u ^ c;
}

__device__ __forceinline__ uint64_t shfl_xor64(uint64_t v, int mask) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    lo = __shfl_xor(lo, mask, 64);
    hi = __shfl_xor(hi, mask, 64);
    return ((uint64_t)hi << 32) | lo;
}

// sum of an int over the wave (all lanes get the result)
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

}  // namespace rls
