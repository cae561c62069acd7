// The 32-env "half tile": words32[n] is a 32-bit word whose bit e is the spin of node n in env (b0 + e) -- N * 4 bytes of
// LDS instead of N * 8, for graphs past the 64-env tile (20 224 < N <= ~40 000: rls_tile.h keeps one word per node in LDS) and
// as twice as many, half as long workgroups where a launch has few tiles per CU.  Same building blocks as rls_tile.h: a wave
// still turns 64 x 64 bits at a time -- lanes 0..31 hold the 32 envs' bytes of one 64-node block, lanes 32..63 the SAME envs'
// bytes of the next block, so the transpose leaves lane p with (r0 = the 32 envs of node p in the first block, r1 = in the
// second): two words32 per lane and transpose, nothing wasted.
#pragma once
#include "rls_tile.h"

namespace rls {

constexpr int kHalf = 32;   // envs per half tile

// carry-save adder on 32 counters at once
__device__ __forceinline__ void csa32(uint32_t& hi, uint32_t& lo, uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t h = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8), l = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
    hi = h;
    lo = l;
}

// Load the half tile of envs [b0, b0 + 32) x nodes [0, N) into words32[0..N).  W waves share the job (wave w takes every W-th
// 128-node chunk); callers sync afterwards.  `stage` = this wave's kStageBytes of LDS (16-byte aligned) or nullptr.
// Byte spins whose rows are 16-byte multiples on a 16-byte base (VEC) go through the row-piece stage when there is one -- an
// instruction fetches 8 rows x 128 B, whole cache lines; the wave turns the corner through 4 KB of LDS: lane l = (env l & 31,
// block l >> 5) reads back the four 16-byte pieces of ITS 64 nodes -- else each lane reads its 64 bytes itself; anything else
// (float spins, ragged rows) takes one ballot per node.
template <bool XORW> __device__ __forceinline__ void put_word32(uint32_t* words32, int64_t n, uint32_t v) {
    if constexpr (XORW) words32[n] ^= v;   // (x ^ mask without a second tile: the same lane of the same wave owns a word in both passes)
    else words32[n] = v;
}

template <typename T, bool VEC, bool XORW = false>
__device__ __forceinline__ void tile32_load_bits(const T* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                                 uint32_t* __restrict__ words32, int lane, int w, int W, unsigned char* stage) {
    const int env = lane & (kHalf - 1), blk = lane >> 5;
    const int64_t b = b0 + env;
    const bool valid = b < B;
    if constexpr (VEC && sizeof(T) == 1) {
        if ((N & 15) == 0) {
            const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
            const int64_t nchunk = (N + 127) >> 7;                       // 128-node chunks = one transpose each
            const BitXpose xc = bit_xpose_consts(lane);
            const int node = xc.node;                                    // the column -> node map of pack_bits
            if (stage != nullptr) {
                const int r = lane & 7, j = lane >> 3;                   // row within the instruction's 8, piece 0..7 of the row's 128 B
                constexpr int DEPTH = 2;
                for (int64_t ch0 = w; ch0 < nchunk; ch0 += (int64_t)W * DEPTH) {
                    u32x4 g[DEPTH][4];
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        const int64_t off = ((ch0 + (int64_t)d * W) << 7) + j * 16;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int64_t rw = b0 + 8 * i + r;
                            g[d][i] = (rw < B && off < N) ? *reinterpret_cast<const u32x4*>(xb + rw * N + off) : u32x4{0, 0, 0, 0};
                        }
                    }
#pragma unroll
                    for (int d = 0; d < DEPTH; ++d) {
                        const int64_t ch = ch0 + (int64_t)d * W;
                        if (ch < nchunk) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(stage + (j * 32 + 8 * i + r) * 16) = g[d][i];
                            asm volatile("" ::: "memory");               // LDS ops of one wave execute in order
                            u32x4 v[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4*>(stage + ((4 * blk + q) * 32 + env) * 16);
                            asm volatile("" ::: "memory");
                            uint32_t r0 = pack_bits(v[0], v[1]), r1 = pack_bits(v[2], v[3]);
                            bit_transpose64(r0, r1, xc);
                            const int64_t n = (ch << 7) + node;
                            if (n < N) put_word32<XORW>(words32, n, r0);
                            if (n + 64 < N) put_word32<XORW>(words32, n + 64, r1);
                        }
                    }
                }
                return;
            }
            const u32x4* rv = reinterpret_cast<const u32x4*>(xb + (valid ? b : 0) * N);
            const int64_t nv = N >> 4;
            for (int64_t ch = w; ch < nchunk; ch += W) {
                u32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t idx = (ch << 3) + 4 * blk + q;
                    v[q] = (valid && idx < nv) ? rv[idx] : u32x4{0, 0, 0, 0};
                }
                uint32_t r0 = pack_bits(v[0], v[1]), r1 = pack_bits(v[2], v[3]);
                bit_transpose64(r0, r1, xc);
                const int64_t n = (ch << 7) + node;
                if (n < N) put_word32<XORW>(words32, n, r0);
                if (n + 64 < N) put_word32<XORW>(words32, n + 64, r1);
            }
            return;
        }
    }
    // one ballot per node: lanes 0..31 = the half tile's envs (lanes 32..63 vote 0)
    const T* row = x + (valid ? b : 0) * N;
    for (int64_t n0 = (int64_t)w * 64; n0 < N; n0 += (int64_t)W * 64) {
        uint32_t mine = 0;
        const int lim = (int)((N - n0) < 64 ? (N - n0) : 64);
        for (int k = 0; k < lim; ++k) {
            const T v = (valid && blk == 0) ? row[n0 + k] : T(0);
            const uint32_t wd = (uint32_t)ballot64(spin_is_set(v));
            if (lane == k) mine = wd;
        }
        if (lane < lim) put_word32<XORW>(words32, n0 + lane, mine);
    }
}

// Write the half tile back as env-major bytes (0 | 1); rows with store_row == false stay untouched.  Every lane of the wave
// takes part whatever it stores.
template <bool VEC>
__device__ __forceinline__ void tile32_store_bytes(uint8_t* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                                   const uint32_t* __restrict__ words32, int lane, int w, int W, bool store_row) {
    const int env = lane & (kHalf - 1), blk = lane >> 5;
    const int64_t b = b0 + env;
    const bool valid = b < B && store_row;
    uint8_t* row = x + (b < B ? b : 0) * N;
    if constexpr (VEC) {
        if ((N & 15) == 0) {
            // inverse of the load: lane p fetches the two words of node p (blocks 2c, 2c + 1), the transpose hands lane l = (env,
            // block) that env's 64 bits of that block, unpacked to 64 bytes = four 16-byte stores
            u32x4* rv = reinterpret_cast<u32x4*>(row);
            const int64_t nv = N >> 4;
            const int64_t nchunk = (N + 127) >> 7;
            const BitXpose xc = bit_xpose_consts(lane);
            for (int64_t ch = w; ch < nchunk; ch += W) {
                const int64_t n = (ch << 7) + xc.node;
                uint32_t r0 = (n < N) ? words32[n] : 0u, r1 = (n + 64 < N) ? words32[n + 64] : 0u;
                bit_transpose64(r0, r1, xc);
                u32x4 v[4];
                unpack_bits(r0, v[0], v[1]);
                unpack_bits(r1, v[2], v[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t idx = (ch << 3) + 4 * blk + q;
                    if (valid && idx < nv) rv[idx] = v[q];
                }
            }
            return;
        }
    }
    if (!valid || blk != 0) return;
    for (int64_t n = w; n < N; n += W) row[n] = (uint8_t)((words32[n] >> env) & 1u);
}

// K1 core on a half tile: every lane takes every 64th stored edge, XORs the two 32-env words and feeds a bit-sliced
// Harley-Seal counter (rls_cutcount.h on 32-bit planes); the 64 lanes' counts are summed by transposing two planes at a time
// across the wave (lanes 0..31 receive plane p of env lane, lanes 32..63 plane p + 1 of env lane - 32) and popcounting.
// Returns, in lanes 0..31, wave w's partial count for env b0 + lane (lanes 32..63: 0).
template <int P>
__device__ __forceinline__ int64_t tile32_cut_count(const uint32_t* __restrict__ words32, const int32_t* __restrict__ eu,
                                                    const int32_t* __restrict__ ev, int64_t E, int lane, int w, int W) {
    constexpr int PL = (P - 5) < 5 ? 5 : (P - 5);           // per-lane count <= ceil(E / 64) < 2^(P - 5)
    constexpr int PLE = (PL + 1) & ~1;
    uint32_t ones = 0, twos = 0, fours = 0, eights = 0, c[PLE];
#pragma unroll
    for (int p = 0; p < PLE; ++p) c[p] = 0;
    auto eight = [&](const uint32_t (&d)[8]) -> uint32_t {
        uint32_t twosA, twosB, foursA, foursB, e8;
        csa32(twosA, ones, ones, d[0], d[1]);
        csa32(twosB, ones, ones, d[2], d[3]);
        csa32(foursA, twos, twos, twosA, twosB);
        csa32(twosA, ones, ones, d[4], d[5]);
        csa32(twosB, ones, ones, d[6], d[7]);
        csa32(foursB, twos, twos, twosA, twosB);
        csa32(e8, fours, fours, foursA, foursB);
        return e8;
    };
    auto block16 = [&](const uint32_t (&dA)[8], const uint32_t (&dB)[8]) {
        const uint32_t eA = eight(dA), eB = eight(dB);
        uint32_t carry;
        csa32(carry, eights, eights, eA, eB);
#pragma unroll
        for (int p = 4; p < PL; ++p) {
            const uint32_t t = c[p] & carry;
            c[p] ^= carry;
            carry = t;
        }
    };
    constexpr int64_t BLK = 16 * kWave;
    const int64_t nfull = E / BLK;
    for (int64_t blk = w; blk < nfull; blk += W) {
        const int32_t* pu = eu + blk * BLK + lane;
        const int32_t* pv = ev + blk * BLK + lane;
        int u[16], v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { u[k] = pu[k * kWave]; v[k] = pv[k * kWave]; }
        uint32_t dA[8], dB[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dA[k] = words32[u[k]] ^ words32[v[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) dB[k] = words32[u[8 + k]] ^ words32[v[8 + k]];
        block16(dA, dB);
    }
    if (nfull * BLK < E && (nfull % W) == w) {   // the ragged last block: clamped, unconditional loads, masked afterwards
        int u[16], v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t e = nfull * BLK + k * kWave + lane;
            const int64_t ec = e < E ? e : E - 1;
            u[k] = eu[ec];
            v[k] = ev[ec];
        }
        uint32_t dA[8], dB[8];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t e = nfull * BLK + k * kWave + lane;
            const uint32_t d = (e < E) ? (words32[u[k]] ^ words32[v[k]]) : 0u;
            if (k < 8) dA[k] = d; else dB[k - 8] = d;
        }
        block16(dA, dB);
    }
    c[0] = ones; c[1] = twos; c[2] = fours; c[3] = eights;
    const BitXpose xc = bit_xpose_consts(lane);
    const int half = lane >> 5;
    int64_t total = 0;
#pragma unroll
    for (int p = 0; p < PLE; p += 2) {
        // row l = (c[p + 1] : c[p]) of lane l; column q < 32 = bit q of plane p over the 64 lanes, column 32 + q = of plane p + 1
        uint32_t r0 = c[p], r1 = c[p + 1];
        bit_transpose64(r0, r1, xc);
        const int cnt = __builtin_popcount(r0) + __builtin_popcount(r1);
        // this lane holds column xc.node-independent index = lane itself (plain 64 x 64 transpose: no pack_bits permutation here)
        total += (int64_t)cnt << (p + half);
    }
    // lane q (< 32) has the even planes of env q, lane 32 + q the odd ones: add the two
    const int64_t other = shfl_xor64((uint64_t)total, 32);
    return half == 0 ? total + other : 0;
}

}  // namespace rls
