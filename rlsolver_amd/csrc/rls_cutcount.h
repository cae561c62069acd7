// Bit-sliced cut counting over a 64-env bit tile (shared by the MaxCut and MCPG kernels).
#pragma once
#include "rls_tile.h"

namespace rls {

// =====================================================================================
// K1 core: cut value of 64 envs held as a bit tile.
// Each lane takes every 64th stored edge, XORs the two 64-env words (one XOR = one edge
// in 64 envs) and feeds the result into a bit-sliced Harley-Seal counter (8 edges per
// block: 7 carry-save adders + one ripple into the upper planes).  The 64 per-lane
// bit-sliced counts are then summed with a butterfly of bit-sliced full adders, after
// which every lane holds the total planes and extracts its own env's count.
// P = number of planes (E' < 2^P).
// =====================================================================================
template <int P>
__device__ __forceinline__ int64_t tile_cut_count(const uint64_t* __restrict__ words,
                                                  const int32_t* __restrict__ eu,
                                                  const int32_t* __restrict__ ev,
                                                  int64_t E, int lane) {
    uint64_t c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) c[p] = 0;
    uint64_t ones = 0, twos = 0, fours = 0;
    constexpr int PL = (P - 5) < 4 ? 4 : (P - 5);  // per-lane count <= ceil(E/64) < 2^(P-5)

    for (int64_t base = 0; base < E; base += 8 * kWave) {
        uint64_t d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t e = base + k * kWave + lane;
            if (e < E) {
                const int u = eu[e], v = ev[e];
                d[k] = words[u] ^ words[v];
            } else {
                d[k] = 0;
            }
        }
        uint64_t twosA, twosB, foursA, foursB, eights;
        csa(twosA, ones, ones, d[0], d[1]);
        csa(twosB, ones, ones, d[2], d[3]);
        csa(foursA, twos, twos, twosA, twosB);
        csa(twosA, ones, ones, d[4], d[5]);
        csa(twosB, ones, ones, d[6], d[7]);
        csa(foursB, twos, twos, twosA, twosB);
        csa(eights, fours, fours, foursA, foursB);
        uint64_t carry = eights;
#pragma unroll
        for (int p = 3; p < PL; ++p) {
            const uint64_t t = c[p] & carry;
            c[p] ^= carry;
            carry = t;
        }
    }
    c[0] = ones; c[1] = twos; c[2] = fours;

    // butterfly: after the step with mask m every lane holds the sum over its 2m-group
#pragma unroll
    for (int m = 1; m < kWave; m <<= 1) {
        uint64_t carry = 0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const uint64_t o = shfl_xor64(c[p], m);
            const uint64_t u = c[p] ^ o;
            const uint64_t s = u ^ carry;
            carry = (c[p] & o) | (u & carry);
            c[p] = s;
        }
    }
    int64_t total = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) total |= (int64_t)((c[p] >> lane) & 1ull) << p;
    return total;
}


inline int pick_planes(int64_t E) {
    if (E < (1 << 12)) return 12;
    if (E < (1 << 16)) return 16;
    if (E < (1 << 20)) return 20;
    if (E < (1 << 24)) return 24;
    return 0;
}

}  // namespace rls
