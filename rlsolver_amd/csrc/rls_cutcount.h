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
// P = number of planes (E' < 2^P).  Wave w of W takes every W-th block of 512 edges and the
// result is that wave's partial count (sum the W partials, see block_sum_partials).
//
// Full blocks run without bounds checks and with all 16 endpoint loads issued before the
// 16 LDS reads (a per-edge `if (e < E)` made hipcc serialise load -> wait -> read per edge:
// 8 L2 round trips per block, the whole kernel at 10 % of its HBM bound).
// =====================================================================================
template <int P>
__device__ __forceinline__ void hs_block(const uint64_t (&d)[8], uint64_t& ones, uint64_t& twos, uint64_t& fours,
                                         uint64_t (&c)[P]) {
    constexpr int PL = (P - 5) < 4 ? 4 : (P - 5);  // per-lane count <= ceil(E/64) < 2^(P-5)
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

template <int P>
__device__ __forceinline__ int64_t tile_cut_count(const uint64_t* __restrict__ words,
                                                  const int32_t* __restrict__ eu,
                                                  const int32_t* __restrict__ ev,
                                                  int64_t E, int lane, int w = 0, int W = 1) {
    uint64_t c[P];
#pragma unroll
    for (int p = 0; p < P; ++p) c[p] = 0;
    uint64_t ones = 0, twos = 0, fours = 0;
    constexpr int64_t BLK = 8 * kWave;
    const int64_t nfull = E / BLK;

    for (int64_t blk = w; blk < nfull; blk += W) {
        const int32_t* pu = eu + blk * BLK + lane;
        const int32_t* pv = ev + blk * BLK + lane;
        int u[8], v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { u[k] = pu[k * kWave]; v[k] = pv[k * kWave]; }
        uint64_t a[8], b[8], d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a[k] = words[u[k]]; b[k] = words[v[k]]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = a[k] ^ b[k];
        hs_block<P>(d, ones, twos, fours, c);
    }
    if (nfull * BLK < E && (nfull % W) == w) {  // the ragged last block
        uint64_t d[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t e = nfull * BLK + k * kWave + lane;
            const bool in = e < E;
            const int u = in ? eu[e] : 0, v = in ? ev[e] : 0;
            d[k] = in ? (words[u] ^ words[v]) : 0ull;
        }
        hs_block<P>(d, ones, twos, fours, c);
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

// Sum the per-wave partial counts of a W-wave workgroup through LDS (scratch: W*64 int64).
// Every wave returns the full total for its lane's env.
template <int W>
__device__ __forceinline__ int64_t block_sum_partials(int64_t partial, int64_t* scratch, int lane, int w) {
    if constexpr (W == 1) return partial;
    scratch[w * kWave + lane] = partial;
    __syncthreads();
    int64_t t = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) t += scratch[k * kWave + lane];
    return t;
}

inline int pick_planes(int64_t E) {
    if (E < (1 << 12)) return 12;
    if (E < (1 << 16)) return 16;
    if (E < (1 << 20)) return 20;
    if (E < (1 << 24)) return 24;
    return 0;
}

}  // namespace rls
