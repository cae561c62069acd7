// Bit-sliced cut counting over a 64-env bit tile (shared by the MaxCut and MCPG kernels).
#pragma once
#include "rls_tile.h"

namespace rls {
#ifdef RLS_PROF
static __device__ unsigned long long g_prof[8];   // per translation unit (dev profiling only)
static __device__ unsigned long long g_prof_w[5 * 65536];   // five stamps per wave, plain stores (atomics distort)
#endif

// =====================================================================================
// K1 core: cut value of 64 envs held as a bit tile.
// Each lane takes every 64th stored edge, XORs the two 64-env words (one XOR = one edge
// in 64 envs) and feeds the result into a bit-sliced Harley-Seal counter (16 edges per
// block: 15 carry-save adders + one ripple into the upper planes).  The 64 lanes' bit-sliced
// counts are then summed by TRANSPOSING each plane across the wave (bit_transpose64): lane e ends up
// with "bit p of every lane's count for env e", whose popcount weighs 2^p.  (The butterfly of
// bit-sliced full adders this replaces was 1400 of the kernel's 3000 VALU instructions per wave.)
// P = number of planes (E' < 2^P).  Wave w of W takes every W-th block of 1024 edges and the
// result is that wave's partial count (sum the W partials, see block_sum_partials).
//
// Full blocks run without bounds checks and with all endpoint loads issued before the
// LDS reads (a per-edge `if (e < E)` made hipcc serialise load -> wait -> read per edge:
// 8 L2 round trips per block, the whole kernel at 10 % of its HBM bound).
// =====================================================================================
template <int P> struct HsState {
    static constexpr int PL = (P - 5) < 5 ? 5 : (P - 5);   // per-lane count <= ceil(E/64) < 2^(P-5)
    uint64_t ones = 0, twos = 0, fours = 0, eights = 0;
    uint64_t c[PL];                                        // planes 4.. live in c[4..]; c[0..3] unused until finish()
};

template <int P>
__device__ __forceinline__ uint64_t hs_eight(const uint64_t (&d)[8], HsState<P>& st) {
    uint64_t twosA, twosB, foursA, foursB, eights;
    csa(twosA, st.ones, st.ones, d[0], d[1]);
    csa(twosB, st.ones, st.ones, d[2], d[3]);
    csa(foursA, st.twos, st.twos, twosA, twosB);
    csa(twosA, st.ones, st.ones, d[4], d[5]);
    csa(twosB, st.ones, st.ones, d[6], d[7]);
    csa(foursB, st.twos, st.twos, twosA, twosB);
    csa(eights, st.fours, st.fours, foursA, foursB);
    return eights;
}

template <int P>
__device__ __forceinline__ void hs_ripple16(uint64_t carry, HsState<P>& st) {
#pragma unroll
    for (int p = 4; p < HsState<P>::PL; ++p) {
        const uint64_t t = st.c[p] & carry;
        st.c[p] ^= carry;
        carry = t;
    }
}

template <int P>
__device__ __forceinline__ void hs_block16(const uint64_t (&dA)[8], const uint64_t (&dB)[8], HsState<P>& st) {
    const uint64_t eA = hs_eight<P>(dA, st), eB = hs_eight<P>(dB, st);
    uint64_t sixteens;
    csa(sixteens, st.eights, st.eights, eA, eB);
    hs_ripple16<P>(sixteens, st);
}

template <int P>
__device__ __forceinline__ int64_t tile_cut_count(const uint64_t* __restrict__ words,
                                                  const int32_t* __restrict__ eu,
                                                  const int32_t* __restrict__ ev,
                                                  int64_t E, int lane, int w = 0, int W = 1) {
    HsState<P> st;
#pragma unroll
    for (int p = 0; p < HsState<P>::PL; ++p) st.c[p] = 0;
    constexpr int64_t BLK = 16 * kWave;
    const int64_t nfull = E / BLK;

    for (int64_t blk = w; blk < nfull; blk += W) {
        const int32_t* pu = eu + blk * BLK + lane;
        const int32_t* pv = ev + blk * BLK + lane;
        int u[16], v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { u[k] = pu[k * kWave]; v[k] = pv[k * kWave]; }
        uint64_t dA[8], dB[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) dA[k] = words[u[k]] ^ words[v[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) dB[k] = words[u[8 + k]] ^ words[v[8 + k]];
        hs_block16<P>(dA, dB, st);
    }
    if (nfull * BLK < E && (nfull % W) == w) {  // the ragged last block
        // clamped, unconditional loads (all 32 in flight at once -- a guarded load per edge made this one
        // block cost 16 serial L2 round trips, several times the rest of the loop), masked afterwards
        int u[16], v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t e = nfull * BLK + k * kWave + lane;
            const int64_t ec = e < E ? e : E - 1;
            u[k] = eu[ec];
            v[k] = ev[ec];
        }
        uint64_t dA[8], dB[8];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t e = nfull * BLK + k * kWave + lane;
            const uint64_t d = (e < E) ? (words[u[k]] ^ words[v[k]]) : 0ull;
            if (k < 8) dA[k] = d; else dB[k - 8] = d;
        }
        hs_block16<P>(dA, dB, st);
    }
    st.c[0] = st.ones; st.c[1] = st.twos; st.c[2] = st.fours; st.c[3] = st.eights;

    // sum over the 64 lanes: transpose each plane, popcount, weigh
    const BitXpose xc = bit_xpose_consts(lane);
    int64_t total = 0;
#pragma unroll
    for (int p = 0; p < HsState<P>::PL; ++p) {
        uint32_t r0 = (uint32_t)st.c[p], r1 = (uint32_t)(st.c[p] >> 32);
        bit_transpose64(r0, r1, xc);
        total += (int64_t)(__builtin_popcount(r0) + __builtin_popcount(r1)) << p;
    }
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
