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
// 16 bytes of a byte row at offset `off` (a multiple of 16) of a row whose start is 16- (a16) or only 8-byte aligned (rows of 8-byte
// multiples: every other row starts mid-vector -- the Gset sizes 1000 ... 9000): one 16-byte access or two 8-byte ones; bytes at and
// past N are neither read nor written (with N % 16 == 8 the last piece is half a piece)
typedef uint32_t u32x2_t32 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 row_ld16(const uint8_t* p, bool a16, int64_t off, int64_t N) {
    if (a16) return *reinterpret_cast<const u32x4*>(p);
    const u32x2_t32 lo = *reinterpret_cast<const u32x2_t32*>(p);
    const u32x2_t32 hi = (off + 8 < N) ? *reinterpret_cast<const u32x2_t32*>(p + 8) : u32x2_t32{0, 0};
    return u32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ void row_st16(uint8_t* p, bool a16, int64_t off, int64_t N, u32x4 v) {
    if (a16) { *reinterpret_cast<u32x4*>(p) = v; return; }
    *reinterpret_cast<u32x2_t32*>(p) = u32x2_t32{v.x, v.y};
    if (off + 8 < N) *reinterpret_cast<u32x2_t32*>(p + 8) = u32x2_t32{v.z, v.w};
}

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
        if ((N & 7) == 0) {
            const bool a16 = (N & 15) == 0;
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
                            g[d][i] = (rw < B && off < N) ? row_ld16(xb + rw * N + off, a16, off, N) : u32x4{0, 0, 0, 0};
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
            const uint8_t* rowb = xb + (valid ? b : 0) * N;
            for (int64_t ch = w; ch < nchunk; ch += W) {
                u32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t off = ((ch << 3) + 4 * blk + q) << 4;
                    v[q] = (valid && off < N) ? row_ld16(rowb + off, a16, off, N) : u32x4{0, 0, 0, 0};
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
                                                   const uint32_t* __restrict__ words32, int lane, int w, int W, bool store_row,
                                                   unsigned char* stage = nullptr) {
    const int env = lane & (kHalf - 1), blk = lane >> 5;
    const int64_t b = b0 + env;
    const bool valid = b < B && store_row;
    uint8_t* row = x + (b < B ? b : 0) * N;
    const bool a16 = (N & 15) == 0;
    if constexpr (VEC) {
        if ((N & 7) == 0 && stage != nullptr) {
            // through the row-piece stage (the loader's corner turn backwards): lane l = (env, block) parks its four 16-byte pieces,
            // instruction i then writes rows 8 i .. 8 i + 7 as 128-byte runs (lane -> row 8 i + (l & 7), piece l >> 3)
            const uint32_t rows_ok = (uint32_t)ballot64(valid && blk == 0);
            const int64_t nchunk = (N + 127) >> 7;
            const BitXpose xc = bit_xpose_consts(lane);
            const int r = lane & 7, j = lane >> 3;
            for (int64_t ch = w; ch < nchunk; ch += W) {
                const int64_t n = (ch << 7) + xc.node;
                uint32_t r0 = (n < N) ? words32[n] : 0u, r1 = (n + 64 < N) ? words32[n + 64] : 0u;
                bit_transpose64(r0, r1, xc);
                u32x4 v[4];
                unpack_bits(r0, v[0], v[1]);
                unpack_bits(r1, v[2], v[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(stage + ((4 * blk + q) * 32 + env) * 16) = v[q];
                asm volatile("" ::: "memory");   // LDS ops of one wave execute in order
                const int64_t off = (ch << 7) + j * 16;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int rr = 8 * i + r;
                    const u32x4 o = *reinterpret_cast<const u32x4*>(stage + (j * 32 + rr) * 16);
                    if (((rows_ok >> rr) & 1u) && off < N) row_st16(x + (b0 + rr) * N + off, a16, off, N, o);
                }
                asm volatile("" ::: "memory");
            }
            return;
        }
        if ((N & 7) == 0) {
            // inverse of the load: lane p fetches the two words of node p (blocks 2c, 2c + 1), the transpose hands lane l = (env,
            // block) that env's 64 bits of that block, unpacked to 64 bytes = four 16-byte stores
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
                    const int64_t off = ((ch << 3) + 4 * blk + q) << 4;
                    if (valid && off < N) row_st16(row + off, a16, off, N, v[q]);
                }
            }
            return;
        }
    }
    if (!valid || blk != 0) return;
    for (int64_t n = w; n < N; n += W) row[n] = (uint8_t)((words32[n] >> env) & 1u);
}

// ---- NARROW tiles (round 5): 16 or 8 envs per workgroup, words of 16 / 8 bits -- 2 N / N bytes of LDS, for graphs past the half tile
// (N > ~40 000: up to ~80 000 / ~160 000 nodes).  The schedules, the counters and the compare are the half tile's (planes in 32-bit
// registers whose upper bits stay zero); only the LDS word and the byte <-> bit corner turn differ: a wave still turns 64 x 64 bits
// at a time, row l of the transpose = env l % E of node block l / E (E = 16: four blocks of 64 nodes per turn), so lane p ends
// with the E-bit words of node p in 64 / E consecutive blocks.  Per env the work is 32 / E times the half tile's (the edge list and
// the schedule are walked once per E envs) -- against one env per WAVE on a byte row, which these sizes fell to before.
template <typename WT> struct narrow_tile {
    static constexpr int E = 8 * (int)sizeof(WT);       // envs per tile
    static constexpr int NB = 64 / E;                   // 64-node blocks per transpose
    static constexpr uint32_t MASK = sizeof(WT) == 4 ? 0xFFFFFFFFu : ((1u << (E & 31)) - 1u);
};

template <typename WT, bool XORW> __device__ __forceinline__ void put_word_n(WT* words, int64_t n, uint32_t v) {
    if constexpr (XORW) words[n] = (WT)((uint32_t)words[n] ^ v);
    else words[n] = (WT)v;
}

// env-major bytes -> narrow bit tile.  VEC: rows are 16-byte multiples on a 16-byte base (each lane reads the 64 bytes of its (env,
// block) itself); anything else takes one ballot per node.
template <typename T, typename WT, bool VEC, bool XORW = false>
__device__ __forceinline__ void tilen_load_bits(const T* __restrict__ x, int64_t B, int64_t N, int64_t b0, WT* __restrict__ words,
                                                int lane, int w, int W) {
    constexpr int E = narrow_tile<WT>::E, NB = narrow_tile<WT>::NB;
    const int env = lane % E, blk = lane / E;
    const int64_t b = b0 + env;
    const bool valid = b < B;
    if constexpr (VEC && sizeof(T) == 1) {
        if ((N & 7) == 0) {
            // rows of 16-byte multiples: 16-byte loads; of 8-byte multiples (every other row starts mid-vector -- the Gset sizes 1000,
            // 3000, 5000, 7000, 9000): the same 64 bytes as eight 8-byte loads
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
            const u32x4* rv = reinterpret_cast<const u32x4*>(xb + (valid ? b : 0) * N);
            const u32x2* r2 = reinterpret_cast<const u32x2*>(rv);
            const bool a16 = (N & 15) == 0;
            const int64_t nv = N >> 4, n8 = N >> 3;
            const int64_t nchunk = (N + 64 * NB - 1) / (64 * NB);        // chunk = NB blocks of 64 nodes = one transpose
            const BitXpose xc = bit_xpose_consts(lane);
            for (int64_t ch = w; ch < nchunk; ch += W) {
                u32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t idx = (ch * NB + blk) * 4 + q;
                    if (a16) {
                        v[q] = (valid && idx < nv) ? rv[idx] : u32x4{0, 0, 0, 0};
                    } else {
                        const u32x2 lo = (valid && 2 * idx < n8) ? r2[2 * idx] : u32x2{0, 0};
                        const u32x2 hi = (valid && 2 * idx + 1 < n8) ? r2[2 * idx + 1] : u32x2{0, 0};
                        v[q] = u32x4{lo.x, lo.y, hi.x, hi.y};
                    }
                }
                uint32_t r0 = pack_bits(v[0], v[1]), r1 = pack_bits(v[2], v[3]);
                bit_transpose64(r0, r1, xc);
                // this lane: node xc.node of each of the NB blocks; rows k E .. k E + E - 1 of the turn = block k
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const uint32_t half = (k * E) < 32 ? r0 : r1;
                    const uint32_t wd = (half >> ((k * E) & 31)) & narrow_tile<WT>::MASK;
                    const int64_t n = (ch * NB + k) * 64 + xc.node;
                    if (n < N) put_word_n<WT, XORW>(words, n, wd);
                }
            }
            return;
        }
    }
    const T* row = x + (valid ? b : 0) * N;
    for (int64_t n0 = (int64_t)w * 64; n0 < N; n0 += (int64_t)W * 64) {
        uint32_t mine = 0;
        const int lim = (int)((N - n0) < 64 ? (N - n0) : 64);
        for (int k = 0; k < lim; ++k) {
            const T v = (valid && blk == 0) ? row[n0 + k] : T(0);
            const uint32_t wd = (uint32_t)ballot64(spin_is_set(v)) & narrow_tile<WT>::MASK;
            if (lane == k) mine = wd;
        }
        if (lane < lim) put_word_n<WT, XORW>(words, n0 + lane, mine);
    }
}

// narrow bit tile -> env-major bytes (0 | 1); rows with store_row == false stay untouched (store_row: per lane, by lane % E)
template <typename WT, bool VEC>
__device__ __forceinline__ void tilen_store_bytes(uint8_t* __restrict__ x, int64_t B, int64_t N, int64_t b0, const WT* __restrict__ words,
                                                  int lane, int w, int W, bool store_row) {
    constexpr int E = narrow_tile<WT>::E, NB = narrow_tile<WT>::NB;
    const int env = lane % E, blk = lane / E;
    const int64_t b = b0 + env;
    const bool valid = b < B && store_row;
    uint8_t* row = x + (b < B ? b : 0) * N;
    if constexpr (VEC) {
        if ((N & 7) == 0) {      // (16-byte stores, or 8-byte ones for rows of 8-byte multiples: tilen_load_bits)
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            u32x4* rv = reinterpret_cast<u32x4*>(row);
            u32x2* r2 = reinterpret_cast<u32x2*>(row);
            const bool a16 = (N & 15) == 0;
            const int64_t nv = N >> 4, n8 = N >> 3;
            const int64_t nchunk = (N + 64 * NB - 1) / (64 * NB);
            const BitXpose xc = bit_xpose_consts(lane);
            for (int64_t ch = w; ch < nchunk; ch += W) {
                uint32_t r0 = 0, r1 = 0;
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    const int64_t n = (ch * NB + k) * 64 + xc.node;
                    const uint32_t wd = (n < N) ? (uint32_t)words[n] : 0u;
                    if ((k * E) < 32) r0 |= wd << ((k * E) & 31);
                    else r1 |= wd << ((k * E) & 31);
                }
                bit_transpose64(r0, r1, xc);
                u32x4 v[4];
                unpack_bits(r0, v[0], v[1]);
                unpack_bits(r1, v[2], v[3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t idx = (ch * NB + blk) * 4 + q;
                    if (a16) {
                        if (valid && idx < nv) rv[idx] = v[q];
                    } else {
                        if (valid && 2 * idx < n8) r2[2 * idx] = u32x2{v[q].x, v[q].y};
                        if (valid && 2 * idx + 1 < n8) r2[2 * idx + 1] = u32x2{v[q].z, v[q].w};
                    }
                }
            }
            return;
        }
    }
    if (!valid || blk != 0) return;
    for (int64_t n = w; n < N; n += W) row[n] = (uint8_t)(((uint32_t)words[n] >> env) & 1u);
}

// K1 core on a half tile: every lane takes every 64th stored edge, XORs the two 32-env words and feeds a bit-sliced
// Harley-Seal counter (rls_cutcount.h on 32-bit planes); the 64 lanes' counts are summed by transposing two planes at a time
// across the wave (lanes 0..31 receive plane p of env lane, lanes 32..63 plane p + 1 of env lane - 32) and popcounting.
// Returns, in lanes 0..31, wave w's partial count for env b0 + lane (lanes 32..63: 0).
// (WT: the LDS word -- uint32_t for the half tile, uint16_t / uint8_t for the narrow tiles below, whose planes simply leave the upper
// bits of the 32-bit registers zero: lanes beyond the tile's envs end with a count of 0)
template <int P, typename WT = uint32_t>
__device__ __forceinline__ int64_t tile32_cut_count(const WT* __restrict__ words32, const int32_t* __restrict__ eu,
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
        for (int k = 0; k < 8; ++k) dA[k] = (uint32_t)words32[u[k]] ^ (uint32_t)words32[v[k]];
#pragma unroll
        for (int k = 0; k < 8; ++k) dB[k] = (uint32_t)words32[u[8 + k]] ^ (uint32_t)words32[v[8 + k]];
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
            const uint32_t d = (e < E) ? ((uint32_t)words32[u[k]] ^ (uint32_t)words32[v[k]]) : 0u;
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

// ---- the level-parallel sweep (rls_sweep.h: sweep_tile_levels) on half-tile words.  The SAME schedule: its entries are byte
// offsets of 64-bit words (node * 8), halved here; a group costs the same instructions per plane and HALF as many planes' halves.
template <int X>
__device__ __forceinline__ uint32_t lv32_lane_xor(uint32_t v) {
    if constexpr (X == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);
    else return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, X == 1 ? 0xB1 : 0x4E, 0xF, 0xF, false);
}
template <int X>
__device__ __forceinline__ void lv32_merge_planes(uint32_t (&pl)[8], uint32_t take) {
    uint32_t carry = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) csa32(carry, pl[p], pl[p], lv32_lane_xor<X>(pl[p]) & take, carry);
}
template <int NP>
__device__ __forceinline__ uint32_t lv32_le_const(const uint32_t (&pl)[8], uint32_t thr) {   // rls_tile.h: lv_le_const
    const int nt = (int)~thr;
    uint32_t c0 = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) c0 = __builtin_amdgcn_bitop3_b32(pl[p], (uint32_t)__builtin_amdgcn_sbfe(nt, p, 1), c0, 0xE8);
    return ~c0;
}

// the tile word of the node whose 64-bit-word byte offset is `nb` (the schedule's entries): the tile sits at LDS address 0
template <typename WT>
__device__ __forceinline__ uint32_t sweep32_word_at(uint32_t nb) {
    typedef const WT __attribute__((address_space(3))) lds_cwt;
    constexpr int SH = sizeof(WT) == 4 ? 1 : (sizeof(WT) == 2 ? 2 : 3);
    return (uint32_t)*(lds_cwt*)(uintptr_t)(nb >> SH);
}

template <int NC, typename WT>
__device__ __forceinline__ void sweep32_count_block(const uint32_t (&nb)[8], uint32_t own, uint32_t& ones, uint32_t& twos, uint32_t& fours,
                                                    uint32_t (&c)[5]) {
    uint32_t d[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) d[q] = sweep32_word_at<WT>(nb[q]) ^ own;   // padding = the node itself: 0
    uint32_t twosA, twosB, foursA, foursB, carry;
    csa32(twosA, ones, ones, d[0], d[1]);
    csa32(twosB, ones, ones, d[2], d[3]);
    csa32(foursA, twos, twos, twosA, twosB);
    csa32(twosA, ones, ones, d[4], d[5]);
    csa32(twosB, ones, ones, d[6], d[7]);
    csa32(foursB, twos, twos, twosA, twosB);
    csa32(carry, fours, fours, foursA, foursB);
#pragma unroll
    for (int p = 0; p < NC; ++p) {
        const uint32_t t = c[p] & carry;
        c[p] ^= carry;
        carry = t;
    }
}

// (rls_sweep.h: sweep_group_flips -- NB = the group's blocks when 1 or 2, 0 = the loop; blk = this lane's slab of block 2)
template <int NB, int NC, int NP, typename WT>
__device__ __forceinline__ uint32_t sweep32_group_flips(const int32_t* __restrict__ blk, int rounds, const uint32_t (&nb0)[8],
                                                        const uint32_t (&nb1)[8], uint32_t own, uint32_t thr, uint32_t lcode) {
    uint32_t ones = 0, twos = 0, fours = 0, c[5] = {0, 0, 0, 0, 0};
    if constexpr (NB == 1) {
        sweep32_count_block<NC, WT>(nb0, own, ones, twos, fours, c);
    } else if constexpr (NB == 2) {
        sweep32_count_block<NC, WT>(nb0, own, ones, twos, fours, c);
        sweep32_count_block<NC, WT>(nb1, own, ones, twos, fours, c);
    } else {
        uint32_t nb[8], nx[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { nb[q] = nb0[q]; nx[q] = nb1[q]; }
        for (int r0 = 0; r0 < rounds; r0 += 8, blk += 512) {
            u32x4 na, nbv;
            if (r0 + 16 < rounds) {
                na = *reinterpret_cast<const u32x4*>(blk);
                nbv = *reinterpret_cast<const u32x4*>(blk + 256);
            }
            sweep32_count_block<NC, WT>(nb, own, ones, twos, fours, c);
#pragma unroll
            for (int q = 0; q < 8; ++q) nb[q] = nx[q];
            nx[0] = na.x; nx[1] = na.y; nx[2] = na.z; nx[3] = na.w;
            nx[4] = nbv.x; nx[5] = nbv.y; nx[6] = nbv.z; nx[7] = nbv.w;
        }
    }
    const int gl = __builtin_amdgcn_readlane((int)lcode, 0);
    if (gl > 0) {
        uint32_t pv[8] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4]};
        lv32_merge_planes<1>(pv, 0u - (uint32_t)(lcode >= 1u));
        if (gl > 1) lv32_merge_planes<2>(pv, 0u - (uint32_t)(lcode >= 2u));
        if (gl > 2) lv32_merge_planes<4>(pv, 0u - (uint32_t)(lcode >= 3u));
        return lv32_le_const<8>(pv, thr);
    }
    const uint32_t pl[8] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4]};
    return lv32_le_const<NP>(pl, thr);
}

// a hub: lane = neighbour, per-lane counters over its rounds, every plane transposed across the wave and popcounted (two planes
// per transpose: lanes 0..31 get plane p of env lane, lanes 32..63 plane p + 1 of env lane - 32)
template <typename WT>
__device__ __forceinline__ uint32_t sweep32_hub_flips(const int32_t* __restrict__ ent, int rounds, const uint32_t (&nb0)[8], uint32_t own,
                                                      uint32_t deg, int lane) {     // ent = the record's entries (behind its 64 header words)
    uint32_t cv[8] = {0, 0, 0, 0, 0, 0, 0, 0};                      // rounds <= 64: 7 planes (+ 1 to pair them)
    auto add = [&](uint32_t off) {
        uint32_t carry = sweep32_word_at<WT>(off) ^ own;
#pragma unroll
        for (int p = 0; p < 7; ++p) { const uint32_t t = cv[p] & carry; cv[p] ^= carry; carry = t; }
    };
#pragma unroll
    for (int q = 0; q < 8; ++q) add(nb0[q]);
    for (int r = 8; r < rounds; ++r) add((uint32_t)ent[(r >> 3) * 512 + ((r >> 2) & 1) * 256 + lane * 4 + (r & 3)]);
    const BitXpose xc = bit_xpose_consts(lane);
    const int half = lane >> 5;
    int cnt = 0;
#pragma unroll
    for (int p = 0; p < 8; p += 2) {
        uint32_t r0 = cv[p], r1 = cv[p + 1];
        bit_transpose64(r0, r1, xc);
        cnt += (__builtin_popcount(r0) + __builtin_popcount(r1)) << (p + half);
    }
    cnt += __shfl_xor(cnt, 32, 64);
    return (uint32_t)ballot64(lane < 8 * (int)sizeof(WT) && (uint32_t)cnt <= (deg >> 1));
}

template <int W, typename WT = uint32_t>
__device__ __forceinline__ void sweep32_tile_levels(WT* words32, const int32_t* lvp, const int32_t* __restrict__ data, int64_t G,
                                                    int64_t N, int lane, int w) {
    constexpr uint32_t M = 0x3fffffffu;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)words32 != 0u) __builtin_trap();   // entries are LDS addresses (rls_sweep.h)
    uint32_t hdr = (uint32_t)N;
    uint32_t nb0[8], nb1[8];
    auto prefetch = [&](int64_t k) {
        if (k < G) {
            const int32_t* rec = data + ((uint32_t)lvp[k] & M);
            hdr = (uint32_t)rec[lane];
            const u32x4 a = *reinterpret_cast<const u32x4*>(rec + 64 + 4 * lane), b = *reinterpret_cast<const u32x4*>(rec + 320 + 4 * lane);
            const u32x4 c = *reinterpret_cast<const u32x4*>(rec + 576 + 4 * lane), d = *reinterpret_cast<const u32x4*>(rec + 832 + 4 * lane);
            nb0[0] = a.x; nb0[1] = a.y; nb0[2] = a.z; nb0[3] = a.w; nb0[4] = b.x; nb0[5] = b.y; nb0[6] = b.z; nb0[7] = b.w;
            nb1[0] = c.x; nb1[1] = c.y; nb1[2] = c.z; nb1[3] = c.w; nb1[4] = d.x; nb1[5] = d.y; nb1[6] = d.z; nb1[7] = d.w;
        }
    };
    prefetch(w);
    int num_levels = 0;
    for (int64_t k0 = 0; k0 < G; k0 += kWave)
        num_levels += __builtin_popcountll(ballot64(k0 + lane < G && (lvp[k0 + lane < G ? k0 + lane : G] >> 31) != 0));
    int chunk = 0, chunk_next = 0;
    int64_t cbase = -1;
    uint64_t lmask = 0;
    int lev_base = 0, passed = 0;
    for (int64_t k = w; k < G; k += W) {
        if ((k & ~(int64_t)63) != cbase) {
            cbase = k & ~(int64_t)63;
            lev_base += __builtin_popcountll(lmask);
            const int64_t a0 = cbase + lane <= G ? cbase + lane : G, a1 = cbase + 1 + lane <= G ? cbase + 1 + lane : G;
            chunk = lvp[a0];
            chunk_next = lvp[a1];
            lmask = ballot64(cbase + lane < G && (chunk >> 31) != 0);
        }
        const int need = lev_base + __builtin_popcountll(lmask & ((2ull << (k & 63)) - 1ull));
        for (; passed < need; ++passed) __syncthreads();
        const int64_t p0 = (uint32_t)__builtin_amdgcn_readlane(chunk, (int)(k & 63)) & M;
        const int64_t p1 = (uint32_t)__builtin_amdgcn_readlane(chunk_next, (int)(k & 63)) & M;
        const int rounds = (int)((p1 - p0) >> 6) - 1;
        if (((uint32_t)__builtin_amdgcn_readlane(chunk, (int)(k & 63)) >> 30) & 1u) {
            const uint32_t hnode = (uint32_t)__builtin_amdgcn_readlane((int)hdr, 0), hdeg = (uint32_t)__builtin_amdgcn_readlane((int)hdr, 1);
            const uint32_t hown = (uint32_t)words32[hnode];
            const uint32_t hflip = sweep32_hub_flips<WT>(data + p0 + 64, rounds, nb0, hown, hdeg, lane);
            if (lane == 0) words32[hnode] = (WT)(hown ^ hflip);
            prefetch(k + W);
            continue;
        }
        const uint32_t node = hdr & 0xFFFFFu, thr = (hdr >> 20) & 0xFFu, lcode = (hdr >> 28) & 3u;
        const uint32_t own = (uint32_t)words32[node];
        const int32_t* blk = data + p0 + 64 + 1024 + 4 * lane;      // this lane's slab of block 2
        uint32_t flip;
        if (rounds == 8) flip = sweep32_group_flips<1, 1, 4, WT>(blk, rounds, nb0, nb1, own, thr, lcode);
        else if (rounds == 16) flip = sweep32_group_flips<2, 2, 5, WT>(blk, rounds, nb0, nb1, own, thr, lcode);
        else if (rounds <= 24) flip = sweep32_group_flips<0, 2, 5, WT>(blk, rounds, nb0, nb1, own, thr, lcode);
        else flip = sweep32_group_flips<0, 4, 7, WT>(blk, rounds, nb0, nb1, own, thr, lcode);
        if (node < (uint32_t)N && (lane & ((1 << lcode) - 1)) == 0) words32[node] = (WT)(own ^ flip);
        prefetch(k + W);
    }
    for (; passed < num_levels; ++passed) __syncthreads();
    __syncthreads();
}

}  // namespace rls
