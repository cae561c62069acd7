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
#include <cstdlib>

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
// (HIP's __ballot takes an int: bool -> v_cndmask 0 / 1 -> v_cmp_ne, two VALU instructions per ballot on top of the compare
// that already wrote the lane mask -- the builtin takes the i1 itself)
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// ---- 64x64 bit-matrix transpose across a wavefront.
// Lane l holds row l as (r1:r0); afterwards lane p holds column p (bit l = old bit p of lane l).
// Six index-bit exchanges: lanes l and l^s trade the bits whose position differs in bit s.
//   s = 32 : one v_permlane32_swap (gfx950)
//   s = 16, 8 : partner's dword through ds_swizzle (LDS crossbar, no memory) + one v_perm_b32
//   s = 4, 2, 1 : ds_swizzle + v_alignbit (rotate) + one v_bitop3 (bit select)
// 17 VALU + 10 swizzles per 4096 bits; the ballot-per-node form it replaces cost ~5 VALU per 64 bits.
struct BitXpose {
    uint32_t sel16, sel8, m4, m2, m1;
    int r4, r2, r1;
    int node;   // node offset (within a 64-node block) of the column this lane ends up with, see pack_bits
};

__device__ __forceinline__ BitXpose bit_xpose_consts(int lane) {
    BitXpose c;
    c.sel16 = (lane & 16) ? 0x03020706u : 0x05040100u;
    c.sel8 = (lane & 8) ? 0x03070105u : 0x06020400u;
    c.m4 = (lane & 4) ? 0xF0F0F0F0u : 0x0F0F0F0Fu;
    c.m2 = (lane & 2) ? 0xCCCCCCCCu : 0x33333333u;
    c.m1 = (lane & 1) ? 0xAAAAAAAAu : 0x55555555u;
    c.r4 = (lane & 4) ? 4 : 28;
    c.r2 = (lane & 2) ? 2 : 30;
    c.r1 = (lane & 1) ? 1 : 31;
    c.node = (lane & 32) | ((lane & 7) << 2) | ((lane >> 3) & 3);
    return c;
}

template <int XOR> __device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x1F | (XOR << 10));
}

__device__ __forceinline__ uint32_t xpose_sub(uint32_t x, uint32_t other, uint32_t keep, int rot) {
    const uint32_t t = __builtin_amdgcn_alignbit(other, other, rot);
    return (keep & x) | (~keep & t);
}

__device__ __forceinline__ void bit_transpose64(uint32_t& r0, uint32_t& r1, const BitXpose& c) {
    auto sw = __builtin_amdgcn_permlane32_swap(r0, r1, false, false);
    r0 = sw[0];
    r1 = sw[1];
    uint32_t o0 = lane_xor<16>(r0), o1 = lane_xor<16>(r1);
    r0 = __builtin_amdgcn_perm(o0, r0, c.sel16);
    r1 = __builtin_amdgcn_perm(o1, r1, c.sel16);
    o0 = lane_xor<8>(r0); o1 = lane_xor<8>(r1);
    r0 = __builtin_amdgcn_perm(o0, r0, c.sel8);
    r1 = __builtin_amdgcn_perm(o1, r1, c.sel8);
    o0 = lane_xor<4>(r0); o1 = lane_xor<4>(r1);
    r0 = xpose_sub(r0, o0, c.m4, c.r4);
    r1 = xpose_sub(r1, o1, c.m4, c.r4);
    o0 = lane_xor<2>(r0); o1 = lane_xor<2>(r1);
    r0 = xpose_sub(r0, o0, c.m2, c.r2);
    r1 = xpose_sub(r1, o1, c.m2, c.r2);
    o0 = lane_xor<1>(r0); o1 = lane_xor<1>(r1);
    r0 = xpose_sub(r0, o0, c.m1, c.r1);
    r1 = xpose_sub(r1, o1, c.m1, c.r1);
}

// 32 spin bytes (0|1) of one env -> 32 bits: bit 8*b + q = byte b of dword q, i.e. node 4*q + b.
__device__ __forceinline__ uint32_t pack_bits(const u32x4& lo, const u32x4& hi) {
    uint32_t r = lo[0];
    r |= lo[1] << 1; r |= lo[2] << 2; r |= lo[3] << 3;
    r |= hi[0] << 4; r |= hi[1] << 5; r |= hi[2] << 6; r |= hi[3] << 7;
    return r;
}
__device__ __forceinline__ void unpack_bits(uint32_t r, u32x4& lo, u32x4& hi) {
    lo = u32x4{r & 0x01010101u, (r >> 1) & 0x01010101u, (r >> 2) & 0x01010101u, (r >> 3) & 0x01010101u};
    hi = u32x4{(r >> 4) & 0x01010101u, (r >> 5) & 0x01010101u, (r >> 6) & 0x01010101u, (r >> 7) & 0x01010101u};
}

// direct global -> LDS load, 16 B per lane: lane l's bytes land at lds_wave_base + 16 * l
template <bool NT = false>   // NT: nontemporal (aux = 2) for bytes that are read exactly once
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, NT ? 2 : 0);
}

// long rows take 128-byte row pieces in the staged tile loader (measured: G70-sized rows, N = 10^4: K1 314 -> 285 us;
// G22-sized rows, N = 2000: no difference, and the shorter chunk list balances better over the waves)
__device__ __forceinline__ bool tile_row128(int64_t N) { return N >= 4096; }

// ---- row-piece staging (the env-major <-> lane-per-env corner turn).
// A lane-per-env global access touches 64 different cache lines per instruction and the texture path
// spends ~3 clocks per line: the tile load ran at 2.5 TB/s however many waves were resident.  So the
// global side moves contiguous 64-byte row pieces instead (instruction i: rows 16i..16i+15, lane l ->
// row 16i + (l & 15), 16-byte piece l >> 4), and the wave turns the corner through a 4 KB LDS stage:
// each lane writes what it fetched (slot = its lane id in the instruction's 1 KB region) and reads back
// the 4 pieces of ITS env's row.  Both sides are conflict-free: the 16 lanes of a ds_*_b128 pass touch 16
// different bank quads.  Loads land in registers, kStageDepth chunks (4 KB each) per wave in flight --
// an LDS-DMA version needed one stage per chunk in flight and LDS capped the bytes in flight per CU.
constexpr int kStagePieces = 4;                      // 16-byte pieces per row per chunk
constexpr int kStageRows = 64 / kStagePieces;        // rows per load instruction
constexpr int kStageBytes = 1024 * kStagePieces;     // per wave
constexpr int kStageNodes = 16 * kStagePieces;       // nodes per chunk (= one 64x64 transpose block)
constexpr int kStageDepth = 3;                       // chunks in flight per wave

// Host side: append W stages to a kernel's dynamic LDS when they fit; returns their byte offset or -1.
inline int tile_stage_offset(size_t* lds_bytes, int W, bool wanted) {
    const bool off = knob_on(KN_TILE_NOSTAGE);   // dev knob: lane-per-env global access
    const size_t base = (*lds_bytes + 15) & ~(size_t)15;
    if (!wanted || off || base + (size_t)W * kStageBytes > (size_t)kLdsBytes) return -1;
    *lds_bytes = base + (size_t)W * kStageBytes;
    return (int)base;
}

// what lane l of load/store instruction i moves: row 16i + r, piece j
__device__ __forceinline__ void stage_io_lane(int l, int& r, int& j) { r = l & 15; j = l >> 4; }
__device__ __forceinline__ int stage_slot_off(int env, int piece) {   // byte offset of (env's row, piece j)
    return ((env >> 4) << 10) + (((piece << 4) + (env & 15)) << 4);
}

// ---- staged tile I/O, generic over the piece size PB = 16, 8 or 4 bytes (rows need only be PB-aligned:
// N % 16 == 0 -> 16, N % 8 == 0 -> 8 (Gset's 1000 / 3000 / 5000 / 7000-node graphs), N % 4 == 0 -> 4).
// A chunk = 64 nodes = 64 bytes of every row = PP = 64 / PB pieces per row; instruction i of PP moves the
// rows RPI*i .. RPI*i + RPI-1 (RPI = 64 / PP): lane l -> row RPI*i + (l % RPI), piece l / RPI, i.e. 64-byte runs
// on the global side whatever PB.  Stage layout: slot (piece j, row) at byte (64 j + row) * PB, so the read
// back by lane = row is contiguous across lanes (conflict-free); the writes of PB < 16 see 4- / 16-way
// bank conflicts, which is noise next to the memory time.
template <int PB> struct PieceVec;
template <> struct PieceVec<16> { using type = u32x4; };
template <> struct PieceVec<8> { typedef uint32_t type __attribute__((ext_vector_type(2))); };
template <> struct PieceVec<4> { using type = uint32_t; };

template <int PB>
__device__ __forceinline__ void piece_to_dwords(const typename PieceVec<PB>::type& v, uint32_t* d) {
    if constexpr (PB == 16) { d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
    else if constexpr (PB == 8) { d[0] = v[0]; d[1] = v[1]; }
    else d[0] = v;
}
template <int PB>
__device__ __forceinline__ typename PieceVec<PB>::type dwords_to_piece(const uint32_t* d) {
    if constexpr (PB == 16) return u32x4{d[0], d[1], d[2], d[3]};
    else if constexpr (PB == 8) return typename PieceVec<8>::type{d[0], d[1]};
    else return d[0];
}

// XORW: XOR the loaded bits into words[] instead of overwriting (x ^ mask without a second tile; every word is
// produced by the same lane of the same wave in both passes, so no barrier is needed in between)
template <bool XORW> __device__ __forceinline__ void put_word(uint64_t* words, int64_t n, uint64_t v) {
    if constexpr (XORW) words[n] ^= v;
    else words[n] = v;
}

template <int PB, int DEPTH, bool XORW>
__device__ __forceinline__ void tile_load_bits_staged(const uint8_t* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                                      uint64_t* __restrict__ words, int lane, int w, int W,
                                                      unsigned char* stage) {
    using PV = typename PieceVec<PB>::type;
    constexpr int PP = 64 / PB, RPI = 64 / PP, DW = PB / 4;
    const int64_t nchunk = (N + 63) >> 6;
    const BitXpose xc = bit_xpose_consts(lane);
    const int r = lane % RPI, j = lane / RPI;
    for (int64_t ch0 = w; ch0 < nchunk; ch0 += (int64_t)W * DEPTH) {
        PV g[DEPTH][PP];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int64_t off = ((ch0 + (int64_t)d * W) << 6) + j * PB;   // byte (= node) offset of this lane's piece
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int64_t rw = b0 + RPI * i + r;
                PV z{};
                g[d][i] = (rw < B && off < N) ? *reinterpret_cast<const PV*>(x + rw * N + off) : z;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int64_t ch = ch0 + (int64_t)d * W;
            if (ch < nchunk) {
#pragma unroll
                for (int i = 0; i < PP; ++i)
                    *reinterpret_cast<PV*>(stage + (j * 64 + RPI * i + r) * PB) = g[d][i];
                asm volatile("" ::: "memory");   // LDS ops of one wave execute in order
                uint32_t dw[16];
#pragma unroll
                for (int q = 0; q < PP; ++q) {
                    const PV v = *reinterpret_cast<const PV*>(stage + (q * 64 + lane) * PB);
                    piece_to_dwords<PB>(v, dw + q * DW);
                }
                asm volatile("" ::: "memory");
                uint32_t r0 = pack_bits(u32x4{dw[0], dw[1], dw[2], dw[3]}, u32x4{dw[4], dw[5], dw[6], dw[7]});
                uint32_t r1 = pack_bits(u32x4{dw[8], dw[9], dw[10], dw[11]}, u32x4{dw[12], dw[13], dw[14], dw[15]});
                bit_transpose64(r0, r1, xc);
                const int64_t n = (ch << 6) + xc.node;
                if (n < N) put_word<XORW>(words, n, ((uint64_t)r1 << 32) | r0);
            }
        }
    }
}

// The same corner turn with 128-BYTE row pieces per load instruction (lane l -> row 8 i + (l & 7), 16-byte piece l >> 3):
// an instruction touches 8 WHOLE 128-byte cache lines instead of 16 half lines, so the texture path handles half as
// many lines for the same bytes.  A chunk is 64 rows x 128 nodes = two 64 x 64 transpose blocks, turned through the
// same 4 KB stage one block at a time.
template <int DEPTH, bool XORW>
__device__ __forceinline__ void tile_load_bits_staged128(const uint8_t* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                                         uint64_t* __restrict__ words, int lane, int w, int W,
                                                         unsigned char* stage) {
    using PV = u32x4;
    const int64_t nchunk = (N + 127) >> 7;                         // 128-node chunks
    const BitXpose xc = bit_xpose_consts(lane);
    const int r = lane & 7, j = lane >> 3;                         // row within the instruction's 8, piece 0..7 of the row's 128 B
    for (int64_t ch0 = w; ch0 < nchunk; ch0 += (int64_t)W * DEPTH) {
        PV g[DEPTH][8];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int64_t off = ((ch0 + (int64_t)d * W) << 7) + j * 16;   // byte (= node) offset of this lane's piece
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t rw = b0 + 8 * i + r;
                PV z{};
                g[d][i] = (rw < B && off < N) ? *reinterpret_cast<const PV*>(x + rw * N + off) : z;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int64_t ch = ch0 + (int64_t)d * W;
            if (ch < nchunk) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {                     // block h = nodes 128 ch + 64 h .. + 63 = pieces 4h .. 4h + 3
                    // lanes whose piece belongs to this block park it: slot (piece-in-block, row)
                    if ((j >> 2) == h) {
#pragma unroll
                        for (int i = 0; i < 8; ++i)
                            *reinterpret_cast<PV*>(stage + ((j & 3) * 64 + 8 * i + r) * 16) = g[d][i];
                    }
                    asm volatile("" ::: "memory");               // LDS ops of one wave execute in order
                    uint32_t dw[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const PV v = *reinterpret_cast<const PV*>(stage + (q * 64 + lane) * 16);
                        dw[q * 4 + 0] = v[0]; dw[q * 4 + 1] = v[1]; dw[q * 4 + 2] = v[2]; dw[q * 4 + 3] = v[3];
                    }
                    asm volatile("" ::: "memory");
                    uint32_t r0 = pack_bits(u32x4{dw[0], dw[1], dw[2], dw[3]}, u32x4{dw[4], dw[5], dw[6], dw[7]});
                    uint32_t r1 = pack_bits(u32x4{dw[8], dw[9], dw[10], dw[11]}, u32x4{dw[12], dw[13], dw[14], dw[15]});
                    bit_transpose64(r0, r1, xc);
                    const int64_t n = (ch << 7) + 64 * h + xc.node;
                    if (n < N) put_word<XORW>(words, n, ((uint64_t)r1 << 32) | r0);
                }
            }
        }
    }
}

template <int PB>
__device__ __forceinline__ void tile_store_bytes_staged(uint8_t* __restrict__ x, int64_t N, int64_t b0,
                                                        const uint64_t* __restrict__ words, int lane, int w, int W,
                                                        uint64_t rows_ok, unsigned char* stage) {
    using PV = typename PieceVec<PB>::type;
    constexpr int PP = 64 / PB, RPI = 64 / PP, DW = PB / 4;
    const int64_t nchunk = (N + 63) >> 6;
    const BitXpose xc = bit_xpose_consts(lane);
    const int r = lane % RPI, j = lane / RPI;
    for (int64_t ch = w; ch < nchunk; ch += W) {
        const int64_t n = (ch << 6) + xc.node;
        const uint64_t wd = (n < N) ? words[n] : 0ull;
        uint32_t r0 = (uint32_t)wd, r1 = (uint32_t)(wd >> 32);
        bit_transpose64(r0, r1, xc);
        u32x4 v[4];
        unpack_bits(r0, v[0], v[1]);
        unpack_bits(r1, v[2], v[3]);
        const uint32_t dw[16] = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3],
                                 v[2][0], v[2][1], v[2][2], v[2][3], v[3][0], v[3][1], v[3][2], v[3][3]};
#pragma unroll
        for (int q = 0; q < PP; ++q)
            *reinterpret_cast<PV*>(stage + (q * 64 + lane) * PB) = dwords_to_piece<PB>(dw + q * DW);
        asm volatile("" ::: "memory");   // LDS ops of one wave execute in order
        const int64_t off = (ch << 6) + j * PB;
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int rr = RPI * i + r;
            const PV o = *reinterpret_cast<const PV*>(stage + (j * 64 + rr) * PB);
            if (((rows_ok >> rr) & 1ull) && off < N) *reinterpret_cast<PV*>(x + (b0 + rr) * N + off) = o;
        }
        asm volatile("" ::: "memory");
    }
}

// ---- the same corner turn for rows that are NOT 4-byte aligned (N % 4 != 0, or a base pointer that is not 16-byte aligned):
// 4-byte pieces at row offsets 4 j, each fetched as the one or two ALIGNED dwords that hold it and funnel-shifted
// (v_alignbyte); stores go out as bytes.  Twice the load instructions of the 4-byte form and four times its stores, against 64
// byte loads and 64 ballots per lane and chunk without it (K1 on N = 1999 / 2^16 envs: 482 us, 35 at N = 2000).
// `last` = address of the last byte of the whole [B, N] array: the second dword of a piece is only read when it holds bytes
// of the array (the first one always does).
template <int DEPTH, bool XORW>
__device__ __forceinline__ void tile_load_bits_staged_unal(const uint8_t* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                                           uint64_t* __restrict__ words, int lane, int w, int W,
                                                           unsigned char* stage) {
    constexpr int PB = 4, PP = 16, RPI = 4;
    const int64_t nchunk = (N + 63) >> 6;
    const BitXpose xc = bit_xpose_consts(lane);
    const int r = lane % RPI, j = lane / RPI;
    const uintptr_t last = (uintptr_t)(x + B * N - 1);
    for (int64_t ch0 = w; ch0 < nchunk; ch0 += (int64_t)W * DEPTH) {
        uint32_t g[DEPTH][PP];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int64_t off = ((ch0 + (int64_t)d * W) << 6) + j * PB;
#pragma unroll
            for (int i = 0; i < PP; ++i) {
                const int64_t rw = b0 + RPI * i + r;
                uint32_t v = 0;
                if (rw < B && off < N) {
                    const uintptr_t addr = (uintptr_t)(x + rw * N + off), a = addr & ~(uintptr_t)3;
                    const uint32_t lo = *reinterpret_cast<const uint32_t*>(a);
                    const uint32_t hi = (a + 4 <= last) ? *reinterpret_cast<const uint32_t*>(a + 4) : 0u;
                    v = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(addr & 3));
                    // bytes past the end of the row are the next row's -- or, in the last row, whatever follows the array:
                    // pack_bits takes 0 | 1 bytes only, anything else would spill into the bits of the row's own nodes
                    if (N - off < 4) v &= (1u << (8 * (int)(N - off))) - 1u;
                }
                g[d][i] = v;
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int64_t ch = ch0 + (int64_t)d * W;
            if (ch < nchunk) {
#pragma unroll
                for (int i = 0; i < PP; ++i) *reinterpret_cast<uint32_t*>(stage + (j * 64 + RPI * i + r) * PB) = g[d][i];
                asm volatile("" ::: "memory");   // LDS ops of one wave execute in order
                uint32_t dw[16];
#pragma unroll
                for (int q = 0; q < PP; ++q) dw[q] = *reinterpret_cast<const uint32_t*>(stage + (q * 64 + lane) * PB);
                asm volatile("" ::: "memory");
                uint32_t r0 = pack_bits(u32x4{dw[0], dw[1], dw[2], dw[3]}, u32x4{dw[4], dw[5], dw[6], dw[7]});
                uint32_t r1 = pack_bits(u32x4{dw[8], dw[9], dw[10], dw[11]}, u32x4{dw[12], dw[13], dw[14], dw[15]});
                bit_transpose64(r0, r1, xc);
                const int64_t n = (ch << 6) + xc.node;
                if (n < N) put_word<XORW>(words, n, ((uint64_t)r1 << 32) | r0);
            }
        }
    }
}

__device__ __forceinline__ void tile_store_bytes_staged_unal(uint8_t* __restrict__ x, int64_t N, int64_t b0,
                                                             const uint64_t* __restrict__ words, int lane, int w, int W,
                                                             uint64_t rows_ok, unsigned char* stage) {
    constexpr int PB = 4, PP = 16, RPI = 4;
    const int64_t nchunk = (N + 63) >> 6;
    const BitXpose xc = bit_xpose_consts(lane);
    const int r = lane % RPI, j = lane / RPI;
    for (int64_t ch = w; ch < nchunk; ch += W) {
        const int64_t n = (ch << 6) + xc.node;
        const uint64_t wd = (n < N) ? words[n] : 0ull;
        uint32_t r0 = (uint32_t)wd, r1 = (uint32_t)(wd >> 32);
        bit_transpose64(r0, r1, xc);
        u32x4 v[4];
        unpack_bits(r0, v[0], v[1]);
        unpack_bits(r1, v[2], v[3]);
        const uint32_t dw[16] = {v[0][0], v[0][1], v[0][2], v[0][3], v[1][0], v[1][1], v[1][2], v[1][3],
                                 v[2][0], v[2][1], v[2][2], v[2][3], v[3][0], v[3][1], v[3][2], v[3][3]};
#pragma unroll
        for (int q = 0; q < PP; ++q) *reinterpret_cast<uint32_t*>(stage + (q * 64 + lane) * PB) = dw[q];
        asm volatile("" ::: "memory");   // LDS ops of one wave execute in order
        const int64_t off = (ch << 6) + j * PB;
#pragma unroll
        for (int i = 0; i < PP; ++i) {
            const int rr = RPI * i + r;
            const uint32_t o = *reinterpret_cast<const uint32_t*>(stage + (j * 64 + rr) * PB);
            if ((rows_ok >> rr) & 1ull) {
                uint8_t* dst = x + (b0 + rr) * N + off;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (off + k < N) dst[k] = (uint8_t)(o >> (8 * k));
            }
        }
        asm volatile("" ::: "memory");
    }
}

// Load the tile of envs [b0, b0+64) x nodes [0, N) into words[0..N).
// VEC = true requires x 16-byte aligned and rows 4-byte aligned (tile_rows_aligned); without a stage, byte spins
// with N % 16 != 0 fall back to the element-wise path.  VEC = false with a stage (byte spins): any alignment, through the
// funnel-shift form of the stage.
// W waves of one workgroup may share the job (wave w of W takes every W-th batch of columns); every
// wave sees all 64 envs, so each ballot still yields a complete word.  Callers sync afterwards.
template <typename T, bool VEC, int DEPTH = kStageDepth, bool XORW = false>
__device__ __forceinline__ void tile_load_bits(const T* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                               uint64_t* __restrict__ words, int lane, int w = 0, int W = 1,
                                               unsigned char* stage = nullptr) {
    const int64_t b = b0 + lane;
    const bool valid = b < B;
    const T* row = x + (valid ? b : 0) * N;
    if constexpr (VEC && sizeof(T) == 1) {
        if (stage != nullptr) {   // `stage`: this wave's kStageBytes of LDS, 16-byte aligned; rows 4-byte aligned
            const uint8_t* xb = reinterpret_cast<const uint8_t*>(x);
            if ((N & 15) == 0) {
                if (tile_row128(N)) tile_load_bits_staged128<(DEPTH > 2 ? 2 : DEPTH), XORW>(xb, B, N, b0, words, lane, w, W, stage);
                else tile_load_bits_staged<16, DEPTH, XORW>(xb, B, N, b0, words, lane, w, W, stage);
            }
            else if ((N & 7) == 0) tile_load_bits_staged<8, DEPTH, XORW>(xb, B, N, b0, words, lane, w, W, stage);
            else tile_load_bits_staged<4, DEPTH, XORW>(xb, B, N, b0, words, lane, w, W, stage);
            return;
        }
        if ((N & 15) == 0) {
        // 64 nodes per block: every lane packs 64 bytes of its env's row into 64 bits, the wave transposes
        // the 64x64 bit matrix, lane p then holds the word of node n0 + node(p).  Bytes must be 0|1.
        const u32x4* rv = reinterpret_cast<const u32x4*>(row);
        const int64_t nv = N >> 4;
        const int64_t nblk = (N + 63) >> 6;
        const BitXpose xc = bit_xpose_consts(lane);
        constexpr int BLKS = 2;  // blocks (8 row loads) in flight per lane
        for (int64_t blk0 = (int64_t)w * BLKS; blk0 < nblk; blk0 += (int64_t)W * BLKS) {
            u32x4 v[BLKS][4];
#pragma unroll
            for (int q = 0; q < BLKS; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t idx = ((blk0 + q) << 2) + j;
                    v[q][j] = (valid && idx < nv) ? rv[idx] : u32x4{0, 0, 0, 0};
                }
#pragma unroll
            for (int q = 0; q < BLKS; ++q) {
                uint32_t r0 = pack_bits(v[q][0], v[q][1]), r1 = pack_bits(v[q][2], v[q][3]);
                bit_transpose64(r0, r1, xc);
                const int64_t n = ((blk0 + q) << 6) + xc.node;
                if (n < N) put_word<XORW>(words, n, ((uint64_t)r1 << 32) | r0);
            }
        }
            return;
        }
        // no stage and rows not 16-byte aligned: the element-wise path below
    }
    if constexpr (VEC && sizeof(T) == 4) {
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
            if (lane < 4) put_word<XORW>(words, (i << 2) + lane, mine);
        }
        return;
    }
    if constexpr (!VEC && sizeof(T) == 1) {
        if (stage != nullptr) {   // rows (or the base) not 4-byte aligned: the funnel-shift form of the row-piece stage
            tile_load_bits_staged_unal<(DEPTH > 2 ? 2 : DEPTH), XORW>(reinterpret_cast<const uint8_t*>(x), B, N, b0, words, lane, w, W, stage);
            return;
        }
    }
    for (int64_t n0 = (int64_t)w * 64; n0 < N; n0 += (int64_t)W * 64) {
        uint64_t mine = 0;
        const int lim = (int)((N - n0) < 64 ? (N - n0) : 64);
        for (int k = 0; k < lim; ++k) {
            T v = valid ? row[n0 + k] : T(0);
            uint64_t w = ballot64(spin_is_set(v));
            if (lane == k) mine = w;
        }
        if (lane < lim) put_word<XORW>(words, n0 + lane, mine);
    }
}

// Write the tile back as env-major bytes (uint8 0|1); lanes with store_row == false leave their row untouched.
template <bool VEC>
__device__ __forceinline__ void tile_store_bytes(uint8_t* __restrict__ x, int64_t B, int64_t N, int64_t b0,
                                                 const uint64_t* __restrict__ words, int lane, int w = 0, int W = 1,
                                                 bool store_row = true, unsigned char* stage = nullptr) {
    const int64_t b = b0 + lane;
    const bool valid = b < B && store_row;   // all 64 lanes take part in the transpose whatever they store
    uint8_t* row = x + (valid ? b : 0) * N;
    const int half = lane >> 5, sh = lane & 31;
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(words);
    if constexpr (VEC) {
        if (stage != nullptr) {   // through the row-piece stage: 64-byte runs on the global side
            const uint64_t rows_ok = ballot64(valid);
            if ((N & 15) == 0) tile_store_bytes_staged<16>(x, N, b0, words, lane, w, W, rows_ok, stage);
            else if ((N & 7) == 0) tile_store_bytes_staged<8>(x, N, b0, words, lane, w, W, rows_ok, stage);
            else tile_store_bytes_staged<4>(x, N, b0, words, lane, w, W, rows_ok, stage);
            return;
        }
        if ((N & 15) == 0) {
        // inverse of the load: lane p fetches the word of node n0 + node(p), transpose, unpack to bytes
        u32x4* rv = reinterpret_cast<u32x4*>(row);
        const int64_t nv = N >> 4;
        const int64_t nblk = (N + 63) >> 6;
        const BitXpose xc = bit_xpose_consts(lane);
        for (int64_t blk = w; blk < nblk; blk += W) {
            const int64_t n = (blk << 6) + xc.node;
            const uint64_t wd = (n < N) ? words[n] : 0ull;
            uint32_t r0 = (uint32_t)wd, r1 = (uint32_t)(wd >> 32);
            bit_transpose64(r0, r1, xc);
            u32x4 v[4];
            unpack_bits(r0, v[0], v[1]);
            unpack_bits(r1, v[2], v[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t idx = (blk << 2) + j;
                if (valid && idx < nv) rv[idx] = v[j];
            }
        }
            return;
        }
    }
    if constexpr (!VEC) {
        if (stage != nullptr) {
            tile_store_bytes_staged_unal(x, N, b0, words, lane, w, W, ballot64(valid), stage);
            return;
        }
    }
    if (!valid) return;
    for (int64_t n = w; n < N; n += W) row[n] = (uint8_t)((w32[(n << 1) + half] >> sh) & 1u);
}

// ---- bit-sliced (vertical) counters: plane p holds bit p of 64 independent counts.
// carry-save adder on 64 counters at once; gfx950's v_bitop3 does a 3-input majority / parity in one op
__device__ __forceinline__ void csa(uint64_t& hi, uint64_t& lo, uint64_t a, uint64_t b, uint64_t c) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32),
                   c0 = (uint32_t)c, c1 = (uint32_t)(c >> 32);
    const uint32_t h0 = __builtin_amdgcn_bitop3_b32(a0, b0, c0, 0xE8), h1 = __builtin_amdgcn_bitop3_b32(a1, b1, c1, 0xE8);
    const uint32_t l0 = __builtin_amdgcn_bitop3_b32(a0, b0, c0, 0x96), l1 = __builtin_amdgcn_bitop3_b32(a1, b1, c1, 0x96);
    hi = ((uint64_t)h1 << 32) | h0;
    lo = ((uint64_t)l1 << 32) | l0;
}

// lane exchange of a 64-bit plane with lane ^ 1, ^ 2 (DPP quad_perm 0xB1 / 0x4E) or ^ 4 (ds_swizzle bit mode), no memory
template <int X>
__device__ __forceinline__ uint64_t lv_lane_xor(uint64_t v) {
    uint32_t lo, hi;
    if constexpr (X == 4) {
        lo = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)v, 0x101F);          // and 0x1F, or 0, xor 4
        hi = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(uint32_t)(v >> 32), 0x101F);
    } else {
        constexpr int CTRL = X == 1 ? 0xB1 : 0x4E;
        lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, false);
        hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, false);
    }
    return ((uint64_t)hi << 32) | lo;
}

// pl[0..7] += lane ^ X's pl[0..7] where `take` is all-ones (bit-sliced ripple-carry add, counts <= 128 in total)
template <int X>
__device__ __forceinline__ void lv_merge_planes(uint64_t (&pl)[8], uint64_t take) {
    uint64_t carry = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) csa(carry, pl[p], pl[p], lv_lane_xor<X>(pl[p]) & take, carry);
}

// Bit-sliced compare of 64 vertical counters (planes pl[0 .. NP), plane p = bit p of every count) with a per-LANE constant
// thr < 2^NP:   c <= thr  <=>  c + (2^NP - 1 - thr) < 2^NP  <=>  adding ~thr (NP bits) to c carries nothing out of plane
// NP - 1.  Only the carry chain is needed -- one 3-input majority (v_bitop3) per plane and half, one v_bfe_i32 per plane for
// the constant's bit -- where the scan from the top plane down ("smaller so far / equal so far") took four operations per
// plane and half.  Returns the mask of counters that are <= thr.
template <int NP, int CAP>
__device__ __forceinline__ uint64_t lv_le_const(const uint64_t (&pl)[CAP], uint32_t thr) {
    static_assert(NP <= CAP, "planes");
    const int nt = (int)~thr;
    uint32_t c0 = 0, c1 = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t kb = (uint32_t)__builtin_amdgcn_sbfe(nt, p, 1);      // all ones where thr has bit p CLEAR
        c0 = __builtin_amdgcn_bitop3_b32((uint32_t)pl[p], kb, c0, 0xE8);
        c1 = __builtin_amdgcn_bitop3_b32((uint32_t)(pl[p] >> 32), kb, c1, 0xE8);
    }
    return ~(((uint64_t)c1 << 32) | c0);
}

// The same with one more plane below: mask of [2 c + low <= thr2], `low` a 0 | 1 bit per counter (a 64-bit word), thr2 < 2^(NP+1).
template <int NP, int CAP>
__device__ __forceinline__ uint64_t lv_le_const_x2(const uint64_t (&pl)[CAP], uint64_t low, uint32_t thr2) {
    static_assert(NP <= CAP, "planes");
    const int nt = (int)~thr2;
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_sbfe(nt, 0, 1);
    uint32_t c0 = (uint32_t)low & k0, c1 = (uint32_t)(low >> 32) & k0;       // majority with carry-in 0
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t kb = (uint32_t)__builtin_amdgcn_sbfe(nt, p + 1, 1);
        c0 = __builtin_amdgcn_bitop3_b32((uint32_t)pl[p], kb, c0, 0xE8);
        c1 = __builtin_amdgcn_bitop3_b32((uint32_t)(pl[p] >> 32), kb, c1, 0xE8);
    }
    return ~(((uint64_t)c1 << 32) | c0);
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
