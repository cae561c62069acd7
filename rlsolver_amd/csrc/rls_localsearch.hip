// Fused local search: EnvMaxcut.local_search_inplace (envs/env_L2A.py:87-116) and
// LocalSearch.random_search (methods/LocalSearch.py:53-86) as ONE kernel per call.
//
// The decomposed path (K2 -> torch weights/noise/kthvalue -> 8 x K6 -> K5) re-transposes the 64-env
// tile ten times and spends 85 % of its time in torch ops on [B, N] int64/f32 tensors (17.7 ms at
// B = 2^16 on G22).  Here the tile is transposed once and stays in LDS:
//
//   phase 0  4 waves load the bit tile, the CSR rowptr, (optionally) count the initial cut
//   phase 1  thresh[b] = kthvalue(ws + noise_0 * rd_std, k = N - num_spin): every lane keeps the
//            num_spin+1 largest values of its env in registers (max/min insertion network), the four
//            waves' lists are merged through LDS
//   phase 2  num_iters proposal rounds: mask bit = (ws + noise_t * rd_std) > thresh, one ballot per
//            node gives the mask word directly (no byte->bit transpose), proposal = words ^ mask,
//            cut by the bit-sliced counter (4 waves), accepted envs merged with one AND/XOR per word
//   phase 3  wave 0 runs the greedy sweep (rls_sweep.h) on the resident tile
//   phase 4  4 waves write the tile back
//
// ws = int8 / int16 [B, N] (the reference's  n0_num_n1 - k * cutdeg, an exact integer in both flavours, |ws| <= the
// largest degree: one byte per entry up to degree 127) and rd_std f32 [N] come from a pre-pass because rd_std is a
// statistic over the WHOLE batch (max - min over dim 0, env_L2A.py:93-94).  ws is re-read by the threshold pass and
// by every proposal round -- (num_iters + 1) B N entries: as int32 that was 4.7 GB of HBM reads for G22 / 2^16 envs
// (6.3x the algorithmic bytes, half the kernel's time); as int8 it is 1.2 GB.  noise = the randn_like draws (test mode, bit-exact
// against the reference) or NULL for an in-kernel counter-based hash + Box-Muller keyed by (seed, global env,
// node, round).  f32 arithmetic follows torch: (float)ws + (noise * rd_std), two roundings.
#include "rls_cutcount.h"
#include "rls_sweep.h"
#include "rls_tile32.h"
#include <cstdlib>
#include <type_traits>

namespace rls {

constexpr int kLsMergeWaves = 4;   // waves that own the top-k merge and the batched sweep; a tile runs on 4 or 8 waves
constexpr int kTopCap = 16;  // num_spin + 1 <= kTopCap

// insert v into the descending list t[0 .. DEPTH) (entries from DEPTH on are untouched: with DEPTH = num_spin + 1 = the
// rank the threshold is read at, nothing below it can matter)
// One v_med3 per level: with t[j-1] >= t[j], the new t[j] is the median of (t[j-1], t[j], v) -- v between them: v; v above both:
// the old t[j-1] moves down; v below both: t[j] stays -- and every level reads only OLD entries (bottom-up in place), so the levels
// are independent instructions.  (Until round 5: a chain of max / min pairs, two dependent instructions per level -- 18 of the
// threshold pass's ~30 instructions per (env, node).)
template <int DEPTH>
__device__ __forceinline__ void top_insert_n(float (&t)[kTopCap], float v) {
#pragma unroll
    for (int j = DEPTH - 1; j >= 1; --j) t[j] = __builtin_amdgcn_fmed3f(t[j - 1], t[j], v);
    t[0] = fmaxf(t[0], v);
}
__device__ __forceinline__ void top_insert(float (&t)[kTopCap], float v) { top_insert_n<kTopCap>(t, v); }

// The mask words of a piece of NPC nodes: lane k < NPC ends with the 64-env word of node k, bit e = (v[k] > th) in env e.
// One v_cmp per node writes the word into a scalar pair; two v_writelane put it into lane k.  (Until round 5 this was
// ballot64(valid && v > th) + `if (lane == k) mine = mm`: compare, cndmask + cmp_ne of HIP's int ballot, and two mov + two cndmask
// for the lane select -- 8 VALU instructions per node where 3 do, a fifth of a proposal round.)  Envs past the batch are masked
// by `valid_mask` (wave-uniform) once per piece.
template <int NPC>
__device__ __forceinline__ uint64_t ls_piece_mask_words(const float (&v)[NPC], float th, uint64_t valid_mask) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
        const uint64_t mm = __builtin_amdgcn_ballot_w64(v[k] > th) & valid_mask;
        // (the scalar operands come out of an s_and: no VALU-writes-SGPR hazard in front of the v_writelane)
        asm("v_writelane_b32 %0, %1, %2" : "+v"(lo) : "s"((uint32_t)mm), "n"(k));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(hi) : "s"((uint32_t)(mm >> 32)), "n"(k));
    }
    return ((uint64_t)hi << 32) | lo;
}

// Counter-based noise for the production path: murmur3's 32-bit finaliser over (seed, global env,
// node quad, round) -- order-independent like Philox (so results do not depend on how envs are
// sharded or how waves split the nodes) at a quarter of its cost; the statistical bar here is
// "exploration noise", the parity bar is met by the test mode with supplied draws.
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// The per-env half of a draw's key: TWO independent 32-bit mixes of (seed, global env id).  (One 32-bit key per env, as until
// round 3, makes two of a call's 2^16 envs share ALL their draws with probability ~1/2 per call -- 2^31 pairs against 2^32 keys.)
struct EnvKey { uint32_t k0, k1; };
__device__ __forceinline__ EnvKey ls_env_key(uint64_t seed, uint64_t gb) {
    const uint32_t lo = (uint32_t)gb, hi = (uint32_t)(gb >> 32);
    EnvKey k;
    k.k0 = fmix32((uint32_t)seed ^ fmix32((uint32_t)(seed >> 32) ^ fmix32(lo) ^ (hi * 0x9E3779B1u)));
    k.k1 = fmix32((uint32_t)(seed >> 32) + 0x3C6EF372u + fmix32((uint32_t)seed ^ fmix32(lo + 0x7F4A7C15u) ^ (hi * 0x85EBCA77u)));
    return k;
}

// four standard normals (Box-Muller) for nodes 4q..4q+3 in round `it`.
// Two hashes per quad, each of its OWN 32-bit key (k0 / k1 mixed with the quad and the round through different odd
// multipliers): a 32-bit hash value necessarily coincides between ~7 % of the 2.9 * 10^8 (env, quad, round) triples of a
// G22 / 2^16 call -- as 32-bit draws of an ideal generator would -- but with one shared key (round 3) a coincidence gave
// two triples all FOUR normals; with independent keys the pairs coincide independently (all four: 2^-64 per pair of triples).
// Each hash gives a 16-bit radius uniform and a 16-bit angle (the radius tail ends at sqrt(2 ln 2^16) = 4.7 sigma); hardware
// log2 / sqrt / sin / cos (v_sin / v_cos take revolutions).  This is the VALU floor of the proposal rounds: 2000 x 64 normals
// per round and tile.  tests/test_gpu_draw_statistics.py holds the generator to moments, tails, lag correlations and a
// chi-square of the probability integral transform.
__device__ __forceinline__ void normal4(EnvKey env_key, uint32_t q, uint32_t it, float (&z)[4]) {
#ifdef RLS_LS_FAKE_NOISE   // timing experiment only: what the kernel costs when a draw is one hash per quad and no transcendental
    {
        const uint32_t r = fmix32(env_key.k0 ^ (q * 0x9E3779B1u) ^ (it * 0x7FEB352Du + 0x165667B1u));
#pragma unroll
        for (int k = 0; k < 4; ++k) z[k] = (float)(int)((r >> (8 * k)) & 0xFFu) * (1.0f / 64.0f) - 2.0f;
        return;
    }
#endif
    const uint32_t ka = env_key.k0 ^ (q * 0x9E3779B1u) ^ (it * 0x7FEB352Du + 0x165667B1u);
    const uint32_t kb = env_key.k1 ^ (q * 0xC2B2AE3Du) ^ (it * 0x27D4EB2Fu + 0x85EBCA6Bu);
    const uint32_t r0 = fmix32(ka), r1 = fmix32(kb);
    // u = (m + 1) / 65536 in (0, 1] as ONE fma (exact: m 2^-16 + 2^-16 is representable; was add, convert, multiply), and the angle
    // as the float 1 + m / 65536 built from the bits (v_sin / v_cos reduce to the fraction; was and, convert, multiply): two
    // instructions per hash cheaper.  Checked on the hardware for all 65536 angles (tools/ceilings/sin_turn.hip): the radius uniform
    // and the cosine keep every bit; the sine keeps every bit but for m = 1 .. 22 (angles below 3.4e-4 of a turn), where 1 + t gives a
    // value within 7e-7 relative of the one t gave
    const float u1 = __builtin_fmaf((float)(r0 >> 16), 1.0f / 65536.0f, 1.0f / 65536.0f);   // (0, 1]
    const float u3 = __builtin_fmaf((float)(r1 >> 16), 1.0f / 65536.0f, 1.0f / 65536.0f);
    const float t2 = __builtin_bit_cast(float, ((r0 & 0xFFFFu) << 7) | 0x3F800000u);        // [1, 2) revolutions = [0, 1) + a turn
    const float t4 = __builtin_bit_cast(float, ((r1 & 0xFFFFu) << 7) | 0x3F800000u);
    // -2 ln u = -2 ln2 * log2 u
    const float ra = __builtin_amdgcn_sqrtf(-1.3862943611f * __builtin_amdgcn_logf(u1));
    const float rb = __builtin_amdgcn_sqrtf(-1.3862943611f * __builtin_amdgcn_logf(u3));
    z[0] = ra * __builtin_amdgcn_cosf(t2); z[1] = ra * __builtin_amdgcn_sinf(t2);
    z[2] = rb * __builtin_amdgcn_cosf(t4); z[3] = rb * __builtin_amdgcn_sinf(t4);
}

// =====================================================================================
// One proposal round's mask words with LANE = NODE (round 5): a lane owns 8 (int8 weights) / 4 (int16) consecutive nodes of a 512- /
// 256-node slot and walks 32 of the tile's 64 envs; env e's compare writes a lane mask, and ONE v_addc shifts that bit into the
// node's half word (envs 31 .. 0 of the half in that order, so env e lands on bit e).  A task = (slot, half of the envs); wave gw of
// GW takes every GW-th task and writes its 32-bit halves of the mask words -- G22-sized rows: 8 tasks per tile and round.
// Against the lane = env form (kept for the threshold pass, whose top-k is per env): no corner turn of ws through LDS, rd_std in
// registers for the 32 envs, the per-env key / threshold as scalars, 1 instruction per node for the mask bit where v_writelane
// needed 2 -- 14.5 issue slots per (env, node) against 16.75.  The draws are normal4's: the same values whatever the form.
//   key / thresh: the caller's lane = env registers (lane e holds env b0 + e; thresh = +inf in lanes past the batch: their envs
//   propose nothing);  noise_it: recorded draws of this round (f32 [B, N], test hook) or NULL;  emit(node, half, word32) is called by the lane that owns `node`, for node < N.
// ws rows sit `pitch` entries apart (a multiple of 16 bytes), N % 4 == 0.
// =====================================================================================
template <typename WT, bool USE_NOISE, int LB = 8, typename Emit>   // LB: bytes of a ws row per lane and env (8 | 16)
__device__ __forceinline__ void ls_round_words_lane_node(const WT* __restrict__ ws, int64_t pitch, int64_t B, int64_t N, int64_t b0, int lane,
                                                         int gw, int GW, const float* __restrict__ rd_std, EnvKey key, float thresh,
                                                         const float* __restrict__ noise_it, int it, Emit&& emit) {
    constexpr int NPL = LB / (int)sizeof(WT), QPL = NPL / 4;     // nodes / quads per lane
    constexpr int64_t SPAN = 64 * NPL;                           // nodes per slot
    constexpr int PF = 8;                                        // envs whose pieces are in flight
    typedef uint32_t u32x2 __attribute__((ext_vector_type(LB / 4)));
    const int64_t ntasks = 2 * ((N + SPAN - 1) / SPAN);
    for (int64_t t = gw; t < ntasks; t += GW) {
        const int half = (int)(t & 1);
        const int64_t node = (t >> 1) * SPAN + (int64_t)lane * NPL;
        const int64_t nodec = node < N ? node : 0;               // (lanes past the row read its start and emit nothing)
        float sd[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) sd[k] = (node + k < N) ? rd_std[node + k] : 0.0f;
        uint32_t acc[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) acc[k] = 0u;
        const uint32_t q0 = (uint32_t)(nodec >> 2);
        const int ebase = 32 * half;
        auto row_of = [&](int e) { const int64_t b = b0 + e; return b < B ? b : b0; };
        auto fetch = [&](int e) { return *reinterpret_cast<const u32x2*>(ws + row_of(e) * pitch + nodec); };
        auto one_env = [&](int e, u32x2 piece) {
            EnvKey ek;
            ek.k0 = (uint32_t)__builtin_amdgcn_readlane((int)key.k0, e);
            ek.k1 = (uint32_t)__builtin_amdgcn_readlane((int)key.k1, e);
            const float th = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, thresh), e));
#pragma unroll
            for (int sq = 0; sq < QPL; ++sq) {
                float z[4];
                if constexpr (USE_NOISE) {
                    // (a lane's second quad may lie past the row when N % 8 == 4: the last row's would be past the tensor)
                    const f32x4 zz = nodec + 4 * sq < N ? *reinterpret_cast<const f32x4*>(noise_it + row_of(e) * N + nodec + 4 * sq) : f32x4{0.f, 0.f, 0.f, 0.f};
                    z[0] = zz[0]; z[1] = zz[1]; z[2] = zz[2]; z[3] = zz[3];
                } else {
                    normal4(ek, q0 + (uint32_t)sq, (uint32_t)it, z);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int wv;
                    if constexpr (sizeof(WT) == 1) wv = (int)(int8_t)(piece[sq] >> (8 * k));
                    else wv = (int)(int16_t)(piece[2 * sq + (k >> 1)] >> (16 * (k & 1)));
                    const float v = (float)wv + z[k] * sd[4 * sq + k];          // two roundings: torch's ws + randn * rd_std
                    uint64_t m = __builtin_amdgcn_ballot_w64(v > th);           // lane mask of "this lane's node k is proposed in env e"
                    // acc = 2 acc + (this lane's bit of m): the carry-in of an add-with-carry (the carry-out lands in the same pair)
                    asm("v_addc_co_u32_e64 %0, %1, %0, %0, %1" : "+v"(acc[4 * sq + k]), "+s"(m));
                }
            }
        };
        u32x2 cur[PF], nxt[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) nxt[j] = cur[j] = fetch(ebase + 31 - j);
#pragma unroll 1
        for (int e0 = 31; e0 >= 0; e0 -= PF) {
            if (e0 - PF >= 0) {
#pragma unroll
                for (int j = 0; j < PF; ++j) nxt[j] = fetch(ebase + e0 - PF - j);
            }
#pragma unroll
            for (int j = 0; j < PF; ++j) one_env(ebase + e0 - j, cur[j]);
#pragma unroll
            for (int j = 0; j < PF; ++j) cur[j] = nxt[j];
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k)
            if (node + k < N) emit(node + k, half, acc[k]);
    }
}

// ALIGNED (the only form built since round 3): rows of x and noise start 4-byte aligned on 16-byte bases (N % 4 == 0), rows
// of ws sit `pitch` entries apart, a multiple of 16 bytes, so every 16-byte piece of a ws row lies inside the row.
// WT = int8_t | int16_t.
// Waves per SIMD: the 4-wave layout is built for THREE workgroups per CU (round 6: 168 registers, 4 of them spilled outside the
// round loop, and rd_std's LDS copy aliased onto the proposal tile -- 50 KB per workgroup at G22 size instead of 58): the rounds are
// VALU-issue bound and a third wave per SIMD fills the issue gaps of the other two; the 8-wave layout (one tile per CU) keeps two.
template <bool ALIGNED, typename WT, int P, int W>
__global__ __launch_bounds__(W * kWave) __attribute__((amdgpu_waves_per_eu(W == 4 ? 3 : 2, W == 4 ? 3 : 2))) void k_maxcut_local_search(
    uint8_t* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ eu, const int32_t* __restrict__ ev,
    int64_t E, int halve, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ sweep_src, int64_t sweep_len,
    const WT* __restrict__ ws, int64_t pitch, const float* __restrict__ rd_std, const float* __restrict__ noise, uint64_t seed,
    int64_t env_offset, int num_iters, int num_spin, int first_draw_proposes, int64_t* __restrict__ obj,
    int compute_obj, int batched) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // LDS: words | prop | ring | scratch | [tops] | rd_std.  The sweep schedule offsets `rp` live in `prop` (the proposals are
    // over when the sweep starts).  With 4 waves per tile the top-k merge buffer lives in the ring (idle between the threshold
    // pass and the first proposal round): 58 KB for G22, two workgroups per CU (the kernel's ~240 registers allow two waves per
    // SIMD).  With 8 waves per tile (a CU holds one tile: ls_pick_waves) waves 4..7 stage in `tops`.
    constexpr bool COMPACT = (W == kLsMergeWaves);
    constexpr bool VEC = ALIGNED, V4 = ALIGNED;
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    uint64_t* prop = words + (N + 2);
    int32_t* rp = reinterpret_cast<int32_t*>(prop);
    int32_t* ring = reinterpret_cast<int32_t*>(prop + N);
    int64_t* scratch = reinterpret_cast<int64_t*>(ring + kRing);
    float* tops = COMPACT ? reinterpret_cast<float*>(ring)
                          : reinterpret_cast<float*>(scratch + W * kWave);   // [kLsMergeWaves][kTopCap][64]
    // rd_std [N] in LDS: read per quad as one broadcast ds_read_b128 (kept in SGPRs a trip ahead it cost 64 scalar
    // registers and pushed the kernel to ~200 SGPR spills)
    // (compact layout: sdl lives IN the proposal tile -- only the threshold pass reads it (the lane = node rounds take rd_std from
    // memory into registers), and the first proposal is written after that pass's last barrier)
#ifdef RLS_LS_ROUND_LANE_ENV
    float* sdl = COMPACT ? reinterpret_cast<float*>(scratch + W * kWave) : tops + kLsMergeWaves * kTopCap * kWave;
#else
    float* sdl = COMPACT ? reinterpret_cast<float*>(prop) : tops + kLsMergeWaves * kTopCap * kWave;
#endif
    static_assert(kRing * 4 >= kLsMergeWaves * kTopCap * kWave * 4, "the ring holds the merge buffer");
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    const int64_t b = b0 + lane;
    const bool valid = b < B;
    const EnvKey env_key = ls_env_key(seed, (uint64_t)(b + env_offset));

    // ---- phase 0
    if (threadIdx.x == 0) words[N] = 0;
    // the ring is idle outside the sweep: it doubles as the row-piece stage of the tile load / store
    // waves 0..3 stage in the ring, waves 4..7 in the top-k merge buffer (free until the merge, see the barrier there)
    static_assert(kRing * 4 >= kLsMergeWaves * kStageBytes && kLsMergeWaves * kTopCap * kWave * 4 >= (W - kLsMergeWaves) * kStageBytes,
                  "ring (+ merge buffer for waves 4..7) double as the row-piece stages");
    unsigned char* wstage = (w < kLsMergeWaves ? reinterpret_cast<unsigned char*>(ring) + w * kStageBytes
                                               : reinterpret_cast<unsigned char*>(tops) + (w - kLsMergeWaves) * kStageBytes);
    unsigned char* stage = wstage;   // (rows that are not 4-byte aligned go through its funnel-shift form)
    for (int64_t i = threadIdx.x; i < ((N + 3) & ~3ll); i += W * kWave) sdl[i] = i < N ? rd_std[i] : 0.0f;
    tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane, w, W, stage);
    __syncthreads();
    int64_t my_obj;
    if (compute_obj) {
        my_obj = block_sum_partials<W>(tile_cut_count<P>(words, eu, ev, E, lane, w, W), scratch, lane, w);
        if (halve) my_obj >>= 1;
        __syncthreads();
    } else {
        my_obj = valid ? obj[b] : 0;
    }

    const WT* ws_row = ws + (valid ? b : 0) * pitch;   // rows `pitch` entries apart, a multiple of 16 bytes: every 16-byte piece lies inside its row
    constexpr int NPC = 16 / (int)sizeof(WT);     // nodes per 16-byte piece of a ws row
    constexpr int QPP = NPC / 4;                  // quads (4 nodes: the unit of the noise / mask code) per piece
    const int64_t nquads = (N + 3) >> 2;
    const int64_t npieces = (N + NPC - 1) / NPC;
    const int64_t nchunks = (npieces + 3) >> 2;   // chunk = 4 consecutive pieces = 64 bytes of a ws row
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    constexpr int D = 2;                          // chunks per trip; the next trip's loads fly during this trip's math
    // ws rows come in through the row-piece stage (rls_tile.h) like the spins: lane l of load i fetches piece
    // l >> 4 of row 16 i + (l & 15), 64-byte runs on the global side (a lane-per-env read of 16 B touched 64
    // cache lines per instruction and every trip waited a full memory round trip: 120 us per round).
    int io_r, io_j;
    stage_io_lane(lane, io_r, io_j);
    auto issue = [&](int64_t c0, i32x4 (&g)[D][4]) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t c = c0 + (int64_t)d * W;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                g[d][i] = i32x4{0, 0, 0, 0};
                if constexpr (V4) {
                    const int64_t rw = b0 + kStageRows * i + io_r;
                    const int64_t pc = c * 4 + io_j;
                    if (c < nchunks && rw < B && pc < npieces) g[d][i] = *reinterpret_cast<const i32x4*>(ws + rw * pitch + pc * NPC);
                } else {   // unaligned rows: each lane reads its own env's piece i of the chunk, element-wise
                    const int64_t n0 = (c * 4 + i) * NPC;
                    if (c < nchunks && valid) {
#pragma unroll
                        for (int k = 0; k < NPC; ++k) {
                            const uint32_t e = (n0 + k < N) ? (uint32_t)ws_row[n0 + k] & (sizeof(WT) == 1 ? 0xFFu : 0xFFFFu) : 0u;
                            constexpr int per = 4 / (int)sizeof(WT);          // entries per dword
                            g[d][i][k / per] |= (int32_t)(e << (8 * (int)sizeof(WT) * (k % per)));
                        }
                    }
                }
            }
        }
    };
    // spin_rand of quad q: v[k] = (float)ws + noise * rd_std.  Branch-free in the production path (a quad past the end
    // of the row -- the last piece of a row whose length is not a multiple of the piece -- is computed on clamped
    // inputs and masked by the consumer): the quads of a piece are independent chains the scheduler interleaves,
    // which is what hides the latency of the hash -> log -> sqrt -> sin / cos sequence with two waves per SIMD.
    auto spin_rand_quad = [&](int64_t q, int it, const int (&wq)[4], float (&v)[4], auto use_noise) {
        const int64_t qs = q < nquads ? q : nquads - 1;
        const f32x4 sdq = *reinterpret_cast<const f32x4*>(sdl + qs * 4);   // wave-uniform address: broadcast read
        float z[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (decltype(use_noise)::value) {
            if (valid && q < nquads) {
                if constexpr (V4) {
                    const f32x4 zz = *reinterpret_cast<const f32x4*>(noise + ((int64_t)it * B + b) * N + q * 4);
                    z[0] = zz[0]; z[1] = zz[1]; z[2] = zz[2]; z[3] = zz[3];
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (q * 4 + k < N) z[k] = noise[((int64_t)it * B + b) * N + q * 4 + k];
                }
            }
        } else {
            normal4(env_key, (uint32_t)q, (uint32_t)it, z);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = (float)wq[k] + z[k] * sdq[k];   // two roundings (fp-contract off): torch's ws + randn * rd_std
    };
    // the NPC nodes of piece pc (16 bytes of this lane's ws row): f(pc, v[NPC]) sees them all at once, so that a
    // consumer with a divergent store (the mask words) pays for ONE exec-masked region per piece, not one per quad
    auto do_piece = [&](int64_t pc, const i32x4& piece, int it, auto use_noise, auto&& f) {
        float vals[NPC];
#pragma unroll
        for (int sq = 0; sq < QPP; ++sq) {
            const int64_t q = pc * QPP + sq;
            int wv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr (sizeof(WT) == 1) wv[k] = (int)(int8_t)((uint32_t)piece[sq] >> (8 * k));
                else wv[k] = (int)(int16_t)((uint32_t)piece[2 * sq + (k >> 1)] >> (16 * (k & 1)));
            }
            float v[4];
            spin_rand_quad(q, it, wv, v, use_noise);
#pragma unroll
            for (int k = 0; k < 4; ++k) vals[4 * sq + k] = v[k];
        }
        f(pc, vals);
    };
    // one pass over the tile's ws: f(pc, v) for every piece pc of this wave's chunks (f masks nodes >= N itself)
    auto for_each_quad = [&](int it, auto use_noise, auto&& f) {
        i32x4 ga[D][4], gb[D][4];
        issue(w, ga);
        for (int64_t c0 = w; c0 < nchunks; c0 += (int64_t)D * W) {
            issue(c0 + (int64_t)D * W, gb);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int64_t c = c0 + (int64_t)d * W;
                if (c < nchunks) {
                    const int np_here = (int)((npieces - c * 4) < 4 ? (npieces - c * 4) : 4);
                    if constexpr (V4) {
                        // corner turn through this wave's stage: afterwards slot (lane, i) = this lane's env, piece 4c + i.
                        // The pieces are then taken one at a time by a ROLLED loop (a piece = 16 / 8 nodes: a loop body of
                        // a few hundred instructions; unrolled over the chunk the round loop was 10 000 instructions
                        // of single-quad basic blocks)
#pragma unroll
                        for (int i = 0; i < 4; ++i) *reinterpret_cast<i32x4*>(wstage + (i << 10) + (lane << 4)) = ga[d][i];
                        asm volatile("" ::: "memory");   // LDS ops of one wave execute in order
#pragma unroll 1
                        for (int i = 0; i < np_here; ++i) {
                            const i32x4 piece = *reinterpret_cast<const i32x4*>(wstage + stage_slot_off(lane, i));
                            do_piece(c * 4 + i, piece, it, use_noise, f);
                        }
                        asm volatile("" ::: "memory");
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (i < np_here) do_piece(c * 4 + i, ga[d][i], it, use_noise, f);
                    }
                }
            }
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) ga[d][i] = gb[d][i];
        }
    };
    using with_noise = std::integral_constant<bool, true>;
    using no_noise = std::integral_constant<bool, false>;
    // ---- phase 1: threshold = (num_spin + 1)-th largest of the first draw
    float t[kTopCap];
#pragma unroll
    for (int j = 0; j < kTopCap; ++j) t[j] = -INFINITY;
    auto top_pass = [&](auto use_noise, auto depth) {    // depth = entries of the insertion network that can matter
        for_each_quad(0, use_noise, [&](int64_t pc, const float (&v)[NPC]) {
#pragma unroll
            for (int k = 0; k < NPC; ++k) top_insert_n<decltype(depth)::value>(t, (pc * NPC + k < N) ? v[k] : -INFINITY);
        });
    };
    if (num_spin < 9) {
        if (noise) top_pass(with_noise{}, std::integral_constant<int, 9>{}); else top_pass(no_noise{}, std::integral_constant<int, 9>{});
    } else {
        if (noise) top_pass(with_noise{}, std::integral_constant<int, kTopCap>{}); else top_pass(no_noise{}, std::integral_constant<int, kTopCap>{});
    }
    __syncthreads();   // every wave is done with its stage (waves >= 4 stage inside `tops`)
    // waves 4.. hand their lists to waves 0..3, four at a time, through the merge buffer
    for (int base = kLsMergeWaves; base < W; base += kLsMergeWaves) {
        if (w >= base && w < base + kLsMergeWaves) {
#pragma unroll
            for (int j = 0; j < kTopCap; ++j) tops[((w - base) * kTopCap + j) * kWave + lane] = t[j];
        }
        __syncthreads();
        if (w < kLsMergeWaves && base + w < W) {
#pragma unroll
            for (int j = 0; j < kTopCap; ++j) top_insert(t, tops[(w * kTopCap + j) * kWave + lane]);
        }
        __syncthreads();
    }
    if (w < kLsMergeWaves) {
#pragma unroll
        for (int j = 0; j < kTopCap; ++j) tops[(w * kTopCap + j) * kWave + lane] = t[j];
    }
    __syncthreads();
    float thresh = 0.0f;
    if (w < kLsMergeWaves) {
        for (int ow = 0; ow < kLsMergeWaves; ++ow) {
            if (ow == w) continue;
#pragma unroll
            for (int j = 0; j < kTopCap; ++j) top_insert(t, tops[(ow * kTopCap + j) * kWave + lane]);
        }
        thresh = t[0];
#pragma unroll
        for (int j = 1; j < kTopCap; ++j) thresh = (j == num_spin) ? t[j] : thresh;   // kthvalue(k = N - num_spin)
    }
    __syncthreads();
    if constexpr (W > kLsMergeWaves) {   // the other waves read the threshold of their lane's env
        if (w == 0) tops[lane] = thresh;
        __syncthreads();
        thresh = tops[lane];
        __syncthreads();
    }

    // ---- phase 2: proposal rounds
#ifdef RLS_LS_ROUND_LANE_ENV
    const uint64_t valid_mask = ballot64(valid);
#endif
    for (int itp = 0; itp < num_iters; ++itp) {
        const int it = first_draw_proposes ? itp : itp + 1;
#ifdef RLS_LS_ROUND_LANE_ENV   // dev build: the lane = env round (until round 5), for A/B timing
        auto round_pass = [&](auto use_noise) {
        for_each_quad(it, use_noise, [&](int64_t pc, const float (&v)[NPC]) {
            const uint64_t mine = ls_piece_mask_words<NPC>(v, thresh, valid_mask);   // spin_rand.gt(thresh), bit e = env b0 + e
            const int64_t node = pc * NPC + lane;
            if (lane < NPC && node < N) prop[node] = words[node] ^ mine;
        });
        };
        if (noise) round_pass(with_noise{}); else round_pass(no_noise{});
#else
        {   // spin_rand.gt(thresh) with lane = node: a wave's tasks write their 32-bit halves of prop[] (bit e = env b0 + e)
            uint32_t* prop32 = reinterpret_cast<uint32_t*>(prop);
            const uint32_t* words32 = reinterpret_cast<const uint32_t*>(words);
            auto put = [&](int64_t node, int half, uint32_t wd) { prop32[2 * node + half] = words32[2 * node + half] ^ wd; };
            const float th_e = valid ? thresh : INFINITY;
#ifndef RLS_LS_LB
#define RLS_LS_LB 8
#endif
            if (noise) ls_round_words_lane_node<WT, true>(ws, pitch, B, N, b0, lane, w, W, rd_std, env_key, th_e, noise + (int64_t)it * B * N, it, put);
            else ls_round_words_lane_node<WT, false, RLS_LS_LB>(ws, pitch, B, N, b0, lane, w, W, rd_std, env_key, th_e, nullptr, it, put);
        }
#endif
        __syncthreads();
        int64_t total = block_sum_partials<W>(tile_cut_count<P>(prop, eu, ev, E, lane, w, W), scratch, lane, w);
        if (halve) total >>= 1;
        const bool accept = valid && (total >= my_obj);          // update_xs_by_vs: vs1.ge(vs0)
        if (accept) my_obj = total;
        const uint64_t am = ballot64(accept);
        __syncthreads();
        for (int64_t n = threadIdx.x; n < N; n += W * kWave) words[n] ^= (words[n] ^ prop[n]) & am;
        __syncthreads();
    }
    // ---- phase 3: greedy sweep on the resident tile
    // rp: CSR rowptr / schedule offsets (N + 1 entries), or the level-group offsets (sweep_len + 1 <= N + 1 entries); they
    // move into `prop` now that the proposals are over
    for (int64_t i = threadIdx.x; i <= (batched == 2 ? sweep_len : N); i += W * kWave) rp[i] = rowptr[i];
    __syncthreads();
    if (batched == 2) {   // level-parallel (lane = node): sweep_src = group records, sweep_len = number of groups
        // my_obj IS the cut of the tile here (counted in phase 0 or handed in as such, then every accepted proposal's count): the
        // sweep's result is one count of the new tile -- not after - before (a tenth of the kernel's edge counts) -- and the tile's
        // stores go out first, draining under that count (both only read the words)
        sweep_tile_levels<W>(words, rp, sweep_src, sweep_len, N, lane, w);
        tile_store_bytes<VEC>(x, B, N, b0, words, lane, w, W, true, stage);
        const int64_t after = block_sum_partials<W>(tile_cut_count<P>(words, eu, ev, E, lane, w, W), scratch, lane, w);
        if (w == 0 && valid) obj[b] = halve ? (after >> 1) : after;
        return;
    } else if (batched) {   // all W waves over the host-built level schedule, one node per wave step (rp carries the batch flags)
        const int64_t part = sweep_tile_batched<W>(words, rp, ring, sweep_src, sweep_len, N, lane, w);
        my_obj += block_sum_partials<W>(part, scratch, lane, w);
        if (w == 0 && valid) obj[b] = my_obj;
    } else if (w == 0) {   // one wave, strictly sequential
        my_obj += sweep_tile(words, rp, ring, sweep_src, sweep_len, N, lane);
        if (valid) obj[b] = my_obj;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- phase 4
    tile_store_bytes<VEC>(x, B, N, b0, words, lane, w, W, true, stage);
}

// =====================================================================================
// The same local search as separate launches, for graphs the fused kernel does not cover (its LDS layout holds two
// tiles + rd_std: N <= ~6500): the threshold pass and one proposal round per launch, with the fused kernel's
// draws (normal4 keyed by seed, global env, node quad, draw index) -- so for a given seed both forms give the same
// result.  Before, these steps were torch ops on [B, N] f32 tensors (randn, multiply-add, kthvalue, gt) around K6.
// Rows must be 16-byte multiples (the row-piece stage); ws int8 / int16.
// =====================================================================================

// The draws themselves, as a tensor: out[b, n] = normal(seed, env_offset + b, n, draw), the value every local-search kernel
// above and below uses for that (env, node, draw).  For the decomposed path (weights wider than 16 bits, graphs outside
// every tile form), which used torch.randn -- not keyed by the global env -- and for the statistical tests of the draws.
__global__ __launch_bounds__(256) void k_ls_normals(float* __restrict__ out, int64_t B, int64_t N, uint64_t seed,
                                                   int64_t env_offset, int draw) {
    const int64_t nquads = (N + 3) >> 2;
    const int64_t total = B * nquads;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / nquads, q = i - b * nquads;
        float z[4];
        normal4(ls_env_key(seed, (uint64_t)(b + env_offset)), (uint32_t)q, (uint32_t)draw, z);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (q * 4 + k < N) out[b * N + q * 4 + k] = z[k];
    }
}

// One pass over the ws rows of the tile's 64 envs (lane = env): f(pc, v[NPC]) for every 16-byte piece pc of this wave's
// chunks, v = (float)ws + normal * rd_std (two roundings, as torch); f masks nodes >= N itself.  `sd` = rd_std in LDS or
// global memory (wave-uniform addresses either way).
template <typename WT>
__host__ __device__ inline int64_t ls_num_chunks(int64_t N) {   // chunk = 4 pieces of 16 bytes = 64 bytes of a ws row
    constexpr int NPC = 16 / (int)sizeof(WT);
    return ((N + NPC - 1) / NPC + 3) >> 2;
}
// the chunks of slice `sl` of `S` (a tile's pass may be split over S workgroups when there are few tiles)
__device__ __forceinline__ void ls_slice_chunks(int64_t nchunks_all, int sl, int S, int64_t& c_begin, int64_t& c_end) {
    const int64_t per = (nchunks_all + S - 1) / S;
    c_begin = per * sl;
    c_end = c_begin + per < nchunks_all ? c_begin + per : nchunks_all;
}

template <typename WT, int W, typename F>
__device__ __forceinline__ void ls_ws_pass(const WT* __restrict__ ws, int64_t pitch, int64_t B, int64_t N, int64_t b0, int lane, int w,
                                           unsigned char* wstage, const float* sd, EnvKey env_key, int it,
                                           int64_t c_begin, int64_t nchunks, F&& f) {   // chunks [c_begin, nchunks)
    typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
    constexpr int NPC = 16 / (int)sizeof(WT), QPP = NPC / 4, D = 2;
    const int64_t nquads = (N + 3) >> 2, npieces = (N + NPC - 1) / NPC;
    int io_r, io_j;
    stage_io_lane(lane, io_r, io_j);
    auto issue = [&](int64_t c0, i32x4 (&g)[D][4]) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t c = c0 + (int64_t)d * W;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t rw = b0 + kStageRows * i + io_r, pc = c * 4 + io_j;
                g[d][i] = (c < nchunks && rw < B && pc < npieces) ? *reinterpret_cast<const i32x4*>(ws + rw * pitch + pc * NPC)
                                                                  : i32x4{0, 0, 0, 0};
            }
        }
    };
    i32x4 ga[D][4], gb[D][4];
    issue(c_begin + w, ga);
    for (int64_t c0 = c_begin + w; c0 < nchunks; c0 += (int64_t)D * W) {
        issue(c0 + (int64_t)D * W, gb);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int64_t c = c0 + (int64_t)d * W;
            if (c < nchunks) {
                const int np_here = (int)((npieces - c * 4) < 4 ? (npieces - c * 4) : 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) *reinterpret_cast<i32x4*>(wstage + (i << 10) + (lane << 4)) = ga[d][i];
                asm volatile("" ::: "memory");   // LDS ops of one wave execute in order
#pragma unroll 1
                for (int i = 0; i < np_here; ++i) {
                    const i32x4 piece = *reinterpret_cast<const i32x4*>(wstage + stage_slot_off(lane, i));
                    const int64_t pc = c * 4 + i;
                    float vals[NPC];
#pragma unroll
                    for (int sq = 0; sq < QPP; ++sq) {
                        const int64_t q = pc * QPP + sq;
                        const int64_t qs = q < nquads ? q : nquads - 1;
                        const f32x4 sdq = *reinterpret_cast<const f32x4*>(sd + qs * 4);
                        float z[4];
                        normal4(env_key, (uint32_t)q, (uint32_t)it, z);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            int wv;
                            if constexpr (sizeof(WT) == 1) wv = (int)(int8_t)((uint32_t)piece[sq] >> (8 * k));
                            else wv = (int)(int16_t)((uint32_t)piece[2 * sq + (k >> 1)] >> (16 * (k & 1)));
                            vals[4 * sq + k] = (float)wv + z[k] * sdq[k];
                        }
                    }
                    f(pc, vals);
                }
                asm volatile("" ::: "memory");
            }
        }
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) ga[d][i] = gb[d][i];
    }
}

constexpr int kLsRoundWaves = 8;

// thresh[b] = the (num_spin + 1)-th largest of ws[b, :] + normal(draw) * rd_std  (kthvalue(k = N - num_spin))
// SD_LDS: rd_std [N] staged in LDS (one broadcast read per quad); false: read from global memory -- rows so long that the
// 4 N bytes do not fit beside the lists and the stages (N > ~24 900; N % 4 == 0 so that a quad never leaves the array)
template <typename WT, bool SD_LDS = true>
__global__ __launch_bounds__(kLsRoundWaves * kWave) void k_ls_threshold(const WT* __restrict__ ws, int64_t pitch, int64_t B, int64_t N,
                                                                        const float* __restrict__ rd_std, uint64_t seed,
                                                                        int64_t env_offset, int draw, int num_spin,
                                                                        float* __restrict__ thresh, float* __restrict__ partial) {
    // gridDim.y > 1: this workgroup takes one slice of the rows and leaves its merged list in partial[tile][slice][kTopCap][64]
    // (k_ls_threshold_merge reads the threshold off the slices' lists); else it writes thresh itself
    constexpr int W = kLsRoundWaves, NPC = 16 / (int)sizeof(WT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* tops = reinterpret_cast<float*>(smem);                                  // [W][kTopCap][64]
    unsigned char* stages = smem + (size_t)W * kTopCap * kWave * 4;
    float* sdl = reinterpret_cast<float*>(stages + (size_t)W * kStageBytes);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave, b = b0 + lane;
    const EnvKey env_key = ls_env_key(seed, (uint64_t)(b + env_offset));
    if constexpr (SD_LDS) {
        for (int64_t i = threadIdx.x; i < ((N + 3) & ~3ll); i += W * kWave) sdl[i] = i < N ? rd_std[i] : 0.0f;   // rd_std: one broadcast read per quad
        __syncthreads();
    }
    float t[kTopCap];
#pragma unroll
    for (int j = 0; j < kTopCap; ++j) t[j] = -INFINITY;
    int64_t c_begin, c_end;
    ls_slice_chunks(ls_num_chunks<WT>(N), (int)blockIdx.y, (int)gridDim.y, c_begin, c_end);
    auto pass = [&](auto depth) {
        ls_ws_pass<WT, W>(ws, pitch, B, N, b0, lane, w, stages + (size_t)w * kStageBytes, SD_LDS ? sdl : rd_std, env_key, draw, c_begin, c_end,
                          [&](int64_t pc, const float (&v)[NPC]) {
#pragma unroll
                              for (int k = 0; k < NPC; ++k) top_insert_n<decltype(depth)::value>(t, (pc * NPC + k < N) ? v[k] : -INFINITY);
                          });
    };
    if (num_spin < 9) pass(std::integral_constant<int, 9>{}); else pass(std::integral_constant<int, kTopCap>{});
#pragma unroll
    for (int j = 0; j < kTopCap; ++j) tops[(w * kTopCap + j) * kWave + lane] = t[j];
    __syncthreads();
    if (w == 0) {
        for (int ow = 1; ow < W; ++ow)
#pragma unroll
            for (int j = 0; j < kTopCap; ++j) top_insert(t, tops[(ow * kTopCap + j) * kWave + lane]);
        if (gridDim.y > 1) {
            float* dst = partial + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * kTopCap * kWave;
#pragma unroll
            for (int j = 0; j < kTopCap; ++j) dst[j * kWave + lane] = t[j];
        } else {
            float th = t[0];
#pragma unroll
            for (int j = 1; j < kTopCap; ++j) th = (j == num_spin) ? t[j] : th;
            if (b < B) thresh[b] = th;
        }
    }
}

__global__ __launch_bounds__(kWave) void k_ls_threshold_merge(const float* __restrict__ partial, int S, int64_t B, int num_spin,
                                                               float* __restrict__ thresh) {
    const int lane = threadIdx.x;
    const int64_t b = (int64_t)blockIdx.x * kWave + lane;
    float t[kTopCap];
#pragma unroll
    for (int j = 0; j < kTopCap; ++j) t[j] = -INFINITY;
    for (int sl = 0; sl < S; ++sl) {
        const float* src = partial + ((int64_t)blockIdx.x * S + sl) * kTopCap * kWave;
#pragma unroll
        for (int j = 0; j < kTopCap; ++j) top_insert(t, src[j * kWave + lane]);
    }
    float th = t[0];
#pragma unroll
    for (int j = 1; j < kTopCap; ++j) th = (j == num_spin) ? t[j] : th;
    if (b < B) thresh[b] = th;
}

// the mask words of one proposal round for a slice of the nodes: maskw[tile][node] (bit e = env 64 tile + e), for batches of so
// few tiles that one workgroup per tile would leave most of the chip idle through the VALU-bound noise generation
template <typename WT, bool SD_LDS = true>
__global__ __launch_bounds__(kLsRoundWaves * kWave) void k_ls_mask(const WT* __restrict__ ws, int64_t pitch, int64_t B, int64_t N,
                                                                   const float* __restrict__ rd_std, const float* __restrict__ thresh,
                                                                   uint64_t seed, int64_t env_offset, int first_draw,
                                                                   uint64_t* __restrict__ maskw_all, int64_t round_words) {
    // blockIdx.z = the round: draw first_draw + z into maskw_all + z * round_words (all rounds of a call in ONE launch: a round of a
    // 4096-env batch is 128 workgroups -- 8 launches of 12.8 us where the chip does the eight in ~25)
    const int draw = first_draw + (int)blockIdx.z;
    uint64_t* maskw = maskw_all + (int64_t)blockIdx.z * round_words;
    constexpr int W = kLsRoundWaves;
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave, b = b0 + lane;
    const bool valid = b < B;
    const EnvKey env_key = ls_env_key(seed, (uint64_t)(b + env_offset));
    const float th = valid ? thresh[b] : INFINITY;
    // lane = node (ls_round_words_lane_node): the tile's tasks over the waves of its gridDim.y workgroups; rd_std straight from
    // global memory into registers (SD_LDS and the dynamic LDS the launcher still sizes are unused by this form)
    uint32_t* out32 = reinterpret_cast<uint32_t*>(maskw + (int64_t)blockIdx.x * N);
    ls_round_words_lane_node<WT, false>(ws, pitch, B, N, b0, lane, (int)blockIdx.y * W + w, (int)gridDim.y * W, rd_std, env_key, th, nullptr, draw,
                                        [&](int64_t node, int half, uint32_t wd) { out32[2 * node + half] = wd; });
}

// one proposal round: x ^= (ws + normal(draw) * rd_std > thresh) for the envs whose cut does not decrease; obj updated
// PREMASK: the mask words come from k_ls_mask (maskw[tile][node]) instead of being generated here
template <typename WT, int P, bool SD_LDS, bool PREMASK>
__global__ __launch_bounds__(kLsRoundWaves * kWave) void k_ls_propose(uint8_t* __restrict__ x, int64_t B, int64_t N,
                                                                      const int32_t* __restrict__ eu, const int32_t* __restrict__ ev,
                                                                      int64_t E, int halve, const WT* __restrict__ ws, int64_t pitch,
                                                                      const float* __restrict__ rd_std, const float* __restrict__ thresh,
                                                                      uint64_t seed, int64_t env_offset, int draw,
                                                                      int64_t* __restrict__ obj, const uint64_t* __restrict__ maskw,
                                                                      int x_aligned) {   // rows of x 4-byte aligned on a 16-byte base
    constexpr int W = kLsRoundWaves, NPC = 16 / (int)sizeof(WT);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(words + ((N + 1) & ~1ll));
    unsigned char* stages = reinterpret_cast<unsigned char*>(scratch + W * kWave);
    float* sdl = reinterpret_cast<float*>(stages + (size_t)W * kStageBytes);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave, b = b0 + lane;
    const bool valid = b < B;
    const EnvKey env_key = ls_env_key(seed, (uint64_t)(b + env_offset));
    unsigned char* stage = stages + (size_t)w * kStageBytes;
    if constexpr (SD_LDS && !PREMASK)
        for (int64_t i = threadIdx.x; i < ((N + 3) & ~3ll); i += W * kWave) sdl[i] = i < N ? rd_std[i] : 0.0f;
    if (x_aligned) tile_load_bits<uint8_t, true>(x, B, N, b0, words, lane, w, W, stage);
    else tile_load_bits<uint8_t, false>(x, B, N, b0, words, lane, w, W, stage);
    __syncthreads();
    if constexpr (PREMASK) {
        const uint64_t* mw = maskw + (int64_t)blockIdx.x * N;
        for (int64_t n = threadIdx.x; n < N; n += W * kWave) words[n] ^= mw[n];
    } else {
        const float th = valid ? thresh[b] : INFINITY;
        // the mask words go straight into the tile: a node belongs to exactly one piece of one wave
        uint32_t* words32 = reinterpret_cast<uint32_t*>(words);
        ls_round_words_lane_node<WT, false>(ws, pitch, B, N, b0, lane, w, W, rd_std, env_key, th, nullptr, draw,
                                            [&](int64_t node, int half, uint32_t wd) { words32[2 * node + half] ^= wd; });
    }
    __syncthreads();
    int64_t total = block_sum_partials<W>(tile_cut_count<P>(words, eu, ev, E, lane, w, W), scratch, lane, w);
    if (halve) total >>= 1;
    const bool accept = valid && (total >= obj[b]);           // update_xs_by_vs: vs1.ge(vs0)
    __syncthreads();                                          // every wave has read obj[b] before wave 0 updates it
    if (accept && w == 0) obj[b] = total;
    if (x_aligned) tile_store_bytes<true>(x, B, N, b0, words, lane, w, W, accept, stage);
    else tile_store_bytes<false>(x, B, N, b0, words, lane, w, W, accept, stage);
}

// All the proposal rounds of a small batch on ONE load of the tile: the mask words of every round come from k_ls_mask
// (maskw[round][tile][node]; they depend on the weights, the threshold and the draw, not on x), a round is XOR -> count ->
// accept -> XOR back where rejected, the tile goes out once.  Per round that is two passes over 8 N bytes of L2-resident mask
// words instead of a tile load from and a tile store to HBM plus a launch (BA n = 10^4, 4096 envs: 47 us per round).
template <int P, int W>
__global__ __launch_bounds__(W * kWave) void k_ls_apply_rounds(uint8_t* __restrict__ x, int64_t B, int64_t N,
                                                               const int32_t* __restrict__ eu, const int32_t* __restrict__ ev,
                                                               int64_t E, int halve, const uint64_t* __restrict__ maskw,
                                                               int rounds, int64_t* __restrict__ obj, int x_aligned, int has_stage) {
    // W = 8 with a row-piece stage per wave, or W = 4 without stages where the tile nearly fills LDS (N ~ 20 000)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(words + ((N + 1) & ~1ll));
    unsigned char* stages = reinterpret_cast<unsigned char*>(scratch + W * kWave);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave, b = b0 + lane;
    const bool valid = b < B;
    unsigned char* stage = has_stage ? stages + (size_t)w * kStageBytes : nullptr;
    if (x_aligned) tile_load_bits<uint8_t, true>(x, B, N, b0, words, lane, w, W, stage);
    else tile_load_bits<uint8_t, false>(x, B, N, b0, words, lane, w, W, stage);
    int64_t my_obj = valid ? obj[b] : 0;
    bool changed = false;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const uint64_t* mw = maskw + ((int64_t)r * gridDim.x + blockIdx.x) * N;
        for (int64_t n = threadIdx.x; n < N; n += W * kWave) words[n] ^= mw[n];
        __syncthreads();
        int64_t total = block_sum_partials<W>(tile_cut_count<P>(words, eu, ev, E, lane, w, W), scratch, lane, w);
        if (halve) total >>= 1;
        const bool accept = valid && (total >= my_obj);          // update_xs_by_vs: vs1.ge(vs0)
        if (accept) my_obj = total;
        changed = changed || accept;
        const uint64_t am = ballot64(accept);
        __syncthreads();
        if (~am)
            for (int64_t n = threadIdx.x; n < N; n += W * kWave) words[n] ^= mw[n] & ~am;
        __syncthreads();
    }
    if (w == 0 && valid) obj[b] = my_obj;
    if (x_aligned) tile_store_bytes<true>(x, B, N, b0, words, lane, w, W, changed, stage);
    else tile_store_bytes<false>(x, B, N, b0, words, lane, w, W, changed, stage);
}

// the same on HALF tiles (rls_tile32.h) for graphs past the 64-env tile: the mask words stay those of 64-env tiles, half tile h
// takes the low (even h) or high dword of word (h / 2, n)
template <int P, int W>
__global__ __launch_bounds__(W * kWave) void k_ls_apply_rounds32(uint8_t* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ eu,
                                                                 const int32_t* __restrict__ ev, int64_t E, int halve,
                                                                 const uint64_t* __restrict__ maskw, int64_t tiles64, int rounds,
                                                                 int64_t* __restrict__ obj, int x_aligned, int has_stage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* words32 = reinterpret_cast<uint32_t*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + (((size_t)N * 4 + 15) & ~(size_t)15));
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kHalf, b = b0 + (lane & (kHalf - 1));
    const bool valid = b < B && lane < kHalf;
    unsigned char* stage = has_stage ? reinterpret_cast<unsigned char*>(scratch + W * kWave) + (size_t)w * kStageBytes : nullptr;
    if (x_aligned) tile32_load_bits<uint8_t, true>(x, B, N, b0, words32, lane, w, W, stage);
    else tile32_load_bits<uint8_t, false>(x, B, N, b0, words32, lane, w, W, stage);
    int64_t my_obj = valid ? obj[b] : 0;
    bool changed = false;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const uint32_t* mw = reinterpret_cast<const uint32_t*>(maskw + ((int64_t)r * tiles64 + (blockIdx.x >> 1)) * N) + (blockIdx.x & 1);
        for (int64_t n = threadIdx.x; n < N; n += W * kWave) words32[n] ^= mw[2 * n];
        __syncthreads();
        int64_t total = block_sum_partials<W>(tile32_cut_count<P>(words32, eu, ev, E, lane, w, W), scratch, lane, w);
        if (halve) total >>= 1;
        const bool accept = valid && (total >= my_obj);          // update_xs_by_vs: vs1.ge(vs0)
        if (accept) my_obj = total;
        changed = changed || accept;
        const uint32_t am = (uint32_t)ballot64(accept);
        __syncthreads();
        if (~am)
            for (int64_t n = threadIdx.x; n < N; n += W * kWave) words32[n] ^= mw[2 * n] & ~am;
        __syncthreads();
    }
    if (w == 0 && valid) obj[b] = my_obj;
    const bool ch_env = (bool)((ballot64(changed) >> (lane & (kHalf - 1))) & 1ull);      // lanes 32..63 store their env's second block
    if (x_aligned) tile32_store_bytes<true>(x, B, N, b0, words32, lane, w, W, ch_env, stage);
    else tile32_store_bytes<false>(x, B, N, b0, words32, lane, w, W, ch_env, stage);
}

static size_t ls_apply32_lds(int64_t N, int W, bool stage = false) {
    return (((size_t)N * 4 + 15) & ~(size_t)15) + (size_t)W * kWave * 8 + (stage ? (size_t)W * kStageBytes : 0);
}

static size_t ls_apply_lds(int64_t N, int W, bool stage) {
    return (size_t)((N + 1) & ~1ll) * 8 + (size_t)W * kWave * 8 + (stage ? (size_t)W * kStageBytes : 0);
}
// N beyond the proposal kernel's tile + stages but within the bare tile (15 500 < N <= 20 224): mask kernels + the 4-wave apply kernel
// (rows of 16- or 8-byte multiples take half tiles there instead: their loader keeps 64-byte runs, the bare tile's reads 16 B per env)
static bool ls_big_tile(int64_t N) {
    return (size_t)((N + 1) & ~1ll) * 8 + (size_t)kLsRoundWaves * kWave * 8 + (size_t)kLsRoundWaves * kStageBytes > (size_t)kLdsBytes &&
           ls_apply_lds(N, 4, false) <= (size_t)kLdsBytes && (N & 7) != 0;
}
// N past the 64-env tile altogether but within the half tile (20 224 < N <= ~39 900): mask kernels + the apply kernel on half tiles
static bool ls_half_tile(int64_t N) {
    return !ls_big_tile(N) && ls_apply32_lds(N, kLsRoundWaves) <= (size_t)kLdsBytes &&
           (size_t)((N + 1) & ~1ll) * 8 + (size_t)kLsRoundWaves * kWave * 8 + (size_t)kLsRoundWaves * kStageBytes > (size_t)kLdsBytes;
}
static bool ls_sd_global() {   // dev knob: rd_std read from global memory even where it fits LDS
    const bool on = knob_on(KN_LS_SD_GLOBAL);
    return on;
}
// workgroups per tile for the noise passes: 1 once the tiles alone fill the chip, else up to 8 slices of the rows
static int ls_slices(int64_t B, int64_t nchunks) {
    const int force = (int)knob(KN_LS_SLICES, 0);   // dev knob
    const int64_t tiles = ceil_div(B, kWave);
    int S = force > 0 ? force : (int)(num_cus() / (tiles > 0 ? tiles : 1));
    if (S > 8) S = 8;
    if ((int64_t)S * 2 * kLsRoundWaves > nchunks) S = (int)(nchunks / (2 * kLsRoundWaves));   // a slice keeps every wave busy
    return S < 1 ? 1 : S;
}
constexpr size_t kLsMaxMaskBytes = (size_t)1 << 30;   // all rounds' mask words at once only while they stay under 1 GiB
static size_t ls_scratch_bytes(int64_t B, int64_t N, int S, int rounds = 1) {   // rounds: mask words of that many rounds at once
    const size_t tiles = (size_t)ceil_div(B, kWave);
    const size_t lists = S > 1 ? tiles * S * kTopCap * kWave * 4 : 0;
    const size_t all = tiles * (size_t)N * 8 * (size_t)(rounds > 1 ? rounds : 1);
    // one round's words when the noise pass is split (S > 1); every round's whenever more than one is asked for and they fit
    const size_t masks = rounds > 1 ? (all <= kLsMaxMaskBytes ? all : (S > 1 ? tiles * (size_t)N * 8 : 0)) : (S > 1 ? all : 0);
    return lists > masks ? lists : masks;
}
// (rd_std always fits LDS beside the stages here: 4 N bytes, N bounded by the proposal kernel's tile)
// rd_std beside the stages (sd_lds), or read from global memory where 4 N bytes do not fit (needs N % 4 == 0)
// (the lane = node mask kernel keeps nothing in LDS: rd_std in registers, ws read as it lies)
static size_t ls_mask_lds(int64_t, bool = true) { return 0; }
static size_t ls_threshold_lds(int64_t N, bool sd_lds = true) {
    return (size_t)kLsRoundWaves * kTopCap * kWave * 4 + (size_t)kLsRoundWaves * kStageBytes + (sd_lds ? (size_t)((N + 3) & ~3ll) * 4 : 0);
}
static bool ls_noise_sd_lds(int64_t N) { return ls_threshold_lds(N, true) <= (size_t)kLdsBytes; }   // (the threshold kernel is the larger)
static bool ls_noise_passes_fit(int64_t N) { return ls_noise_sd_lds(N) || (N & 3) == 0; }
static size_t ls_propose_lds(int64_t N, bool sd_lds) {
    return (size_t)((N + 1) & ~1ll) * 8 + (size_t)kLsRoundWaves * kWave * 8 + (size_t)kLsRoundWaves * kStageBytes +
           (sd_lds ? (size_t)((N + 3) & ~3ll) * 4 : 0);
}
// N past the half tile too, within the narrow tiles (rls_tile32.h: 16 / 8 envs per workgroup; ~39 900 < N <= ~155 000): mask kernels,
// then every round's proposal through K6 on narrow tiles with the mask words as its bit-packed mask (round 5; such graphs ran the
// search as torch ops on [B, N] float tensors around K6: 16.6 ms for 4096 envs at N = 44 000)
static bool ls_narrow_tile(int64_t N) {
    return !ls_big_tile(N) && !ls_half_tile(N) && ls_propose_lds(N, false) > (size_t)kLdsBytes &&
           (((size_t)(N + 2) + 15) & ~(size_t)15) + (size_t)8 * kWave * 8 <= (size_t)kLdsBytes && knob(KN_NARROW_TILE, 1) != 0;
}

}  // namespace rls

using namespace rls;


// LDS bytes of the W-wave layout (W = 4: compact, the sweep offsets reuse the proposal tile; W = 8: separate tables)
static size_t ls_lds_bytes(int64_t N, int W) {
    const size_t sd_bytes = (size_t)((N + 3) & ~3ll) * 4;   // rd_std in LDS (W = 8; the compact layout keeps it in the proposal tile)
#ifdef RLS_LS_ROUND_LANE_ENV
    const size_t sd4 = sd_bytes;
#else
    const size_t sd4 = 0;
#endif
    return W == kLsMergeWaves
               ? sd4 + (size_t)(N + 2) * 8 + (size_t)N * 8 + (size_t)kRing * 4 + (size_t)W * kWave * 8
               : sd_bytes + (size_t)(N + 2) * 8 + (size_t)N * 8 + (size_t)kRing * 4 + (size_t)W * kWave * 8 +
                     (size_t)kLsMergeWaves * kTopCap * kWave * 4;
}

// Waves per tile.  What counts is waves per SIMD (the VALU issues a wave's stream at ~5 cycles per instruction and two or more
// waves' at ~2.6: tools/ceilings/valu_issue.hip): 8 waves when a CU holds one tile anyway -- at most one tile per CU in the launch, or
// rows so long that two 4-wave layouts do not fit LDS (N > ~3100) -- else 4 waves and two tiles per CU.  Measured
// (tools/timing/ls_waves.py): G22-sized 2^14 envs 0.53 -> 0.46 ms with 8, 2^15 0.83 -> 0.72 with 4; G(5000, 20000) 2^16 4.02 -> 3.13
// with 8.  0 = neither layout fits.
static int ls_pick_waves(int64_t N, int64_t B) {
    const int force_w = (int)knob(KN_LS_WAVES, 0);   // dev knob
    // (which graphs the fused kernel takes, and from where two 4-wave tiles share a CU, is decided on the layout WITH rd_std's own
    // 4N bytes, as until round 6: the compact layout now keeps rd_std inside the proposal tile and launches with less, but the
    // rows that gains -- N ~ 6500 .. 9000, one tile per CU -- are rows the round kernels win on (envs/env_L2A.py))
    const size_t sd = (size_t)((N + 3) & ~3ll) * 4;
    auto fit = [&](int w) { return ls_lds_bytes(N, w) + (w == kLsMergeWaves ? sd : 0); };
    const bool two_small = 2 * fit(4) <= (size_t)kLdsBytes;
    int W = force_w == 4 || force_w == 8 ? force_w : (ceil_div(B, kWave) <= (int64_t)num_cus() || !two_small ? 8 : 4);
    if (W == 8 && fit(8) > (size_t)kLdsBytes) W = 4;
    return fit(W) <= (size_t)kLdsBytes ? W : 0;
}

// ws rows start ws_pitch ENTRIES apart (0 = N): any N works once the pitch is a multiple of 16 bytes
static bool ls_pitch_ok(const void* ws, int64_t pitch, int32_t ws_bytes, int64_t N) {
    return pitch >= N && ((pitch * ws_bytes) & 15) == 0 && (((uintptr_t)ws) & 15) == 0;
}

extern "C" int rls_maxcut_local_search_supported(const rls_graph* g, int64_t B, int32_t num_spin) {
    if (!g || g->num_nodes <= 0) return 0;
    const int64_t N = g->num_nodes;
    if (num_spin < 0 || num_spin + 1 > kTopCap || num_spin >= N) return 0;
    if (g->wgt || g->max_degree >= kRingMaxRun) return 0;
    if (pick_planes(g->num_stored_edges) == 0) return 0;
    if ((N & 3) != 0) return 0;   // rows that are not dword-aligned: the round kernels (any N, and faster there than an element-wise fused form was)
    return ls_pick_waves(N, B > 0 ? B : 1) != 0;
}

extern "C" int rls_maxcut_local_search(const rls_graph* g, uint8_t* x, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                                       const float* rd_std, const float* noise, uint64_t seed, int64_t env_offset,
                                       int32_t num_iters, int32_t num_spin, int32_t first_draw_proposes, int64_t* obj,
                                       int32_t compute_obj, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0 && num_iters >= 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x && ws && rd_std && obj, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(ws_bytes == 1 || ws_bytes == 2, RLS_EINVAL, "ws_bytes must be 1 or 2 (rls_maxcut_ls_weights writes either)");
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    RLS_REQUIRE(num_spin >= 0 && num_spin + 1 <= kTopCap && num_spin < N, RLS_EUNSUPPORTED,
                "num_spin=%d outside [0, %d] (and < N)", num_spin, kTopCap - 1);
    RLS_REQUIRE(!g->wgt && g->max_degree < kRingMaxRun, RLS_EUNSUPPORTED,
                "fused local search needs an unweighted graph with max degree < %d", kRingMaxRun);
    // x and noise: rows that start 4-byte aligned on 16-byte bases (N % 4 == 0).  ws is read in 16-byte pieces, the last piece
    // of a row included: its rows sit ws_pitch entries apart, a multiple of 16 bytes (rls_maxcut_ls_weights writes that layout),
    // so no piece leaves its row -- an exactly sized [B, N] array whose rows are not 16-byte multiples is refused, not over-read
    if (ws_pitch == 0) ws_pitch = N;
    RLS_REQUIRE(ls_pitch_ok(ws, ws_pitch, ws_bytes, N), RLS_EUNSUPPORTED,
                "fused local search reads ws in 16-byte pieces: rows must sit a multiple of 16 bytes apart on a 16-byte base "
                "(ws_pitch=%lld entries of %d bytes, N=%lld)", (long long)ws_pitch, (int)ws_bytes, (long long)N);
    const bool aligned = tile_rows_aligned(x, N, 1) && (((uintptr_t)noise) & 15) == 0;
    RLS_REQUIRE(aligned, RLS_EUNSUPPORTED, "fused local search needs rows of x and noise that start 4-byte aligned on 16-byte bases "
                "(N=%lld): rls_maxcut_ls_threshold / rls_maxcut_ls_rounds take any layout", (long long)N);
    int W = ls_pick_waves(N, B);
    RLS_REQUIRE(W != 0, RLS_EUNSUPPORTED, "N=%lld needs %zu B of LDS (max %d)", (long long)N, ls_lds_bytes(N, 4), kLdsBytes);
    const size_t lds = ls_lds_bytes(N, W);
    const int P = pick_planes(E);
    RLS_REQUIRE(P != 0, RLS_EUNSUPPORTED, "E'=%lld too large", (long long)E);
    const dim3 grid((unsigned)ceil_div(B, kWave)), block(W * kWave);
    hipStream_t s = as_stream(stream);
    const int halve = g->if_bidirectional ? 1 : 0;
    const bool no_levels = knob_on(KN_SWEEP_NO_LEVELS);   // dev knob
    const bool levels = !no_levels && g->sweep_lv_ptr && g->sweep_lv_data && g->num_sweep_groups > 0 &&
                        g->num_sweep_groups <= N;
    const int batched = levels ? 2 : (g->sweep_rowptr != nullptr && g->sweep_stream != nullptr ? 1 : 0);
    // sweep schedule: level groups (lane = node) | level schedule stream (lane = env) | the CSR as it is
    const int32_t* rp_src = batched == 2 ? g->sweep_lv_ptr : (batched ? g->sweep_rowptr : g->rowptr);
    const int32_t* sw_src = batched == 2 ? g->sweep_lv_data : (batched ? g->sweep_stream : g->col);
    const int64_t sw_len = batched == 2 ? g->num_sweep_groups : (batched ? g->nnz + N : g->nnz);
#define LAUNCH_LSF(AL, WT, PP)                                                                                       \
    do {                                                                                                             \
        auto kern = W == 8 ? k_maxcut_local_search<AL, WT, PP, 8> : k_maxcut_local_search<AL, WT, PP, 4>;             \
        if (lds > 64 * 1024)                                                                                         \
            ensure_dyn_lds((const void*)kern, lds);      \
        hipLaunchKernelGGL(kern, grid, block, lds, s, x, B, N, g->eu, g->ev, E, halve, rp_src, sw_src, sw_len,        \
                           (const WT*)ws, ws_pitch, rd_std, noise, seed, env_offset, (int)num_iters, (int)num_spin,   \
                           (int)first_draw_proposes, obj, (int)compute_obj, batched);                                 \
    } while (0)
    // two counter widths (the 12- and 20-plane forms of the edge counter save a few carry steps per 1024 edges: not worth
    // a second pair of 190 KB kernels each); rows that are not 16-byte multiples always take the 4-wave layout
#define DISPATCH_P(AL, WT)                        \
    switch (P) {                                  \
        case 12:                                  \
        case 16: LAUNCH_LSF(AL, WT, 16); break;   \
        default: LAUNCH_LSF(AL, WT, 24); break;   \
    }
    // (an element-wise form for rows that are not dword-aligned existed until round 3: 180 - 300 KB per instantiation, and slower
    // there than the round kernels, which read the weights on a padded pitch)
    if (ws_bytes == 1) { DISPATCH_P(true, int8_t) } else { DISPATCH_P(true, int16_t) }
#undef DISPATCH_P
#undef LAUNCH_LSF
    return check_launch("k_maxcut_local_search");
}

// 1 when rls_maxcut_ls_threshold / rls_maxcut_ls_propose cover this graph (a tile that fits LDS; ws rows on a 16-byte pitch)
extern "C" int rls_maxcut_ls_rounds_supported(const rls_graph* g, int32_t num_spin) {
    if (!g || g->num_nodes <= 0) return 0;
    const int64_t N = g->num_nodes;
    if (num_spin < 0 || num_spin + 1 > kTopCap || num_spin >= N) return 0;
    if (pick_planes(g->num_stored_edges) == 0) return 0;
    // (the bare 64-env tile and the half tile need the scratch buffer: the rounds run through the mask words there)
    return ls_propose_lds(N, false) <= (size_t)kLdsBytes || ((ls_big_tile(N) || ls_half_tile(N) || ls_narrow_tile(N)) && ls_noise_passes_fit(N));
}

// bytes of caller-provided scratch with which the two entry points below split a tile's noise pass over several workgroups
// (0: the batch alone fills the chip, or the rows are too short to split); without it they run one workgroup per tile
extern "C" int64_t rls_maxcut_ls_scratch_bytes(const rls_graph* g, int64_t B, int32_t ws_bytes, int32_t num_draws) {
    if (!g || g->num_nodes <= 0 || B <= 0 || (ws_bytes != 1 && ws_bytes != 2)) return 0;
    const int64_t N = g->num_nodes;
    const int64_t nch = ws_bytes == 1 ? ls_num_chunks<int8_t>(N) : ls_num_chunks<int16_t>(N);
    const size_t need = ls_scratch_bytes(B, N, ls_slices(B, nch), num_draws);
    if (ls_big_tile(N) || ls_half_tile(N) || ls_narrow_tile(N)) {   // the mask words are how the rounds run at all here: at least one round's
        const size_t one = (size_t)ceil_div(B, kWave) * (size_t)N * 8;
        return (int64_t)(need > one ? need : one);
    }
    return (int64_t)need;
}

// workgroups per tile the noise passes of a batch of B envs are split over when the scratch buffer is there (1: the tiles fill
// the chip by themselves)
extern "C" int rls_maxcut_ls_slices(const rls_graph* g, int64_t B, int32_t ws_bytes) {
    if (!g || g->num_nodes <= 0 || B <= 0 || (ws_bytes != 1 && ws_bytes != 2)) return 1;
    return ls_slices(B, ws_bytes == 1 ? ls_num_chunks<int8_t>(g->num_nodes) : ls_num_chunks<int16_t>(g->num_nodes));
}


extern "C" int rls_maxcut_ls_threshold(const rls_graph* g, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                                       const float* rd_std, uint64_t seed, int64_t env_offset, int32_t draw, int32_t num_spin,
                                       float* thresh, void* scratch, int64_t scratch_bytes, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0 && draw >= 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(ws && rd_std && thresh, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(ws_bytes == 1 || ws_bytes == 2, RLS_EINVAL, "ws_bytes must be 1 or 2 (rls_maxcut_ls_weights writes either)");
    const int64_t N = g->num_nodes;
    RLS_REQUIRE(num_spin >= 0 && num_spin + 1 <= kTopCap && num_spin < N, RLS_EUNSUPPORTED, "num_spin=%d outside [0, %d] (and < N)",
                num_spin, kTopCap - 1);
    if (ws_pitch == 0) ws_pitch = N;
    RLS_REQUIRE(ls_pitch_ok(ws, ws_pitch, ws_bytes, N), RLS_EUNSUPPORTED,
                "ws rows must start 16-byte aligned: pitch %lld entries of %d bytes (N=%lld)", (long long)ws_pitch, (int)ws_bytes, (long long)N);
    const bool sd_lds = ls_noise_sd_lds(N);
    RLS_REQUIRE(ls_noise_passes_fit(N), RLS_EUNSUPPORTED, "N=%lld: rd_std does not fit LDS and N is not a multiple of 4", (long long)N);
    const size_t lds = ls_threshold_lds(N, sd_lds);
    int S = ls_slices(B, ws_bytes == 1 ? ls_num_chunks<int8_t>(N) : ls_num_chunks<int16_t>(N));
    if (!scratch || (size_t)scratch_bytes < ls_scratch_bytes(B, N, S) || (((uintptr_t)scratch) & 15) != 0) S = 1;
    const dim3 grid((unsigned)ceil_div(B, kWave), (unsigned)S), block(kLsRoundWaves * kWave);
    hipStream_t s = as_stream(stream);
#define LAUNCH_TH(WT)                                                                                                  \
    do {                                                                                                               \
        auto kern = sd_lds ? k_ls_threshold<WT, true> : k_ls_threshold<WT, false>;                                     \
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, (const WT*)ws, ws_pitch, B, N, rd_std, seed, env_offset, (int)draw, (int)num_spin, thresh, \
                           (float*)scratch);                                                                           \
    } while (0)
    if (ws_bytes == 1) LAUNCH_TH(int8_t); else LAUNCH_TH(int16_t);
#undef LAUNCH_TH
    if (int rc = check_launch("k_ls_threshold")) return rc;
    if (S > 1) {
        hipLaunchKernelGGL(k_ls_threshold_merge, dim3(grid.x), dim3(kWave), 0, s, (const float*)scratch, S, B, (int)num_spin, thresh);
        return check_launch("k_ls_threshold_merge");
    }
    return RLS_OK;
}

extern "C" int rls_maxcut_ls_rounds(const rls_graph* g, uint8_t* x, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                                    const float* rd_std, const float* thresh, uint64_t seed, int64_t env_offset, int32_t first_draw,
                                    int32_t num_draws, int64_t* obj, void* scratch, int64_t scratch_bytes, void* stream);

extern "C" int rls_maxcut_ls_propose(const rls_graph* g, uint8_t* x, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                                     const float* rd_std, const float* thresh, uint64_t seed, int64_t env_offset, int32_t draw,
                                     int64_t* obj, void* scratch, int64_t scratch_bytes, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0 && draw >= 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x && ws && rd_std && thresh && obj, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(ws_bytes == 1 || ws_bytes == 2, RLS_EINVAL, "ws_bytes must be 1 or 2 (rls_maxcut_ls_weights writes either)");
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    if (ws_pitch == 0) ws_pitch = N;
    RLS_REQUIRE(ls_pitch_ok(ws, ws_pitch, ws_bytes, N), RLS_EUNSUPPORTED,
                "ws rows must start 16-byte aligned: pitch %lld entries of %d bytes (N=%lld)", (long long)ws_pitch, (int)ws_bytes, (long long)N);
    const int x_aligned = tile_rows_aligned(x, N, 1) ? 1 : 0;   // else the funnel-shift form of the row-piece stage
    if (ls_propose_lds(N, false) > (size_t)kLdsBytes && (ls_big_tile(N) || ls_half_tile(N) || ls_narrow_tile(N)))   // mask kernel + the apply kernel (needs the scratch)
        return rls_maxcut_ls_rounds(g, x, B, ws, ws_bytes, ws_pitch, rd_std, thresh, seed, env_offset, draw, 1, obj, scratch, scratch_bytes, stream);
    RLS_REQUIRE(ls_propose_lds(N, false) <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld needs %zu B of LDS (max %d)", (long long)N,
                ls_propose_lds(N, false), kLdsBytes);
    const int P = pick_planes(E);
    RLS_REQUIRE(P != 0, RLS_EUNSUPPORTED, "E'=%lld too large", (long long)E);
    int S = ls_slices(B, ws_bytes == 1 ? ls_num_chunks<int8_t>(N) : ls_num_chunks<int16_t>(N));
    if (!scratch || (size_t)scratch_bytes < ls_scratch_bytes(B, N, S) || (((uintptr_t)scratch) & 15) != 0) S = 1;
    const dim3 grid((unsigned)ceil_div(B, kWave)), block(kLsRoundWaves * kWave);
    hipStream_t s = as_stream(stream);
    const int halve = g->if_bidirectional ? 1 : 0;
    if (S > 1) {   // the mask words first, S workgroups per tile
        const size_t ldm = ls_mask_lds(N);
        const dim3 gm(grid.x, (unsigned)S);
#define LAUNCH_MK(WT)                                                                                                  \
    do {                                                                                                               \
        auto kern = k_ls_mask<WT>;                                                                                     \
        if (ldm > 64 * 1024) ensure_dyn_lds((const void*)kern, ldm); \
        hipLaunchKernelGGL(kern, gm, block, ldm, s, (const WT*)ws, ws_pitch, B, N, rd_std, thresh, seed, env_offset, (int)draw, (uint64_t*)scratch, (int64_t)0); \
    } while (0)
        if (ws_bytes == 1) LAUNCH_MK(int8_t); else LAUNCH_MK(int16_t);
#undef LAUNCH_MK
        if (int rc = check_launch("k_ls_mask")) return rc;
    }
    const bool sd_lds = S == 1 && !ls_sd_global() && ls_propose_lds(N, true) <= (size_t)kLdsBytes;
    const size_t lds = ls_propose_lds(N, sd_lds);
#define LAUNCH_PR(WT, PP, SD, PM)                                                                                      \
    do {                                                                                                               \
        auto kern = k_ls_propose<WT, PP, SD, PM>;                                                                      \
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, x, B, N, g->eu, g->ev, E, halve, (const WT*)ws, ws_pitch, rd_std, thresh, seed,  \
                           env_offset, (int)draw, obj, (const uint64_t*)scratch, x_aligned);                           \
    } while (0)
    // (one counter width: the round is bound by its noise generation, the few carry steps a narrower counter saves per 1024
    // edges do not show, and every instantiation is ~40 KB)
#define DISPATCH_PR(WT)                                                                            \
    do {                                                                                           \
        if (sd_lds) LAUNCH_PR(WT, 24, true, false); else LAUNCH_PR(WT, 24, false, false);          \
    } while (0)
    if (S > 1) LAUNCH_PR(int8_t, 24, false, true);   // (WT and SD_LDS play no part once the mask is given)
    else if (ws_bytes == 1) DISPATCH_PR(int8_t); else DISPATCH_PR(int16_t);
#undef DISPATCH_PR
#undef LAUNCH_PR
    return check_launch("k_ls_propose");
}

// num_draws proposal rounds (draws first_draw .. first_draw + num_draws - 1) in place.  A small batch with enough scratch for all
// rounds' mask words (rls_maxcut_ls_scratch_bytes(.., num_draws)) gets them from num_draws mask launches and applies them on one
// load of the tile; otherwise one rls_maxcut_ls_propose per round.  Same result either way.
extern "C" int rls_maxcut_ls_rounds(const rls_graph* g, uint8_t* x, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                                    const float* rd_std, const float* thresh, uint64_t seed, int64_t env_offset, int32_t first_draw,
                                    int32_t num_draws, int64_t* obj, void* scratch, int64_t scratch_bytes, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0 && first_draw >= 0 && num_draws >= 0, RLS_EINVAL, "bad sizes");
    if (B == 0 || num_draws == 0) return RLS_OK;
    RLS_REQUIRE(x && ws && rd_std && thresh && obj, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(ws_bytes == 1 || ws_bytes == 2, RLS_EINVAL, "ws_bytes must be 1 or 2 (rls_maxcut_ls_weights writes either)");
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    if (ws_pitch == 0) ws_pitch = N;
    const int S = ls_slices(B, ws_bytes == 1 ? ls_num_chunks<int8_t>(N) : ls_num_chunks<int16_t>(N));
    const bool per_round = knob_on(KN_LS_PER_ROUND);   // dev knob: one propose launch per round
    const bool half = ls_half_tile(N), narrow = ls_narrow_tile(N);
    const bool big = ls_big_tile(N) || half || narrow;
    RLS_REQUIRE(!(half || narrow) || ls_noise_passes_fit(N), RLS_EUNSUPPORTED, "N=%lld: rd_std does not fit LDS and N is not a multiple of 4", (long long)N);
    const size_t one_round = (size_t)ceil_div(B, kWave) * (size_t)N * 8;
    const bool scratch_ok = scratch && (((uintptr_t)scratch) & 15) == 0 && ls_pitch_ok(ws, ws_pitch, ws_bytes, N) && pick_planes(E) != 0;
    RLS_REQUIRE(!big || (scratch_ok && (size_t)scratch_bytes >= one_round), RLS_EUNSUPPORTED,
                "N=%lld: the proposal rounds need %zu bytes of scratch (rls_maxcut_ls_scratch_bytes) and ws rows on a 16-byte pitch",
                (long long)N, one_round);
    const bool all_at_once = !per_round && num_draws > 1 && scratch_ok && (size_t)scratch_bytes >= one_round * (size_t)num_draws &&
                             (big || ls_propose_lds(N, false) <= (size_t)kLdsBytes);
    if (!all_at_once && !big) {
        for (int32_t r = 0; r < num_draws; ++r)
            if (int rc = rls_maxcut_ls_propose(g, x, B, ws, ws_bytes, ws_pitch, rd_std, thresh, seed, env_offset, first_draw + r, obj, scratch,
                                               scratch_bytes, stream))
                return rc;
        return RLS_OK;
    }
    const dim3 grid((unsigned)ceil_div(B, kWave)), gm(grid.x, (unsigned)S);
    hipStream_t s = as_stream(stream);
    const bool sd_lds = ls_noise_sd_lds(N);
    const size_t ldm = ls_mask_lds(N, sd_lds);
    const int halve = g->if_bidirectional ? 1 : 0, x_aligned = tile_rows_aligned(x, N, 1) ? 1 : 0;
    const int per_launch = all_at_once ? num_draws : 1;       // rounds whose mask words are in the scratch at once
    for (int32_t r0 = 0; r0 < num_draws; r0 += per_launch) {
        {   // the mask words of the per_launch rounds: one launch, blockIdx.z = round
            const dim3 block(kLsRoundWaves * kWave), gz(gm.x, gm.y, (unsigned)per_launch);
            const int64_t round_words = (int64_t)grid.x * N;
            if (ws_bytes == 1) {
                auto kern = sd_lds ? k_ls_mask<int8_t, true> : k_ls_mask<int8_t, false>;
                if (ldm > 64 * 1024) ensure_dyn_lds((const void*)kern, ldm);
                hipLaunchKernelGGL(kern, gz, block, ldm, s, (const int8_t*)ws, ws_pitch, B, N, rd_std, thresh, seed, env_offset,
                                   (int)(first_draw + r0), (uint64_t*)scratch, round_words);
            } else {
                auto kern = sd_lds ? k_ls_mask<int16_t, true> : k_ls_mask<int16_t, false>;
                if (ldm > 64 * 1024) ensure_dyn_lds((const void*)kern, ldm);
                hipLaunchKernelGGL(kern, gz, block, ldm, s, (const int16_t*)ws, ws_pitch, B, N, rd_std, thresh, seed, env_offset,
                                   (int)(first_draw + r0), (uint64_t*)scratch, round_words);
            }
        }
        if (int rc = check_launch("k_ls_mask")) return rc;
        if (narrow) {   // past the half tile: each round's mask words are K6's bit-packed mask (narrow tiles there: rls_maxcut.hip)
            for (int32_t r = 0; r < per_launch; ++r)
                if (int rc = rls_maxcut_propose_accept(g, x, B, (const uint64_t*)scratch + (size_t)r * grid.x * (size_t)N, 1, obj, stream)) return rc;
            continue;
        }
        // half tiles (twice the workgroups, the same mask words): past the 64-env tile, and for batches that leave half the CUs
        // without a 64-env tile (whole local_search_inplace calls, 4096 envs: G22-sized 0.324 -> 0.305 ms, BA n = 10^4 0.893 -> 0.816,
        // G70-sized 0.689 -> 0.642; at 16 384 envs no gain).  Dev knob RLS_LS_APPLY32 = 0 | 1 forces the choice.
        const int knob32 = (int)knob(KN_LS_APPLY32, -1);
        const bool fast32 = x_aligned && (N & 7) == 0 && ls_apply32_lds(N, kLsRoundWaves, true) <= (size_t)kLdsBytes;
        const bool few32 = knob32 >= 0 ? knob32 != 0 : 2 * (int64_t)grid.x <= (int64_t)num_cus();
        if (half || (few32 && fast32)) {
            const int st32 = (x_aligned && (N & 7) == 0 && ls_apply32_lds(N, kLsRoundWaves, true) <= (size_t)kLdsBytes) ? 1 : 0;
            const size_t lds = ls_apply32_lds(N, kLsRoundWaves, st32 != 0);
            auto kern = k_ls_apply_rounds32<24, kLsRoundWaves>;
            if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
            hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(B, (int64_t)kHalf)), dim3(kLsRoundWaves * kWave), lds, s, x, B, N, g->eu, g->ev, E,
                               halve, (const uint64_t*)scratch, (int64_t)grid.x, per_launch, obj, x_aligned, st32);
        } else if (big) {   // the bare tile: 4 waves, lane-per-env loads and stores
            const size_t lds = ls_apply_lds(N, 4, false);
            auto kern = k_ls_apply_rounds<24, 4>;
            if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
            hipLaunchKernelGGL(kern, grid, dim3(4 * kWave), lds, s, x, B, N, g->eu, g->ev, E, halve, (const uint64_t*)scratch, per_launch, obj,
                               x_aligned, 0);
        } else {
            const size_t lds = ls_apply_lds(N, kLsRoundWaves, true);
            auto kern = k_ls_apply_rounds<24, kLsRoundWaves>;
            if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
            hipLaunchKernelGGL(kern, grid, dim3(kLsRoundWaves * kWave), lds, s, x, B, N, g->eu, g->ev, E, halve, (const uint64_t*)scratch,
                               per_launch, obj, x_aligned, 1);
        }
        if (int rc = check_launch("k_ls_apply_rounds")) return rc;
    }
    return RLS_OK;
}

extern "C" int rls_maxcut_ls_normals(float* out, int64_t B, int64_t N, uint64_t seed, int64_t env_offset, int32_t draw, void* stream) {
    RLS_REQUIRE(B >= 0 && N > 0 && draw >= 0 && env_offset >= 0, RLS_EINVAL, "bad sizes B=%lld N=%lld draw=%d", (long long)B, (long long)N, (int)draw);
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(out, RLS_EINVAL, "out is NULL");
    hipLaunchKernelGGL(k_ls_normals, dim3((unsigned)grid_for(B * ((N + 3) >> 2), 256)), dim3(256), 0, as_stream(stream), out, B, N, seed,
                       env_offset, (int)draw);
    return check_launch("k_ls_normals");
}
