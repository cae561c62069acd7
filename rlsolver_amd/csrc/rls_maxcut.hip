// MaxCut environment kernels for gfx950 (MI355X).  See include/rlsolver_hip.h for the
// reference call site each entry point replaces, DESIGN.md for layouts and rooflines.
#include "rls_cutcount.h"
#include "rls_sweep.h"
#include "rls_tile32.h"
#include <cstdlib>

namespace rls {

constexpr int kTileWaves = 4;   // waves cooperating on one 64-env tile (one per SIMD)
constexpr int kTileWavesMax = 8;   // K1 / K6 take 8 when the tile is so large that only one workgroup fits a CU
// waves per tile for a bit tile of N nodes: with one workgroup per CU, 4 waves cannot keep enough loads in flight
static inline int tile_waves_for(int64_t N) { return (size_t)N * 8 + 4 * 4096 + 4096 > 80 * 1024 ? kTileWavesMax : kTileWaves; }

// K1.  One workgroup = one 64-env tile, W waves (4, or 8 for tiles that leave one workgroup per CU): they share
// the tile load (rls_tile.h: row pieces -> corner turn -> pack -> 64x64 bit transpose; wave w takes every W-th
// 64-node chunk) and the edge blocks of the bit-sliced count (rls_cutcount.h).
template <typename T, bool VEC, int P, int W>
__global__ __launch_bounds__(W * kWave) void k_maxcut_obj(const T* __restrict__ x, int64_t B, int64_t N,
                                                                   const int32_t* __restrict__ eu,
                                                                   const int32_t* __restrict__ ev, int64_t E,
                                                                   int halve, int64_t* __restrict__ obj,
                                                                   int stage_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(words + N);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    unsigned char* stage = stage_off >= 0 ? smem + stage_off + w * kStageBytes : nullptr;
#ifdef RLS_PROF
    const unsigned long long t0 = wall_clock64();
#endif
    tile_load_bits<T, VEC>(x, B, N, b0, words, lane, w, W, stage);
#ifdef RLS_PROF
    const unsigned long long t1 = wall_clock64();
#endif
    __syncthreads();
#ifdef RLS_PROF
    const unsigned long long t2 = wall_clock64();
#endif
    const int64_t part = tile_cut_count<P>(words, eu, ev, E, lane, w, W);
#ifdef RLS_PROF
    const unsigned long long t3 = wall_clock64();
#endif
    int64_t total = block_sum_partials<W>(part, scratch, lane, w);
    if (halve) total >>= 1;  // values // 2, env_L2A.py:65 (count is even and >= 0)
    if (w == 0 && b0 + lane < B) obj[b0 + lane] = total;
#ifdef RLS_PROF
    if (lane == 0) {   // dev build only: phase durations summed over waves, first start / last end of the launch
        const unsigned long long t4 = wall_clock64();
        const int64_t wid = (int64_t)blockIdx.x * W + w;
        if (wid < 65536) {
            unsigned long long* q = g_prof_w + wid * 5;
            q[0] = t0; q[1] = t1; q[2] = t2; q[3] = t3; q[4] = t4;
        }
    }
#endif
}

// K1 on HALF tiles (rls_tile32.h): one workgroup = 32 envs, words of 32 bits -- N * 4 bytes of LDS, so graphs past the 64-env tile
// (20 224 < N <= ~40 000) keep the tile form, 2 x the edge-list reads per env instead of the one-env-per-wave kernel's byte gathers.
template <typename T, bool VEC, int P, int W>
__global__ __launch_bounds__(W * kWave) void k_maxcut_obj32(const T* __restrict__ x, int64_t B, int64_t N,
                                                           const int32_t* __restrict__ eu, const int32_t* __restrict__ ev, int64_t E,
                                                           int halve, int64_t* __restrict__ obj, int stage_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* words32 = reinterpret_cast<uint32_t*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + (((size_t)N * 4 + 15) & ~(size_t)15));
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kHalf;
    unsigned char* stage = stage_off >= 0 ? smem + stage_off + w * kStageBytes : nullptr;
    tile32_load_bits<T, VEC>(x, B, N, b0, words32, lane, w, W, stage);
    __syncthreads();
    const int64_t part = tile32_cut_count<P>(words32, eu, ev, E, lane, w, W);
    int64_t total = block_sum_partials<W>(part, scratch, lane, w);
    if (halve) total >>= 1;
    if (w == 0 && lane < kHalf && b0 + lane < B) obj[b0 + lane] = total;
}

// K6: proposal = x ^ mask for 64 envs; accept the row when its cut is >= the incumbent.
template <bool VEC, int P, int W, bool MASK_BITS = false>
__global__ __launch_bounds__(W * kWave) void k_maxcut_propose_accept(uint8_t* __restrict__ x,
                                                                              const uint8_t* __restrict__ mask,
                                                                              int64_t B, int64_t N,
                                                                              const int32_t* __restrict__ eu,
                                                                              const int32_t* __restrict__ ev,
                                                                              int64_t E, int halve,
                                                                              int64_t* __restrict__ obj, int stage_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(words + N);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    unsigned char* stage = stage_off >= 0 ? smem + stage_off + w * kStageBytes : nullptr;
    // proposal = x ^ mask, built in ONE bit tile: the mask pass XORs into the words the x pass wrote (same lane of
    // the same wave owns a word in both passes); a second tile halved the workgroups per CU for N >= 5000
    tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane, w, W, stage);
    if constexpr (MASK_BITS) {
        // the mask arrives as the tile it is going to be: word (t, n) of uint64 [ceil(B / 64), N] -- N / 8 bytes per env instead
        // of N, no byte -> bit pass (with the byte mask K6 moved 1.36x its algorithmic bytes at N = 10^4)
        __syncthreads();
        const uint64_t* mw = reinterpret_cast<const uint64_t*>(mask) + (int64_t)blockIdx.x * N;
        for (int64_t n = threadIdx.x; n < N; n += W * kWave) words[n] ^= mw[n];
    } else {
        tile_load_bits<uint8_t, VEC, kStageDepth, true>(mask, B, N, b0, words, lane, w, W, stage);
    }
    __syncthreads();
    int64_t total = block_sum_partials<W>(tile_cut_count<P>(words, eu, ev, E, lane, w, W), scratch, lane, w);
    if (halve) total >>= 1;
    const int64_t b = b0 + lane;
    const bool accept = (b < B) && (total >= obj[b]);  // vs1.ge(vs0), util_read_data.py:199
    __syncthreads();                                   // every wave has read obj[b] before wave 0 updates it
    if (accept && w == 0) obj[b] = total;
    // accepted rows take the proposal (each wave writes a quarter of the columns); others are untouched
    tile_store_bytes<VEC>(x, B, N, b0, words, lane, w, W, accept, stage);
}

// K6 on half tiles (graphs past the 64-env tile, N <= ~40 000): the same steps on 32-bit words.  A bit-packed mask stays
// uint64 [ceil(B / 64), N]: half tile h takes the low (even h) or high dword of word (h / 2, n).
template <bool VEC, int P, int W, bool MASK_BITS>
__global__ __launch_bounds__(W * kWave) void k_maxcut_propose_accept32(uint8_t* __restrict__ x, const uint8_t* __restrict__ mask,
                                                                      int64_t B, int64_t N, const int32_t* __restrict__ eu,
                                                                      const int32_t* __restrict__ ev, int64_t E, int halve,
                                                                      int64_t* __restrict__ obj, int stage_off) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* words32 = reinterpret_cast<uint32_t*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + (((size_t)N * 4 + 15) & ~(size_t)15));
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kHalf;
    unsigned char* stage = stage_off >= 0 ? smem + stage_off + w * kStageBytes : nullptr;
    tile32_load_bits<uint8_t, VEC>(x, B, N, b0, words32, lane, w, W, stage);
    if constexpr (MASK_BITS) {
        __syncthreads();
        const uint32_t* mw = reinterpret_cast<const uint32_t*>(mask) + ((int64_t)(blockIdx.x >> 1) * N) * 2 + (blockIdx.x & 1);
        for (int64_t n = threadIdx.x; n < N; n += W * kWave) words32[n] ^= mw[n * 2];
    } else {
        tile32_load_bits<uint8_t, VEC, true>(mask, B, N, b0, words32, lane, w, W, stage);
    }
    __syncthreads();
    int64_t total = block_sum_partials<W>(tile32_cut_count<P>(words32, eu, ev, E, lane, w, W), scratch, lane, w);
    if (halve) total >>= 1;
    const int64_t b = b0 + (lane & (kHalf - 1));
    const bool accept = (b < B) && (total >= obj[b]);     // (block_sum_partials leaves lanes 32..63 with 0: they hold their env's verdict below)
    const bool acc_env = (bool)((ballot64(accept && lane < kHalf) >> (lane & (kHalf - 1))) & 1ull);
    __syncthreads();                                      // every wave has read obj[b] before wave 0 updates it
    if (acc_env && w == 0 && lane < kHalf) obj[b] = total;
    tile32_store_bytes<VEC>(x, B, N, b0, words32, lane, w, W, acc_env, stage);
}

// =====================================================================================
// K5: greedy single-flip sweep.  Three forms, fastest first (the launcher picks; all bit-identical):
//   k_maxcut_greedy_sweep_levels   lane = node, one pass per dependency level (rls_sweep.h: sweep_tile_levels)
//   k_maxcut_greedy_sweep_batched  lane = env on the level schedule, one node per wave step (degrees >= 256)
//   k_maxcut_greedy_sweep          lane = env, one wave, strictly sequential (no schedule), described here:
// One lane = one env, 64 envs per wave, state as a bit tile in LDS.
// For node i (sequential, as the reference's semantics demand) every lane counts the set spins
// among i's neighbours in ITS env: the neighbour id is a broadcast LDS read from a ring of CSR
// `col` entries, the neighbour's word a second broadcast read, the lane's bit a v_bfe.  Node i is
// flipped in the envs whose gain is >= 0 (ties accept); the flip mask is one ballot.
//
// Nothing on the per-node path touches global memory: rowptr is staged whole in LDS, `col` streams
// through a 4096-entry LDS ring filled by direct global->LDS loads a quarter ring at a time, at
// least half a ring ahead of the node being processed (first version: one L2 round trip per node
// = 1.5 us/node, 3 ms per G22 sweep).
// =====================================================================================
constexpr int kSweepMaxDeg = kRingMaxRun;

template <bool VEC>
__global__ __launch_bounds__(kWave) void k_maxcut_greedy_sweep(uint8_t* __restrict__ x, int64_t B, int64_t N,
                                                               const int32_t* __restrict__ rowptr,
                                                               const int32_t* __restrict__ col, int64_t nnz,
                                                               int64_t* __restrict__ obj) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int32_t* rp = reinterpret_cast<int32_t*>(smem + (size_t)(N + 2) * 8);
    int32_t* ring = rp + ((N + 1 + 3) & ~3ll);
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    if (lane == 0) words[N] = 0;   // sentinel word
    for (int64_t i = lane; i <= N; i += kWave) rp[i] = rowptr[i];
    // the ring is idle before and after the sweep: it doubles as the row-piece stage of the tile load / store
    unsigned char* stage = reinterpret_cast<unsigned char*>(ring);
    tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane, 0, 1, stage);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int64_t gain = sweep_tile(words, rp, ring, col, nnz, N, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    tile_store_bytes<VEC>(x, B, N, b0, words, lane, 0, 1, true, stage);
    if (b0 + lane < B) obj[b0 + lane] += gain;
}

// SW-wave variant over the host-built level schedule (rls_sweep.h: sweep_tile_batched).  A node costs one
// wave ~0.9 us of dependent LDS round trips whatever the schedule, so the tile's time is (N / SW) x that:
// 16 waves when there is at most one tile per CU, otherwise 8 (G22, B = 2^16: 4 waves 1.40, 8 waves 1.68,
// 16 waves 1.31 x 10^11 candidate flips/s).
constexpr int kSweepLoadWaves = 4;   // waves that move the tile (the ring holds 4 row-piece stages)

template <bool VEC, int SW>
__global__ __launch_bounds__(SW * kWave) void k_maxcut_greedy_sweep_batched(
    uint8_t* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ rowptr_flagged,
    const int32_t* __restrict__ stream, int64_t stream_len, int64_t* __restrict__ obj) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int32_t* rp = reinterpret_cast<int32_t*>(smem + (size_t)(N + 2) * 8);
    int32_t* ring = rp + ((N + 1 + 3) & ~3ll);
    int64_t* scratch = reinterpret_cast<int64_t*>(ring + kRing);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    if (threadIdx.x == 0) words[N] = 0;
    for (int64_t i = threadIdx.x; i <= N; i += SW * kWave) rp[i] = rowptr_flagged[i];
    static_assert(kRing * 4 >= kSweepLoadWaves * kStageBytes, "the ring doubles as the tile stage");
    unsigned char* stage = reinterpret_cast<unsigned char*>(ring) + (w % kSweepLoadWaves) * kStageBytes;
    if (w < kSweepLoadWaves) tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane, w, kSweepLoadWaves, stage);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int64_t part = sweep_tile_batched<SW>(words, rp, ring, stream, stream_len, N, lane, w);
    const int64_t gain = block_sum_partials<SW>(part, scratch, lane, w);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (w < kSweepLoadWaves) tile_store_bytes<VEC>(x, B, N, b0, words, lane, w, kSweepLoadWaves, true, stage);
    if (w == 0 && b0 + lane < B) obj[b0 + lane] += gain;
}

// Level-parallel variant (rls_sweep.h: sweep_tile_levels): lane = node, one pass per dependency level.
// obj = cut(after) from the bit-sliced counter on the resident tile.
template <bool VEC, int SW, int P>
__global__ __launch_bounds__(SW * kWave) void k_maxcut_greedy_sweep_levels(
    uint8_t* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ lv_ptr,
    const int32_t* __restrict__ lv_data, int64_t G, const int32_t* __restrict__ eu, const int32_t* __restrict__ ev,
    int64_t E, int halve, int64_t* __restrict__ obj, int has_stage) {   // has_stage 0: the tile fills LDS (N ~ 20 000)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int32_t* lvp = reinterpret_cast<int32_t*>(smem + (size_t)(N + 2) * 8);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + (size_t)(N + 2) * 8 + (((size_t)(G + 1) * 4 + 15) & ~(size_t)15));
    unsigned char* stages = reinterpret_cast<unsigned char*>(scratch + SW * kWave);
    constexpr int LW = SW < kSweepLoadWaves ? SW : kSweepLoadWaves;   // waves that move the tile
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    if (threadIdx.x == 0) words[N] = 0;
    for (int64_t i = threadIdx.x; i <= G; i += SW * kWave) lvp[i] = lv_ptr[i];
    unsigned char* stage = has_stage ? stages + (w % LW) * kStageBytes : nullptr;
    if (w < LW) tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane, w, LW, stage);
    __syncthreads();
    sweep_tile_levels<SW>(words, lvp, lv_data, G, N, lane, w);
    // obj[b] must be the cut of x[b] on entry (the sweep's accept rule compares against it in the reference: the level form
    // decides by the local gain, which is the same thing exactly then) -- so on exit it is simply the cut of the new x[b],
    // counted once on the resident tile (round 2 counted before AND after and added the difference: a quarter of the
    // kernel's VALU work at G22 size, where the SQ counters show the SIMDs ~90 % busy)
    // the tile's stores go out first and drain while the cut is counted (both only read the words): G22 2^16 114.9 -> 112.9 us,
    // G70 2^17 796 -> 780
    if (w < LW) tile_store_bytes<VEC>(x, B, N, b0, words, lane, w, LW, true, stage);
    const int64_t after = block_sum_partials<SW>(tile_cut_count<P>(words, eu, ev, E, lane, w, SW), scratch, lane, w);
    if (w == 0 && b0 + lane < B) obj[b0 + lane] = halve ? (after >> 1) : after;
}

// The level-parallel sweep on HALF tiles (rls_tile32.h): graphs past the 64-env tile (N <= ~39 000 with the schedule offsets beside
// the words), the same schedule, obj = cut(after).
template <bool VEC, int SW, int P>
__global__ __launch_bounds__(SW * kWave) void k_maxcut_greedy_sweep_levels32(
    uint8_t* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ lv_ptr, const int32_t* __restrict__ lv_data, int64_t G,
    const int32_t* __restrict__ eu, const int32_t* __restrict__ ev, int64_t E, int halve, int64_t* __restrict__ obj, int has_stage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* words32 = reinterpret_cast<uint32_t*>(smem);
    const size_t wbytes = ((size_t)(N + 2) * 4 + 15) & ~(size_t)15;
    int32_t* lvp = reinterpret_cast<int32_t*>(smem + wbytes);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + wbytes + (((size_t)(G + 1) * 4 + 15) & ~(size_t)15));
    unsigned char* stages = reinterpret_cast<unsigned char*>(scratch + SW * kWave);
    constexpr int LW = SW < kSweepLoadWaves ? SW : kSweepLoadWaves;
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * kHalf;
    if (threadIdx.x == 0) words32[N] = 0;
    for (int64_t i = threadIdx.x; i <= G; i += SW * kWave) lvp[i] = lv_ptr[i];
    unsigned char* stage = has_stage ? stages + (w % LW) * kStageBytes : nullptr;
    if (w < LW) tile32_load_bits<uint8_t, VEC>(x, B, N, b0, words32, lane, w, LW, stage);
    __syncthreads();
    sweep32_tile_levels<SW>(words32, lvp, lv_data, G, N, lane, w);
    if (w < LW) tile32_store_bytes<VEC>(x, B, N, b0, words32, lane, w, LW, true, stage);   // (stores first: they drain under the count)
    const int64_t after = block_sum_partials<SW>(tile32_cut_count<P>(words32, eu, ev, E, lane, w, SW), scratch, lane, w);
    if (w == 0 && lane < kHalf && b0 + lane < B) obj[b0 + lane] = halve ? (after >> 1) : after;
}

// ---- K1 / K6 / K5 on NARROW tiles (rls_tile32.h: 16 or 8 envs per workgroup, WT = uint16_t / uint8_t words -- 2 N / N bytes of LDS):
// graphs past the half tile, N > ~40 000 up to ~80 000 / ~160 000 nodes, which ran ONE ENV PER WAVE on a byte row until round 5
// (K5 120 ms, local_search_inplace 158 ms for 4096 envs at N = 44 000 against 0.4 / 2.5 ms at N = 39 936).  Same steps, same
// schedules, same counters as the half-tile kernels; 8 waves per workgroup.
constexpr int kNarrowWaves = 8;
template <typename WT> __host__ __device__ constexpr size_t narrow_words_bytes(int64_t N) { return (((size_t)(N + 2) * sizeof(WT)) + 15) & ~(size_t)15; }

template <typename T, bool VEC, int P, typename WT>
__global__ __launch_bounds__(kNarrowWaves * kWave) void k_maxcut_obj_n(const T* __restrict__ x, int64_t B, int64_t N,
                                                                      const int32_t* __restrict__ eu, const int32_t* __restrict__ ev, int64_t E,
                                                                      int halve, int64_t* __restrict__ obj) {
    constexpr int W = kNarrowWaves, EN = narrow_tile<WT>::E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WT* words = reinterpret_cast<WT*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + narrow_words_bytes<WT>(N));
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * EN;
    tilen_load_bits<T, WT, VEC>(x, B, N, b0, words, lane, w, W);
    __syncthreads();
    int64_t total = block_sum_partials<W>(tile32_cut_count<P, WT>(words, eu, ev, E, lane, w, W), scratch, lane, w);
    if (halve) total >>= 1;
    if (w == 0 && lane < EN && b0 + lane < B) obj[b0 + lane] = total;
}

// (a bit-packed mask stays uint64 [ceil(B / 64), N]: narrow tile h takes piece h % (64 / E) of word (h / (64 / E), n))
template <bool VEC, int P, typename WT, bool MASK_BITS>
__global__ __launch_bounds__(kNarrowWaves * kWave) void k_maxcut_propose_accept_n(uint8_t* __restrict__ x, const uint8_t* __restrict__ mask,
                                                                                 int64_t B, int64_t N, const int32_t* __restrict__ eu,
                                                                                 const int32_t* __restrict__ ev, int64_t E, int halve,
                                                                                 int64_t* __restrict__ obj) {
    constexpr int W = kNarrowWaves, EN = narrow_tile<WT>::E, NB = narrow_tile<WT>::NB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WT* words = reinterpret_cast<WT*>(smem);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + narrow_words_bytes<WT>(N));
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * EN;
    tilen_load_bits<uint8_t, WT, VEC>(x, B, N, b0, words, lane, w, W);
    if constexpr (MASK_BITS) {
        __syncthreads();
        const WT* mw = reinterpret_cast<const WT*>(mask) + ((int64_t)(blockIdx.x / NB) * N) * NB + (blockIdx.x % NB);
        for (int64_t n = threadIdx.x; n < N; n += W * kWave) words[n] = (WT)(words[n] ^ mw[n * NB]);
    } else {
        tilen_load_bits<uint8_t, WT, VEC, true>(mask, B, N, b0, words, lane, w, W);
    }
    __syncthreads();
    int64_t total = block_sum_partials<W>(tile32_cut_count<P, WT>(words, eu, ev, E, lane, w, W), scratch, lane, w);
    if (halve) total >>= 1;
    const int64_t b = b0 + (lane % EN);
    const bool accept = (b < B) && (total >= obj[b]);     // (lanes >= E hold 0: they take their env's verdict below)
    const bool acc_env = (bool)((ballot64(accept && lane < EN) >> (lane % EN)) & 1ull);
    __syncthreads();                                      // every wave has read obj[b] before wave 0 updates it
    if (acc_env && w == 0 && lane < EN) obj[b] = total;
    tilen_store_bytes<WT, VEC>(x, B, N, b0, words, lane, w, W, acc_env);
}

template <bool VEC, int P, typename WT>
__global__ __launch_bounds__(kNarrowWaves * kWave) void k_maxcut_greedy_sweep_levels_n(
    uint8_t* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ lv_ptr, const int32_t* __restrict__ lv_data, int64_t G,
    const int32_t* __restrict__ eu, const int32_t* __restrict__ ev, int64_t E, int halve, int64_t* __restrict__ obj) {
    constexpr int SW = kNarrowWaves, EN = narrow_tile<WT>::E;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    WT* words = reinterpret_cast<WT*>(smem);
    const size_t wbytes = narrow_words_bytes<WT>(N);
    int32_t* lvp = reinterpret_cast<int32_t*>(smem + wbytes);
    int64_t* scratch = reinterpret_cast<int64_t*>(smem + wbytes + (((size_t)(G + 1) * 4 + 15) & ~(size_t)15));
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * EN;
    if (threadIdx.x == 0) words[N] = 0;
    for (int64_t i = threadIdx.x; i <= G; i += SW * kWave) lvp[i] = lv_ptr[i];
    tilen_load_bits<uint8_t, WT, VEC>(x, B, N, b0, words, lane, w, SW);
    __syncthreads();
    sweep32_tile_levels<SW, WT>(words, lvp, lv_data, G, N, lane, w);
    tilen_store_bytes<WT, VEC>(x, B, N, b0, words, lane, w, SW, true);
    const int64_t after = block_sum_partials<SW>(tile32_cut_count<P, WT>(words, eu, ev, E, lane, w, SW), scratch, lane, w);
    if (w == 0 && lane < EN && b0 + lane < B) obj[b0 + lane] = halve ? (after >> 1) : after;
}

// generic fallback (weighted graphs, hubs with degree > kSweepMaxDeg): one global row fetch per node
template <bool VEC, bool WEIGHTED>
__global__ __launch_bounds__(kWave) void k_maxcut_greedy_sweep_generic(uint8_t* __restrict__ x, int64_t B, int64_t N,
                                                                       const int32_t* __restrict__ rowptr,
                                                                       const int32_t* __restrict__ col,
                                                                       const int32_t* __restrict__ wgt,
                                                                       int64_t* __restrict__ obj) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(smem);
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane);
    __syncthreads();
    const int half = lane >> 5, sh = lane & 31;
    int64_t gain = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int r0 = rowptr[i], r1 = rowptr[i + 1];
        const uint32_t xi = (w32[(i << 1) + half] >> sh) & 1u;
        int same_minus_diff = 0;  // sum_j w_ij * (x_i == x_j ? +1 : -1)
        for (int base = r0; base < r1; base += kWave) {
            const int cnt = (r1 - base) < kWave ? (r1 - base) : kWave;
            const int my_nb = (lane < cnt) ? col[base + lane] : 0;
            int my_w = 1;
            if constexpr (WEIGHTED) my_w = (lane < cnt) ? wgt[base + lane] : 0;
            for (int j = 0; j < cnt; ++j) {
                const int nb = __builtin_amdgcn_readlane(my_nb, j);
                const uint32_t xn = (w32[((int64_t)nb << 1) + half] >> sh) & 1u;
                if constexpr (WEIGHTED) {
                    const int wj = __builtin_amdgcn_readlane(my_w, j);
                    same_minus_diff += (xn == xi) ? wj : -wj;
                } else {
                    same_minus_diff += (xn == xi) ? 1 : -1;
                }
            }
        }
        const bool flip = same_minus_diff >= 0;
        gain += flip ? same_minus_diff : 0;
        const uint64_t fm = ballot64(flip);
        if (lane == 0) words[i] ^= fm;  // LDS ops of one wave execute in order
        __syncthreads();                // (single-wave block: compiler fence + waitcnt only)
    }
    tile_store_bytes<VEC>(x, B, N, b0, words, lane);
    if (b0 + lane < B) obj[b0 + lane] += gain;
}

// =====================================================================================
// Straightforward element-parallel kernels (API completeness; not on the headline path)
// =====================================================================================
__global__ void k_edge_cut_mask(const uint8_t* __restrict__ x, int64_t B, int64_t N,
                                const int32_t* __restrict__ eu, const int32_t* __restrict__ ev, int64_t E,
                                uint8_t* __restrict__ out) {
    const int64_t total = B * E;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / E, e = t - b * E;
        const uint8_t* row = x + b * N;
        out[t] = (uint8_t)((row[eu[e]] != 0) != (row[ev[e]] != 0));
    }
}

// cutdeg over the env's *stored* adjacency (out-neighbours only unless bidirectional)
__global__ void k_node_cutdeg(const uint8_t* __restrict__ x, int64_t B, int64_t N,
                              const int32_t* __restrict__ erowptr, const int32_t* __restrict__ ev,
                              int64_t* __restrict__ out) {
    const int64_t total = B * N;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / N, i = t - b * N;
        const uint8_t* row = x + b * N;
        const bool xi = row[i] != 0;
        int c = 0;
        for (int j = erowptr[i]; j < erowptr[i + 1]; ++j) c += ((row[ev[j]] != 0) != xi);
        out[t] = c;
    }
}

// whole-batch min / max of the local-search weights, minmax int32 [2][N] (row 0 = min, row 1 = max, filled with
// INT32_MAX / INT32_MIN by the launcher): an atomic only where this tile improves what is already there -- after the
// first few tiles almost never (unconditional atomics cost 3x the kernel).
__device__ __forceinline__ void ws_minmax_update(int32_t* __restrict__ minmax, int64_t N, int64_t i, int lo, int hi) {
    // agent-scope loads: served by L2, where the atomics land (a CU's L1 would keep showing the fill value)
    if (lo < __hip_atomic_load(minmax + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(minmax + i, lo);
    if (hi > __hip_atomic_load(minmax + N + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(minmax + N + i, hi);
}

// element-parallel local-search weights (no LDS tile: any N): ws[b,i] = #stored out-neighbours - mult * cutdeg[b,i]
template <typename WT>
__global__ void k_ls_weights_elem(const uint8_t* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ erowptr,
                                  const int32_t* __restrict__ ev, int mult, WT* __restrict__ ws, int64_t pitch,
                                  int32_t* __restrict__ minmax) {
    const int64_t total = B * N;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / N, i = t - b * N;
        const uint8_t* row = x + b * N;
        const bool xi = row[i] != 0;
        const int r0 = erowptr[i], r1 = erowptr[i + 1];
        int c = 0;
        for (int j = r0; j < r1; ++j) c += ((row[ev[j]] != 0) != xi);
        const int val = (r1 - r0) - mult * c;
        ws[b * pitch + i] = (WT)val;
        if (minmax) ws_minmax_update(minmax, N, i, val, val);
    }
}

__global__ void k_fill_minmax(int32_t* __restrict__ minmax, int64_t N) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { minmax[i] = INT32_MAX; minmax[N + i] = INT32_MIN; }
}

template <bool WEIGHTED>
__global__ void k_delta_all(const uint8_t* __restrict__ x, int64_t B, int64_t N,
                            const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                            const int32_t* __restrict__ wgt, int32_t* __restrict__ out) {
    const int64_t total = B * N;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / N, i = t - b * N;
        const uint8_t* row = x + b * N;
        const bool xi = row[i] != 0;
        int d = 0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) {
            const int w = WEIGHTED ? wgt[j] : 1;
            d += ((row[col[j]] != 0) == xi) ? w : -w;
        }
        out[t] = d;
    }
}

// =====================================================================================
// K2 / K3 on a bit tile, lane = env form (weighted graphs, degrees >= 256; k_node_stats_bits below is the
// fast path): per-node cut degree (int64, the env's stored adjacency) or flip gain
// (int32, symmetric CSR, optional weights) for all nodes of 64 envs.  Nodes are independent, so the
// kTileWaves waves of the workgroup take alternating blocks of 128/sizeof(OutT) consecutive nodes;
// each lane (= env) collects its block in an LDS staging row and writes it back as full 128-byte
// lines (a lane-per-env layout writing one 4-byte result per node would touch 64 lines per store).
// Per (node, neighbour): v_readlane + v_lshl_add + broadcast ds_read + v_bfe + add, 8 per trip.
// =====================================================================================
template <typename OutT, bool DELTA, bool WEIGHTED, bool VEC>
__global__ __launch_bounds__(kTileWaves * kWave) void k_node_stats_tile(const uint8_t* __restrict__ x, int64_t B,
                                                                         int64_t N,
                                                                         const int32_t* __restrict__ rowptr,
                                                                         const int32_t* __restrict__ col,
                                                                         const int32_t* __restrict__ wgt,
                                                                         OutT* __restrict__ out) {
    constexpr int NB = 128 / (int)sizeof(OutT);       // nodes per block: one 128-B line per env
    constexpr int STRIDE = 144;                       // staging row stride in bytes (16-B aligned, off the 128-B period)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    const unsigned char* wbytes = smem;
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    unsigned char* stage = smem + (size_t)(N + 2) * 8 + (size_t)w * kWave * STRIDE;
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    if (threadIdx.x == 0) words[N] = 0;               // sentinel word: lanes past a row's end read zero
    static_assert(kWave * STRIDE >= kStageBytes, "the output staging rows double as the tile-load stage");
    tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane, w, kTileWaves, stage);
    __syncthreads();
    const int sh = lane & 31;
    const uint32_t half4 = (uint32_t)(lane >> 5) * 4u;
    const int64_t b = b0 + lane;
    for (int64_t i0 = (int64_t)w * NB; i0 < N; i0 += (int64_t)kTileWaves * NB) {
        const int nb_here = (int)((N - i0) < NB ? (N - i0) : NB);
        for (int k = 0; k < nb_here; ++k) {
            const int64_t i = i0 + k;
            const int r0 = rowptr[i], r1 = rowptr[i + 1];
            const uint32_t xi = (*reinterpret_cast<const uint32_t*>(wbytes + ((uint32_t)i * 8u + half4)) >> sh) & 1u;
            int acc = 0, wsum = 0;
            for (int base = r0; base < r1; base += kWave) {
                const int cnt = (r1 - base) < kWave ? (r1 - base) : kWave;
                const uint32_t my_nb = (lane < cnt) ? (uint32_t)col[base + lane] : (uint32_t)N;
                int my_w = 0;
                if constexpr (WEIGHTED) my_w = (lane < cnt) ? wgt[base + lane] : 0;
                for (int j = 0; j < cnt; j += 8) {
                    uint32_t wv[8];
                    int ww[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const uint32_t nb = (uint32_t)__builtin_amdgcn_readlane((int)my_nb, j + q);
                        wv[q] = *reinterpret_cast<const uint32_t*>(wbytes + (nb * 8u + half4));
                        if constexpr (WEIGHTED) ww[q] = __builtin_amdgcn_readlane(my_w, j + q);
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int bit = (int)((wv[q] >> sh) & 1u);
                        if constexpr (WEIGHTED) { acc += bit ? ww[q] : 0; wsum += ww[q]; }
                        else acc += bit;
                    }
                }
            }
            if constexpr (!WEIGHTED) wsum = r1 - r0;
            OutT res;
            if constexpr (DELTA) res = (OutT)(xi ? (2 * acc - wsum) : (wsum - 2 * acc));   // sum w (same ? +1 : -1)
            else res = (OutT)(xi ? (wsum - acc) : acc);                                  // #neighbours that differ
            *reinterpret_cast<OutT*>(stage + lane * STRIDE + k * (int)sizeof(OutT)) = res;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (b < B) {
            OutT* dst = out + b * N + i0;
            const unsigned char* src = stage + lane * STRIDE;
            if (nb_here == NB && ((((uintptr_t)dst) & 15) == 0)) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    reinterpret_cast<u32x4*>(dst)[q] = *reinterpret_cast<const u32x4*>(src + q * 16);
            } else {
                for (int k = 0; k < nb_here; ++k) dst[k] = *reinterpret_cast<const OutT*>(src + k * (int)sizeof(OutT));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// =====================================================================================
// K2 / K3 / local-search weights, bit-sliced with LANE = NODE (unweighted graphs, max degree < 65536).
// In the lane = env form above every (env, node, neighbour) costs ~4.5 instructions of one lane; here a lane owns
// a node and works on 64-env WORDS: per neighbour one random ds_read_b64, one XOR with the node's own word and
// a carry-save add into vertical counters -- c(e, i) = #{j in adj(i) : x_j != x_i} for all 64 envs in ~10
// instructions per (node, neighbour) instead of ~290.  Neighbours come from the lane-per-node slabs of
// rls_graph_ell (coalesced; rows shorter than the group's longest are padded with the node itself).  The
// counters are then read out four envs at a time ((plane >> r) & 0x01010101 lines up envs r, r+8, r+16, r+24 in
// the four bytes of a dword) and every env's 64 consecutive results leave as ONE 256-byte store (the lane = env
// form wrote 16 bytes per lane into 64 different rows).  out = cutdeg (int64) | deg - 2c (int32) | deg - mult c.
// =====================================================================================
constexpr int kNsWaves = 8;

template <int NP>
__device__ __forceinline__ uint32_t ns_extract4(const uint64_t (&pl)[8], int half, int r) {
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t h = half ? (uint32_t)(pl[p] >> 32) : (uint32_t)pl[p];
        acc += ((h >> r) & 0x01010101u) << p;
    }
    return acc;
}

// two envs per dword (16-bit fields: counts up to 65535) from NP planes: envs r and r + 16 of the half
template <int NP, int NPL>
__device__ __forceinline__ uint32_t ns_extract2(const uint64_t (&pl)[NPL], int half, int r) {
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const uint32_t h = half ? (uint32_t)(pl[p] >> 32) : (uint32_t)pl[p];
        acc += ((h >> r) & 0x00010001u) << p;
    }
    return acc;
}

// WIDE: graphs with hubs (256 <= max degree < 65536, e.g. Barabasi-Albert at n = 10^4): 16 counter planes instead of 8; a
// group whose longest row is >= 256 ripples its carries as far as its rows need and leaves through 16-bit fields, every
// other group runs exactly as in the narrow kernel.
// max and min of a bit-sliced counter (planes pl[0..NP), LSB first) over the envs of `vm`: walk the planes from the top,
// keeping the candidates that still can be the extremum
template <int NP, int NPL>
__device__ __forceinline__ void planes_minmax(const uint64_t (&pl)[NPL], uint64_t vm, int& mn, int& mx) {
    uint64_t a = vm, b = vm;
    mx = 0;
    mn = 0;
#pragma unroll
    for (int p = NP - 1; p >= 0; --p) {
        const uint64_t t = a & pl[p];
        if (t) { a = t; mx |= 1 << p; }
        const uint64_t u = b & ~pl[p];
        if (u) b = u; else mn |= 1 << p;
    }
}

// MODE 0: cutdeg int64, 1: flip gain int32, 2: local-search weight WT (int8 / int16 / int32) + the batch min / max per node
template <int MODE, bool VEC, bool WIDE, typename WT = int32_t, int NSW = kNsWaves>
__global__ __launch_bounds__(NSW * kWave) void k_node_stats_bits(const uint8_t* __restrict__ x, int64_t B, int64_t N,
                                                                     const int32_t* __restrict__ rowptr,
                                                                     const int32_t* __restrict__ ell_ptr,
                                                                     const int32_t* __restrict__ ell, int mult,
                                                                     void* __restrict__ out_v, int32_t* __restrict__ minmax,
                                                                     int64_t out_pitch,     // row pitch of out_v in elements (MODE 2; N otherwise)
                                                                     int stage_flags) {     // bit 0: row-piece stages (0: the tile alone fills LDS,
                                                                                            // N ~ 20 000: lane-per-env loads); bit 1: min / max stash
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int has_stage = stage_flags & 1;
    // MODE 2's batch min / max: every tile of the first round of workgroups reaches a node's fold at about the same time, finds the fill
    // value there and sends its atomic -- 768 of them per node, serialised in L2 (G22 2^16: 55 of the pre-pass's 131 us).  So only the
    // first few tiles ("seeds": 16) fold as they go; the others park (lo, hi) per node in LDS (the stages' bytes, idle after the tile load)
    // and fold after their last store, when the seeds' values are in the table and the test in front of the atomic fails almost always
    const bool park = MODE == 2 && (stage_flags & 2) && (int)blockIdx.x >= (stage_flags >> 8);     // (bits 8..: the number of seed tiles)
    uint32_t* stash = reinterpret_cast<uint32_t*>(smem + (((size_t)N * 8 + 15) & ~(size_t)15));      // [N] (lo + 32768) | (hi + 32768) << 16
    unsigned char* stage = smem + (((size_t)N * 8 + 15) & ~(size_t)15) + (size_t)w * kStageBytes;
    const int64_t b0 = (int64_t)blockIdx.x * kWave;
    tile_load_bits<uint8_t, VEC>(x, B, N, b0, words, lane, w, NSW, has_stage ? stage : nullptr);
    __syncthreads();
    const int64_t G = (N + 63) >> 6;
    const int nenv = (int)((B - b0) < kWave ? (B - b0) : kWave);
    const uint64_t vmask = nenv == kWave ? ~0ull : ((1ull << nenv) - 1);
    constexpr int NCP = WIDE ? 13 : 5;
    // c(e, i) for the group's 64 nodes over the neighbour blocks first, first + step, ... of 8 rounds each
    auto count_blocks = [&](int e0, int e1, uint32_t iself, uint64_t own, int ncp, int first, int step, uint64_t& ones, uint64_t& twos,
                            uint64_t& fours, uint64_t (&c)[NCP]) {
        // (requesting the next block's neighbour ids before this block's words are read -- the sweep's prefetch -- was measured here
        // and changes nothing: K3 G22 2^16 168.7 vs 167.2 us, weights 141 vs 139; six waves per SIMD already cover the round trip)
        for (int k = e0 + first * 8 * kWave; k < e1; k += step * 8 * kWave) {
            uint32_t nb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) nb[q] = (k + q * kWave < e1) ? (uint32_t)ell[k + q * kWave + lane] : iself;
            uint64_t d[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) d[q] = words[nb[q]] ^ own;
            uint64_t twosA, twosB, foursA, foursB, carry;
            csa(twosA, ones, ones, d[0], d[1]);
            csa(twosB, ones, ones, d[2], d[3]);
            csa(foursA, twos, twos, twosA, twosB);
            csa(twosA, ones, ones, d[4], d[5]);
            csa(twosB, ones, ones, d[6], d[7]);
            csa(foursB, twos, twos, twosA, twosB);
            csa(carry, fours, fours, foursA, foursB);
#pragma unroll
            for (int p = 0; p < NCP; ++p) {
                if (p < 5 || p < ncp) {
                    const uint64_t t = c[p] & carry;
                    c[p] ^= carry;
                    carry = t;
                }
            }
        }
    };
    auto emit = [&](int64_t i, int deg, int e, int cnt) {
        if constexpr (MODE == 0) reinterpret_cast<int64_t*>(out_v)[(b0 + e) * N + i] = cnt;
        else if constexpr (MODE == 1) reinterpret_cast<int32_t*>(out_v)[(b0 + e) * N + i] = deg - 2 * cnt;
        else reinterpret_cast<WT*>(out_v)[(b0 + e) * out_pitch + i] = (WT)(deg - mult * cnt);
    };
    // a hub group (longest row >= 256): 16-bit fields, two envs per dword
    auto emit_hub = [&](int64_t i, bool in, int deg, int md, const uint64_t (&pw)[16]) {
        if constexpr (MODE == 2) {
            if (minmax && in) {
                int cmn, cmx;
                planes_minmax<16>(pw, vmask, cmn, cmx);
                ws_minmax_update(minmax, N, i, deg - mult * cmx, deg - mult * cmn);
            }
        }
        for (int half = 0; half < 2; ++half) {
            for (int r = 0; r < 16; ++r) {
                const uint32_t acc = md < 1024 ? ns_extract2<10>(pw, half, r) : ns_extract2<16>(pw, half, r);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int e = half * 32 + r + 16 * j;
                    if (e < nenv && in) emit(i, deg, e, (int)((acc >> (16 * j)) & 0xFFFFu));
                }
            }
        }
    };
    // WIDE: the hub groups first, ALL waves on each -- a hub's row is hundreds of rounds (BA n = 10^4: ~370, a star: N), which one
    // wave used to walk alone while the other seven finished their twenty short groups each.  The blocks of 8 rounds are dealt
    // round-robin over the waves, the waves' bit-sliced counters added pairwise through the (now idle) row-piece stages --
    // 16 planes x 64 lanes x 8 B = 8 KB per wave, four waves' worth at a time -- and wave 0 writes the group.
    const bool coop = WIDE && has_stage;
    if constexpr (WIDE) {
        if (coop) {
            uint64_t* xch = reinterpret_cast<uint64_t*>(smem + (((size_t)N * 8 + 15) & ~(size_t)15));   // [4][16][64]
            static_assert(NSW == 8 && NSW * kStageBytes >= 4 * 16 * kWave * 8, "four waves' planes fit the stages");
            for (int64_t g = 0; g < G; ++g) {
                const int e0 = ell_ptr[g], e1 = ell_ptr[g + 1];
                const int md = (e1 - e0) >> 6;
                if (md < 256) continue;                          // (wave-uniform)
                const int64_t i = (g << 6) + lane;
                const bool in = i < N;
                const uint32_t iself = in ? (uint32_t)i : 0u;
                const uint64_t own = words[iself];
                const int deg = in ? rowptr[i + 1] - rowptr[i] : 0;
                uint64_t ones = 0, twos = 0, fours = 0, c[NCP];
#pragma unroll
                for (int p = 0; p < NCP; ++p) c[p] = 0;
                count_blocks(e0, e1, iself, own, NCP, w, NSW, ones, twos, fours, c);
                uint64_t pw[16] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4], c[5 % NCP], c[6 % NCP], c[7 % NCP], c[8 % NCP],
                                   c[9 % NCP], c[10 % NCP], c[11 % NCP], c[12 % NCP]};
                for (int half = NSW / 2; half >= 1; half >>= 1) {
                    __syncthreads();                             // (the previous step's readers are done with xch)
                    if (w >= half && w < 2 * half) {
#pragma unroll
                        for (int p = 0; p < 16; ++p) xch[((w - half) * 16 + p) * kWave + lane] = pw[p];
                    }
                    __syncthreads();
                    if (w < half) {
                        uint64_t cy = 0;
#pragma unroll
                        for (int p = 0; p < 16; ++p) {           // ripple add of two 16-plane numbers
                            const uint64_t o = xch[(w * 16 + p) * kWave + lane];
                            csa(cy, pw[p], pw[p], o, cy);
                        }
                    }
                }
                if (w == 0) emit_hub(i, in, deg, md, pw);
            }
            if (park) __syncthreads();                           // the stash lies where the last exchange was read
        }
    }
    // ---- K2 / K3 of a FULL tile through row staging (round 6, stage_flags bit 2): the per-group form below writes, per env, the 64
    // nodes of a group -- 256-byte pieces of rows 4N bytes apart, every other one starting mid-line: three lines touched, two of them
    // partially, per piece (4.3 TB/s in isolation at N = 2000, 4.1 at 10^4; 1 KB pieces of the same rows 5.4 / 4.8:
    // tools/ceilings/store_width.hip).  Here a wave counts TWO consecutive groups, parks their counts as bytes in its OWN 8 KB of
    // LDS [env][128 nodes] (the int8 weights' quad transpose makes a lane's dword four consecutive nodes of one env) and writes two
    // envs' 128 nodes per store instruction, 16 bytes per lane: 512-byte (K3) / 1 KB (K2) pieces.  Wave-private: no barrier, the
    // eight waves of a tile stay out of step as they are in the per-group form (a version with a tile-wide staging of 256-node blocks
    // behind two barriers per block was built first: G70 2^17 1908 us against 1662 per group -- its count and store phases ran in turn).
    if constexpr (MODE != 2 && NSW == 8) {
        if ((stage_flags & 4) && nenv == kWave) {
            constexpr int kRowPitch = 132;                                // bytes per env row of a wave's staging: 128 + 4 (banks)
            unsigned char* const stg = smem + (((size_t)N * 8 + 15) & ~(size_t)15) + (size_t)w * (kWave * kRowPitch);
            const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
            if (coop) __syncthreads();                                    // (the hub groups' last exchange was read from these bytes)
            const int64_t NP = (G + 1) >> 1;                              // pairs of groups
            for (int64_t pr = w; pr < NP; pr += NSW) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int64_t g = 2 * pr + h2;
                    if (g >= G) break;
                    const int64_t i = (g << 6) + lane;
                    const bool in = i < N;
                    const uint32_t iself = in ? (uint32_t)i : 0u;
                    const uint64_t own = words[iself];
                    const int e0 = ell_ptr[g], e1 = ell_ptr[g + 1];
                    const int md = (e1 - e0) >> 6;                        // longest row of the group (wave-uniform)
                    if (WIDE && md >= 256) continue;                      // (a hub group: written by the cooperative pass above)
                    uint64_t ones = 0, twos = 0, fours = 0, c[NCP];
#pragma unroll
                    for (int p = 0; p < NCP; ++p) c[p] = 0;
                    int ncp = 5;
                    if constexpr (WIDE) while ((8 << ncp) <= md) ++ncp;
                    count_blocks(e0, e1, iself, own, ncp, 0, 1, ones, twos, fours, c);
                    const uint64_t pl[8] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4]};
                    unsigned char* const obase = stg + (size_t)((lane & 3) * 8) * kRowPitch + h2 * 64 + (lane & ~3);
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
#pragma unroll 2
                        for (int r = 0; r < 8; ++r) {
                            uint32_t v = md < 16 ? ns_extract4<4>(pl, half, r)
                                                 : (md < 64 ? ns_extract4<6>(pl, half, r) : ns_extract4<8>(pl, half, r));   // 4 envs x count
                            uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);      // lane ^ 1
                            v = __builtin_amdgcn_perm(o, v, sel1);
                            o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);               // lane ^ 2
                            v = __builtin_amdgcn_perm(o, v, sel2);                    // 4 consecutive nodes of env half * 32 + r + 8 (lane & 3)
                            *reinterpret_cast<uint32_t*>(obase + (size_t)(half * 32 + r) * kRowPitch) = v;
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // LDS ops of one wave execute in order; the compiler must not move them
                __builtin_amdgcn_wave_barrier();
                // the pair's 128 nodes of two envs per store: lanes 0 .. 31 env 2k, lanes 32 .. 63 env 2k + 1, four nodes each
                const int sub = lane >> 5, q4 = (lane & 31) * 4;
                const int64_t i4 = (pr << 7) + q4;
                bool on = i4 < N;                                          // (N % 4 == 0: all four or none)
                int dg[4] = {0, 0, 0, 0};
                if (on) {
                    int rp[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) rp[j] = rowptr[i4 + j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) dg[j] = rp[j + 1] - rp[j];
                    if constexpr (WIDE) {
                        const int64_t gi = i4 >> 6;
                        if (((ell_ptr[gi + 1] - ell_ptr[gi]) >> 6) >= 256) on = false;       // a hub group's nodes: already written
                    }
                }
#pragma unroll 4
                for (int k = 0; k < kWave / 2; ++k) {
                    const int e = 2 * k + sub;
                    const uint32_t d = *reinterpret_cast<const uint32_t*>(stg + (size_t)e * kRowPitch + q4);
                    if (on) {
                        if constexpr (MODE == 1) {
                            typedef int i32x4 __attribute__((ext_vector_type(4)));
                            const i32x4 o = {dg[0] - 2 * (int)(d & 0xFFu), dg[1] - 2 * (int)((d >> 8) & 0xFFu),
                                             dg[2] - 2 * (int)((d >> 16) & 0xFFu), dg[3] - 2 * (int)(d >> 24)};
                            *reinterpret_cast<i32x4*>(reinterpret_cast<int32_t*>(out_v) + (b0 + e) * N + i4) = o;
                        } else {
                            typedef int64_t i64x2 __attribute__((ext_vector_type(2)));
                            int64_t* const q = reinterpret_cast<int64_t*>(out_v) + (b0 + e) * N + i4;
                            *reinterpret_cast<i64x2*>(q) = i64x2{(int64_t)(d & 0xFFu), (int64_t)((d >> 8) & 0xFFu)};
                            *reinterpret_cast<i64x2*>(q + 2) = i64x2{(int64_t)((d >> 16) & 0xFFu), (int64_t)(d >> 24)};
                        }
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
    }
    for (int64_t g = w; g < G; g += NSW) {
        const int64_t i = (g << 6) + lane;
        const bool in = i < N;
        const uint32_t iself = in ? (uint32_t)i : 0u;
        const uint64_t own = words[iself];
        const int e0 = ell_ptr[g], e1 = ell_ptr[g + 1];
        const int deg = in ? rowptr[i + 1] - rowptr[i] : 0;
        const int md = (e1 - e0) >> 6;                        // longest row of the group (wave-uniform)
        if (coop && md >= 256) continue;                      // done above
        uint64_t ones = 0, twos = 0, fours = 0, c[NCP];
#pragma unroll
        for (int p = 0; p < NCP; ++p) c[p] = 0;
        int ncp = 5;                                          // carry planes this group can reach: counts < 8 << ncp
        if constexpr (WIDE) while ((8 << ncp) <= md) ++ncp;
        count_blocks(e0, e1, iself, own, ncp, 0, 1, ones, twos, fours, c);
        const uint64_t pl[8] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4]};
        if constexpr (WIDE) {
            if (md >= 256) {                                  // a hub group without the stages (N ~ 20 000): one wave walks it
                const uint64_t pw[16] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4], c[5 % NCP], c[6 % NCP], c[7 % NCP], c[8 % NCP],
                                         c[9 % NCP], c[10 % NCP], c[11 % NCP], c[12 % NCP]};
                emit_hub(i, in, deg, md, pw);
                continue;
            }
        }
        if constexpr (MODE == 2) {
            if (minmax && in) {
                int cmn, cmx;
                planes_minmax<8>(pl, vmask, cmn, cmx);
                const int lo = deg - mult * cmx, hi = deg - mult * cmn;
                if (park) stash[i] = (uint32_t)(lo + 32768) | ((uint32_t)(hi + 32768) << 16);
                else ws_minmax_update(minmax, N, i, lo, hi);
            }
        }
        if constexpr (MODE == 2 && sizeof(WT) == 1) {
            if (nenv == kWave && (out_pitch & 3) == 0 && (N & 3) == 0) {
                // int8 weights of a full tile leave as DWORDS: per (half, r) every lane holds its node's values for the 4 envs
                // r, r + 8, r + 16, r + 24 of the half as bytes; a 4 x 4 byte transpose inside each lane quad (two DPP moves + two
                // v_perm) turns that into 4 consecutive NODES of ONE env per lane -- 16 store instructions of 4 x 64 bytes per
                // group where the byte stores took 64 of 64 bytes (the pre-pass was store-issue bound: 0.24 of HBM)
                const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
                const uint32_t bias = (uint32_t)(deg + 128) * 0x01010101u;       // per byte deg + 128 - mult * cnt in [1, 255]: no borrows
                int8_t* const obase = reinterpret_cast<int8_t*>(out_v) + (b0 + (lane & 3) * 8) * out_pitch + ((g << 6) + (lane & ~3));
                if (in) {      // (N % 4 == 0: a quad of nodes is inside the row or outside it as a whole)
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
#pragma unroll 2
                        for (int r = 0; r < 8; ++r) {
                            const uint32_t acc = md < 16 ? ns_extract4<4>(pl, half, r)
                                                         : (md < 64 ? ns_extract4<6>(pl, half, r) : ns_extract4<8>(pl, half, r));
                            uint32_t v = (bias - (uint32_t)mult * acc) ^ 0x80808080u;     // 4 envs x int8(deg - mult * cnt)
                            uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);      // lane ^ 1
                            v = __builtin_amdgcn_perm(o, v, sel1);
                            o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);               // lane ^ 2
                            v = __builtin_amdgcn_perm(o, v, sel2);
                            *reinterpret_cast<uint32_t*>(obase + (int64_t)(half * 32 + r) * out_pitch) = v;
                        }
                    }
                }
                continue;
            }
        }
        // (K3 with 16-BYTE stores -- the int8 weights' quad transpose applied to the int32 gains, 16 store instructions of 4 rows x
        // 256 bytes per group instead of 64 of 256 bytes, ~30 % fewer VALU -- was built and measured in round 5: G22 2^16 164.6 ->
        // 161.8 us, G70 2^17 1647 -> 1719, BA-1e4 1099 -> 1083: nothing.  K3 sits at what its WRITE PATTERN reaches on this pool --
        // 64 rows x 256-byte pieces, rows 4N bytes apart and every other one starting mid-line: 4.0 TB/s in isolation
        // (tools/ceilings/store_width.hip) -- not at an instruction bound.  Non-temporal result stores, also round 5: G70 2^17 K3 1627 vs
        // 1628 us, K2 2320-2340 either way, BA-1e4 2^15 546 vs 545: nothing)
        if (nenv == kWave) {
            // full tile: no per-store guards (each cost a scalar compare / exec save / branch around a 4-instruction store)
            if (in) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll 2
                    for (int r = 0; r < 8; ++r) {
                        const uint32_t acc = md < 16 ? ns_extract4<4>(pl, half, r)
                                                     : (md < 64 ? ns_extract4<6>(pl, half, r) : ns_extract4<8>(pl, half, r));
#pragma unroll
                        for (int j = 0; j < 4; ++j) emit(i, deg, half * 32 + r + 8 * j, (int)((acc >> (8 * j)) & 0xFFu));
                    }
                }
            }
        } else {
            for (int half = 0; half < 2; ++half) {
                for (int r = 0; r < 8; ++r) {
                    if (half * 32 + r >= nenv) break;         // envs are consecutive: nothing further in this half
                    const uint32_t acc = md < 16 ? ns_extract4<4>(pl, half, r)
                                                 : (md < 64 ? ns_extract4<6>(pl, half, r) : ns_extract4<8>(pl, half, r));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = half * 32 + r + 8 * j;
                        if (e < nenv && in) emit(i, deg, e, (int)((acc >> (8 * j)) & 0xFFu));
                    }
                }
            }
        }
    }
    if constexpr (MODE == 2) {
        if (park && minmax) {      // the parked folds, in the order the seeds folded theirs (every wave reads what it wrote itself)
            // ... in eight phases ~1 us apart (a tile's phase = its index mod 8), so that a phase finds what the phases before it folded
            // (G22 2^16: 110 -> 96 us; without the fold the pre-pass is 80)
            for (int k = 0; k < (int)(blockIdx.x & 7); ++k) __builtin_amdgcn_s_sleep(32);
            // four groups at a time: their eight table reads are in flight together (one after the other they were a chain of L2 round
            // trips per wave)
            for (int64_t g0 = w; g0 < G; g0 += 4 * NSW) {
                int cur_lo[4], cur_hi[4];
                bool on[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t g = g0 + (int64_t)q * NSW, i = (g << 6) + lane;
                    on[q] = g < G && i < N && ((ell_ptr[g + 1] - ell_ptr[g]) >> 6) < 256;   // (a hub group: folded by emit_hub as it went)
                    const int64_t ic = on[q] ? i : 0;
                    cur_lo[q] = __hip_atomic_load(minmax + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cur_hi[q] = __hip_atomic_load(minmax + N + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (!on[q]) continue;
                    const int64_t i = ((g0 + (int64_t)q * NSW) << 6) + lane;
                    const uint32_t v = stash[i];
                    const int lo = (int)(v & 0xFFFFu) - 32768, hi = (int)(v >> 16) - 32768;
                    if (lo < cur_lo[q]) atomicMin(minmax + i, lo);
                    if (hi > cur_hi[q]) atomicMax(minmax + N + i, hi);
                }
            }
        }
    }
}

// K2 / K3 / the weights pre-pass on HALF tiles (rls_tile32.h), for graphs past the 64-env tile: lane = node over the same ELL slabs,
// counters on 32-bit planes, every group walked by one wave (hub rows included: 16 planes, 16-bit fields), plain per-env stores.
template <int MODE, bool VEC, bool WIDE, typename WT, typename NT = uint32_t>
__global__ __launch_bounds__(kNsWaves * kWave) void k_node_stats_bits32(const uint8_t* __restrict__ x, int64_t B, int64_t N,
                                                                       const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ell_ptr,
                                                                       const int32_t* __restrict__ ell, int mult, void* __restrict__ out_v,
                                                                       int32_t* __restrict__ minmax, int64_t out_pitch, int has_stage) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // NT: the tile's word -- uint32_t = the half tile; uint16_t / uint8_t = the narrow tiles of 16 / 8 envs (rls_tile32.h) for rows
    // whose half tile is past the LDS; the planes stay 32-bit registers with the upper bits clear
    constexpr int EN = narrow_tile<NT>::E;
    NT* words32 = reinterpret_cast<NT*>(smem);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t b0 = (int64_t)blockIdx.x * EN;
    if constexpr (sizeof(NT) == 4) {
        unsigned char* stage = smem + (((size_t)N * 4 + 15) & ~(size_t)15) + (size_t)w * kStageBytes;
        tile32_load_bits<uint8_t, VEC>(x, B, N, b0, words32, lane, w, kNsWaves, (has_stage & 1) ? stage : nullptr);
    } else {
        tilen_load_bits<uint8_t, NT, VEC>(x, B, N, b0, words32, lane, w, kNsWaves);
    }
    __syncthreads();
    const int64_t G = (N + 63) >> 6;
    const int nenv = (int)((B - b0) < EN ? (B - b0) : EN);
    const uint64_t vmask = (1ull << nenv) - 1;
    constexpr int NCP = WIDE ? 13 : 5;
    auto emit = [&](int64_t i, int deg, int e, int cnt) {
        if constexpr (MODE == 0) reinterpret_cast<int64_t*>(out_v)[(b0 + e) * N + i] = cnt;
        else if constexpr (MODE == 1) reinterpret_cast<int32_t*>(out_v)[(b0 + e) * N + i] = deg - 2 * cnt;
        else reinterpret_cast<WT*>(out_v)[(b0 + e) * out_pitch + i] = (WT)(deg - mult * cnt);
    };
    // K3 of a FULL half tile through wave-private row staging (round 6; has_stage bit 2; the 64-env kernel's comment): two groups'
    // counts parked as bytes [env][128 nodes], two envs' 128 nodes per store instruction, 16 bytes per lane -- 512-byte row pieces
    // where the per-group form below writes 256-byte ones.  The staging takes the bytes of the (idle) row-piece stages + 2 KB.
    if constexpr (MODE == 1 && sizeof(NT) == 4) {
        if ((has_stage & 4) && nenv == EN) {
            constexpr int kRowPitch = 132;
            unsigned char* const stg = smem + (((size_t)N * 4 + 15) & ~(size_t)15) + (size_t)w * (EN * kRowPitch);
            const uint32_t sel1 = (lane & 1) ? 0x03070105u : 0x06020400u, sel2 = (lane & 2) ? 0x03020706u : 0x05040100u;
            const int64_t NP = (G + 1) >> 1;
            for (int64_t pr = w; pr < NP; pr += kNsWaves) {
                bool hub[2] = {false, false};
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int64_t g = 2 * pr + h2;
                    if (g >= G) break;
                    const int64_t i = (g << 6) + lane;
                    const bool in = i < N;
                    const uint32_t iself = in ? (uint32_t)i : 0u;
                    const uint32_t own = (uint32_t)words32[iself];
                    const int e0 = ell_ptr[g], e1 = ell_ptr[g + 1];
                    const int md = (e1 - e0) >> 6;
                    uint32_t ones = 0, twos = 0, fours = 0, c[NCP];
#pragma unroll
                    for (int p = 0; p < NCP; ++p) c[p] = 0;
                    int ncp = 5;
                    if constexpr (WIDE) while ((8 << ncp) <= md) ++ncp;
                    for (int k = e0; k < e1; k += 8 * kWave) {
                        uint32_t nb[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) nb[q] = (k + q * kWave < e1) ? (uint32_t)ell[k + q * kWave + lane] : iself;
                        uint32_t d[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) d[q] = (uint32_t)words32[nb[q]] ^ own;
                        uint32_t twosA, twosB, foursA, foursB, carry;
                        csa32(twosA, ones, ones, d[0], d[1]);
                        csa32(twosB, ones, ones, d[2], d[3]);
                        csa32(foursA, twos, twos, twosA, twosB);
                        csa32(twosA, ones, ones, d[4], d[5]);
                        csa32(twosB, ones, ones, d[6], d[7]);
                        csa32(foursB, twos, twos, twosA, twosB);
                        csa32(carry, fours, fours, foursA, foursB);
#pragma unroll
                        for (int p = 0; p < NCP; ++p) {
                            if (p < 5 || p < ncp) {
                                const uint32_t t = c[p] & carry;
                                c[p] ^= carry;
                                carry = t;
                            }
                        }
                    }
                    const uint32_t pw[16] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4], c[5 % NCP], c[6 % NCP], c[7 % NCP], c[8 % NCP],
                                             c[9 % NCP], c[10 % NCP], c[11 % NCP], c[12 % NCP]};
                    if (WIDE && md >= 256) {                      // a hub group: 16-bit counts, written here as the per-group form writes them
                        hub[h2] = true;
                        if (in) {
                            const int deg = rowptr[i + 1] - rowptr[i];
                            for (int r = 0; r < 16; ++r) {
                                uint32_t acc = 0;
#pragma unroll
                                for (int p = 0; p < 16; ++p) acc += ((pw[p] >> r) & 0x00010001u) << p;
                                emit(i, deg, r, (int)(acc & 0xFFFFu));
                                emit(i, deg, r + 16, (int)(acc >> 16));
                            }
                        }
                        continue;
                    }
                    unsigned char* const obase = stg + (size_t)((lane & 3) * 8) * kRowPitch + h2 * 64 + (lane & ~3);
#pragma unroll 2
                    for (int r = 0; r < 8; ++r) {
                        uint32_t v = 0;
#pragma unroll
                        for (int p = 0; p < 8; ++p) v += ((pw[p] >> r) & 0x01010101u) << p;                 // envs r, r + 8, r + 16, r + 24 of this node
                        uint32_t o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);      // lane ^ 1
                        v = __builtin_amdgcn_perm(o, v, sel1);
                        o = (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);               // lane ^ 2
                        v = __builtin_amdgcn_perm(o, v, sel2);                        // 4 consecutive nodes of env r + 8 (lane & 3)
                        *reinterpret_cast<uint32_t*>(obase + (size_t)r * kRowPitch) = v;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                const int sub = lane >> 5, q4 = (lane & 31) * 4;
                const int64_t i4 = (pr << 7) + q4;
                bool on = i4 < N && !hub[(q4 >> 6) & 1];                  // (N % 4 == 0: all four nodes or none)
                int dg[4] = {0, 0, 0, 0};
                if (on) {
                    int rp[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) rp[j] = rowptr[i4 + j];
#pragma unroll
                    for (int j = 0; j < 4; ++j) dg[j] = rp[j + 1] - rp[j];
                }
#pragma unroll 4
                for (int k = 0; k < EN / 2; ++k) {
                    const int e = 2 * k + sub;
                    const uint32_t d = *reinterpret_cast<const uint32_t*>(stg + (size_t)e * kRowPitch + q4);
                    if (on) {
                        typedef int i32x4 __attribute__((ext_vector_type(4)));
                        const i32x4 o = {dg[0] - 2 * (int)(d & 0xFFu), dg[1] - 2 * (int)((d >> 8) & 0xFFu),
                                         dg[2] - 2 * (int)((d >> 16) & 0xFFu), dg[3] - 2 * (int)(d >> 24)};
                        *reinterpret_cast<i32x4*>(reinterpret_cast<int32_t*>(out_v) + (b0 + e) * N + i4) = o;
                    }
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
            return;
        }
    }
    for (int64_t g = w; g < G; g += kNsWaves) {
        const int64_t i = (g << 6) + lane;
        const bool in = i < N;
        const uint32_t iself = in ? (uint32_t)i : 0u;
        const uint32_t own = (uint32_t)words32[iself];
        const int e0 = ell_ptr[g], e1 = ell_ptr[g + 1];
        const int deg = in ? rowptr[i + 1] - rowptr[i] : 0;
        const int md = (e1 - e0) >> 6;                        // longest row of the group (wave-uniform)
        uint32_t ones = 0, twos = 0, fours = 0, c[NCP];
#pragma unroll
        for (int p = 0; p < NCP; ++p) c[p] = 0;
        int ncp = 5;
        if constexpr (WIDE) while ((8 << ncp) <= md) ++ncp;
        for (int k = e0; k < e1; k += 8 * kWave) {
            uint32_t nb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) nb[q] = (k + q * kWave < e1) ? (uint32_t)ell[k + q * kWave + lane] : iself;
            uint32_t d[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) d[q] = (uint32_t)words32[nb[q]] ^ own;
            uint32_t twosA, twosB, foursA, foursB, carry;
            csa32(twosA, ones, ones, d[0], d[1]);
            csa32(twosB, ones, ones, d[2], d[3]);
            csa32(foursA, twos, twos, twosA, twosB);
            csa32(twosA, ones, ones, d[4], d[5]);
            csa32(twosB, ones, ones, d[6], d[7]);
            csa32(foursB, twos, twos, twosA, twosB);
            csa32(carry, fours, fours, foursA, foursB);
#pragma unroll
            for (int p = 0; p < NCP; ++p) {
                if (p < 5 || p < ncp) {
                    const uint32_t t = c[p] & carry;
                    c[p] ^= carry;
                    carry = t;
                }
            }
        }
        const uint32_t pw[16] = {ones, twos, fours, c[0], c[1], c[2], c[3], c[4], c[5 % NCP], c[6 % NCP], c[7 % NCP], c[8 % NCP],
                                 c[9 % NCP], c[10 % NCP], c[11 % NCP], c[12 % NCP]};
        const int np = (WIDE && md >= 256) ? 16 : 8;          // planes that can be set (wave-uniform)
        if constexpr (MODE == 2) {
            if (minmax && in) {
                uint64_t a = vmask, bm = vmask;
                int mx = 0, mn = 0;
                for (int p = np - 1; p >= 0; --p) {           // planes_minmax on the half tile's envs
                    const uint64_t t = a & pw[p];
                    if (t) { a = t; mx |= 1 << p; }
                    const uint64_t u = bm & ~(uint64_t)pw[p];
                    if (u) bm = u; else mn |= 1 << p;
                }
                ws_minmax_update(minmax, N, i, deg - mult * mx, deg - mult * mn);
            }
        }
        if (!in) continue;
        if (np == 16) {                                       // a hub group: 16-bit fields, envs r and r + 16
            for (int r = 0; r < (EN < 16 ? EN : 16); ++r) {
                uint32_t acc = 0;
#pragma unroll
                for (int p = 0; p < 16; ++p) acc += ((pw[p] >> r) & 0x00010001u) << p;
                if (r < nenv) emit(i, deg, r, (int)(acc & 0xFFFFu));
                if constexpr (EN > 16)
                    if (r + 16 < nenv) emit(i, deg, r + 16, (int)(acc >> 16));
            }
        } else {                                              // byte fields, envs r, r + 8, r + 16, r + 24
            for (int r = 0; r < 8; ++r) {
                uint32_t acc = 0;
#pragma unroll
                for (int p = 0; p < 8; ++p) acc += ((pw[p] >> r) & 0x01010101u) << p;
#pragma unroll
                for (int j = 0; j < EN / 8; ++j)
                    if (r + 8 * j < nenv) emit(i, deg, r + 8 * j, (int)((acc >> (8 * j)) & 0xFFu));
            }
        }
    }
}

static inline size_t node_stats_bits32_lds(int64_t N, bool with_stage) {
    return (((size_t)N * 4 + 15) & ~(size_t)15) + (with_stage ? (size_t)kNsWaves * kStageBytes : 0);
}
static inline size_t node_stats_bits_lds(int64_t N, bool with_stage = true, int waves = kNsWaves) {
    return (((size_t)N * 8 + 15) & ~(size_t)15) + (with_stage ? (size_t)waves * kStageBytes : 0);
}
// the bit-sliced kernel needs the slabs, an unweighted graph, byte-sized counters and a tile that fits.  A tile costs about
// 0.011 us per node however few envs it holds, the element-parallel kernels about 2e-6 us per (env, node + entry): K3 on a
// G22-sized graph 28 us flat vs 9 / 21 / 68 us at 64 / 256 / 1024 envs, N = 10^4 with 10^4 edges 88 flat vs 7 / 17 / 70
// (tools/sweeps/node_stats_forms.py) -- so small batches go element-parallel
static inline bool node_stats_batch_fills_tiles(const rls_graph* g, int64_t B) {
    const int64_t force = (int64_t)knob(KN_NODE_STATS_MIN_B, -1);   // dev knob
    if (force >= 0) return B >= force;
    return (double)B * (double)(g->num_nodes + g->nnz) > 4000.0 * (double)g->num_nodes;
}
static inline bool node_stats_use_bits(const rls_graph* g, const int32_t* ell_ptr, const int32_t* ell, int64_t B) {
    const bool off = knob_on(KN_NODE_STATS_LANE_ENV);   // dev knob: the lane = env kernels
    return !off && ell_ptr && ell && !g->wgt && g->max_degree < 65536 && node_stats_batch_fills_tiles(g, B) &&
           (knob(KN_NARROW_TILE, 1) != 0 ? narrow_words_bytes<uint8_t>(g->num_nodes)     // (half tiles without the row-piece stage if
                                         : node_stats_bits32_lds(g->num_nodes, false)) <= (size_t)kLdsBytes;   // need be, narrow ones past them)
}
template <int MODE, typename WT = int32_t>
static int launch_node_stats_bits(const rls_graph* g, const uint8_t* x, int64_t B, const int32_t* rowptr,
                                  const int32_t* ell_ptr, const int32_t* ell, int mult, void* out, void* stream,
                                  int32_t* minmax = nullptr, int64_t out_pitch = 0) {
    const int64_t N = g->num_nodes;
    const bool vec = tile_rows_aligned(x, N, 1);
    const bool wide = g->max_degree >= 256;
    const int knob32 = (int)knob(KN_NS_TILE32, -1);   // dev knob: half tiles at any size
    // Half tiles: past the 64-env tile; and, for byte rows of 16-byte multiples whose half tile has room for its stage, where they
    // measure faster (this round's GPU runs): batches of few tiles -- a tile costs the same however few envs it holds, so twice
    // as many half as long workgroups win while CUs are idle: K3 / K2 up to 8192 envs (G22-sized 4096: 31 -> 23 us, BA n = 10^4
    // 135 -> 101, G70-sized 88 -> 62), the weights pre-pass up to 2048 (its 64-env kernel has the dword stores) --, K3 on short
    // rows at full batches (G22-sized 2^16: 166 -> 154 us), and -- few tiles again -- rows whose 64-env tile has no room for its
    // stage (N > ~15 800: K3 at N = 20 000, 4096 envs 278 -> 186 us)
    const int64_t t64 = ceil_div(B, kWave);
    const bool few = MODE == 2 ? 8 * t64 <= (int64_t)num_cus() : 2 * t64 <= (int64_t)num_cus();
    // (round 6: from two 64-env tiles per CU on -- N = 3008, 2^15 envs: 121 us on 64-env tiles, 109-112 on half tiles)
    const bool full_k3 = MODE == 1 && t64 >= 2 * (int64_t)num_cus() && (size_t)N * 8 <= 64 * 1024;
    const bool prefer32 = knob32 < 0 && vec && (N & 7) == 0 && node_stats_bits32_lds(N, true) <= (size_t)kLdsBytes &&
                          (few || full_k3 || (node_stats_bits_lds(N, true) > (size_t)kLdsBytes && 2 * t64 <= (int64_t)num_cus()));
    // Narrow tiles (16 / 8 envs, rls_tile32.h) where the half tile is past the LDS (N > 40 960): the rows these took before went
    // element-parallel -- one L2 gather per (env, entry)
    // -- and, like K1 / K6 / K5 (narrow_policy), for batches of few tiles on rows the narrow loader's fast path takes: a tile's load,
    // count and stores are one chain whatever it holds, so 256 CUs want 256+ tiles (tools/timing/narrow_ns_ab.py, K3 / K2 / weights in us,
    // wide -> narrow: N = 10^4, 4096 envs 62 / 83 / 95 -> 44 / 74 / 91 (16 envs); N = 20 000, 1024 envs 128 / 102 / 148 -> 75 / 55 / 114
    // (8 envs); G22-sized, 256 envs 18.4 / 15.3 / 25.5 -> 11.9 / 9.5 / 22.5).  The weights' 8-env tiles only up to 512 envs (every tile
    // folds its min / max into the table with atomics: G22-sized 1024 envs 27.8 -> 32.6)
    const int64_t nk = knob(KN_NARROW_TILE, 1);
    int auto_n = 0;
    if (nk == 1 && knob32 < 0 && vec && (N & 7) == 0) {
        const int64_t cus = num_cus();
        if (MODE == 2) {    // (G22-sized, 4096 envs on 16-env tiles: 31 -> 57 us -- 256 tiles' worth of min / max atomics per node)
            if (ceil_div(B, 8) <= cus / 4) auto_n = 8;
            else if (N >= 8192 && ceil_div(B, 16) <= cus / 4) auto_n = 16;
        } else {
            if (ceil_div(B, 8) <= cus) auto_n = 8;
            else if (ceil_div(B, 16) <= cus) auto_n = 16;
        }
    }
    if (nk != 0 && (nk >= 2 || auto_n || node_stats_bits32_lds(N, false) > (size_t)kLdsBytes)) {
        const size_t l16 = narrow_words_bytes<uint16_t>(N), l8 = narrow_words_bytes<uint8_t>(N);
        const bool w16 = nk != 3 && auto_n != 8 && l16 <= (size_t)kLdsBytes;
        const size_t ln = w16 ? l16 : l8;
        const dim3 gn((unsigned)ceil_div(B, (int64_t)(w16 ? 16 : 8))), bn(kNsWaves * kWave);
#define RLS_NSN_LAUNCH(VEC, WIDE)                                                                                    \
    do {                                                                                                            \
        if (w16) { auto kern = k_node_stats_bits32<MODE, VEC, WIDE, WT, uint16_t>; ensure_dyn_lds((const void*)kern, ln);   \
                   hipLaunchKernelGGL(kern, gn, bn, ln, as_stream(stream), x, B, N, rowptr, ell_ptr, ell, mult, out, minmax, \
                                      out_pitch > 0 ? out_pitch : N, 0); }                                          \
        else     { auto kern = k_node_stats_bits32<MODE, VEC, WIDE, WT, uint8_t>; ensure_dyn_lds((const void*)kern, ln);    \
                   hipLaunchKernelGGL(kern, gn, bn, ln, as_stream(stream), x, B, N, rowptr, ell_ptr, ell, mult, out, minmax, \
                                      out_pitch > 0 ? out_pitch : N, 0); }                                          \
    } while (0)
        if (wide) { if (vec) RLS_NSN_LAUNCH(true, true); else RLS_NSN_LAUNCH(false, true); }
        else      { if (vec) RLS_NSN_LAUNCH(true, false); else RLS_NSN_LAUNCH(false, false); }
#undef RLS_NSN_LAUNCH
        return check_launch("k_node_stats_bits32<narrow>");
    }
    if (knob32 > 0 || prefer32 || node_stats_bits_lds(N, false) > (size_t)kLdsBytes) {   // half tiles (rls_tile32.h)
        int st32 = (vec && (N & 7) == 0 && node_stats_bits32_lds(N, true) <= (size_t)kLdsBytes) ? 1 : 0;
        size_t l32 = node_stats_bits32_lds(N, st32 != 0);
        {   // K3's row staging on full half tiles (the kernel's comment): 16-byte-aligned output, rows of 4-node multiples
            const int rk = (int)knob(KN_NS_ROWS, -1);
            const size_t lrows = node_stats_bits32_lds(N, false) + (size_t)kNsWaves * kHalf * 132;
            // measured (tools/timing/k3_rows.py, half tiles per group -> rows): N = 3008 112 -> 109 us, G70-sized 1603 -> 1527, BA n = 10^4
            // 1093 -> 1043; G22-sized 162 -> 167 (sixteen 512-byte pieces per pair are too few stores to pay for the staging): from 3000 nodes
            if (MODE == 1 && st32 && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && (rk < 0 ? N >= 3000 : rk != 0) &&
                lrows <= (size_t)kLdsBytes) {
                st32 |= 4;
                if (lrows > l32) l32 = lrows;
            }
        }
        const dim3 g32((unsigned)ceil_div(B, (int64_t)kHalf)), b32(kNsWaves * kWave);
#define RLS_NS32_LAUNCH(KERN)                                                                                        \
    do {                                                                                                            \
        auto kern = KERN;                                                                                           \
        if (l32 > 64 * 1024) ensure_dyn_lds((const void*)kern, l32); \
        hipLaunchKernelGGL(kern, g32, b32, l32, as_stream(stream), x, B, N, rowptr, ell_ptr, ell, mult, out, minmax,  \
                           out_pitch > 0 ? out_pitch : N, st32);                                                     \
    } while (0)
        if (wide) {
            if (vec) RLS_NS32_LAUNCH((k_node_stats_bits32<MODE, true, true, WT>));
            else RLS_NS32_LAUNCH((k_node_stats_bits32<MODE, false, true, WT>));
        } else {
            if (vec) RLS_NS32_LAUNCH((k_node_stats_bits32<MODE, true, false, WT>));
            else RLS_NS32_LAUNCH((k_node_stats_bits32<MODE, false, false, WT>));
        }
#undef RLS_NS32_LAUNCH
        return check_launch("k_node_stats_bits32");
    }
    const int has_stage = node_stats_bits_lds(N, true) <= (size_t)kLdsBytes ? 1 : 0;
    // 8 waves per tile, or 4 where that turns a launch of one-and-a-bit rounds of workgroups into ONE round: a G22-sized tile
    // is 16 KB + 4 KB of row-piece stage per wave -- 3 eight-wave workgroups per CU (768 resident: 1024 tiles = a full round
    // and a third of one), 5 four-wave ones (every tile resident at once)
    const int force_w = (int)knob(KN_NS_WAVES, 0);     // dev knob
    const int64_t tiles = ceil_div(B, kWave);
    auto resident = [&](int wv) {
        const int64_t by_lds = (int64_t)((size_t)kLdsBytes / node_stats_bits_lds(N, has_stage != 0, wv)), by_waves = 32 / wv;
        return (int64_t)num_cus() * (by_lds < by_waves ? by_lds : by_waves);
    };
    // measured (tools/timing/k7_packed.py, ls_parts.py with RLS_NS_WAVES=4 | 8; G22 2^16): K3 171 vs 167 us, weights 156 vs 139 us,
    // G70 2^17 K3 1889 vs 1656 -- the single round does not pay for halving a tile's waves: 8 stays, 4 is a knob
    (void)resident;
    const bool four = force_w == 4 && !wide && has_stage;
    const int waves = four ? 4 : kNsWaves;
    size_t lds = node_stats_bits_lds(N, has_stage != 0, waves);
    // MODE 2's parked min / max folds (the kernel's comment): 4 N bytes behind the tile, over the stages' bytes -- where they fit
    // without costing the CU a workgroup
    int stage_flags = has_stage;
    const int seeds = (int)knob(KN_NS_PARK, 16);       // dev knob: 0 = every tile folds as it goes
    if (MODE == 2 && minmax && seeds > 0 && tiles > seeds) {
        const size_t with_stash = node_stats_bits_lds(N, false, waves) + (size_t)N * 4;
        const size_t need = with_stash > lds ? with_stash : lds;
        if (need <= (size_t)kLdsBytes && (size_t)kLdsBytes / need == (size_t)kLdsBytes / lds) { lds = need; stage_flags |= 2 | (seeds << 8); }
    }
    // K2 / K3 of full tiles through row staging (the kernel's comment): 16-byte-aligned output rows of 4-node multiples, 8 KB + of
    // LDS per wave behind the tile (where the row-piece stages are: the hub groups' cooperative pass needs those)
    const int rows_knob = (int)knob(KN_NS_ROWS, -1);
    const size_t lds_rows = node_stats_bits_lds(N, false, waves) + (size_t)kNsWaves * kWave * 132;
    // measured (tools/timing/k3_rows.py, per group -> rows): K3 G70-sized 2^17 1648 -> 1574 us, BA n = 10^4 2^16 1136 -> 1032; G22-sized
    // 2^16 169 -> 178 and N = 3008 117 -> 123 (the 64 KB of staging cost the CU a workgroup there); K2's int64 rows lose everywhere
    // (G70 2274 -> 2361): K3, on tiles that leave one workgroup per CU either way
    const bool rows_auto = MODE == 1 && 2 * node_stats_bits_lds(N, true, waves) > (size_t)kLdsBytes;
    if (MODE != 2 && waves == kNsWaves && has_stage && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
        (rows_knob < 0 ? rows_auto : rows_knob != 0) && lds_rows <= (size_t)kLdsBytes) {
        stage_flags |= 4;
        if (lds_rows > lds) lds = lds_rows;
    }
    const dim3 grid((unsigned)tiles), block(waves * kWave);
#define RLS_NS_LAUNCH(KERN)                                                                                         \
    do {                                                                                                            \
        auto kern = KERN;                                                                                           \
        if (lds > 64 * 1024)                                                                                        \
            ensure_dyn_lds((const void*)kern, lds);     \
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), x, B, N, rowptr, ell_ptr, ell, mult, out, minmax, \
                           out_pitch > 0 ? out_pitch : N, stage_flags);                                             \
    } while (0)
    if (four) {
        if (vec) RLS_NS_LAUNCH((k_node_stats_bits<MODE, true, false, WT, 4>));
        else RLS_NS_LAUNCH((k_node_stats_bits<MODE, false, false, WT, 4>));
    } else if (wide) {
        if (vec) RLS_NS_LAUNCH((k_node_stats_bits<MODE, true, true, WT>));
        else RLS_NS_LAUNCH((k_node_stats_bits<MODE, false, true, WT>));
    } else {
        if (vec) RLS_NS_LAUNCH((k_node_stats_bits<MODE, true, false, WT>));
        else RLS_NS_LAUNCH((k_node_stats_bits<MODE, false, false, WT>));
    }
#undef RLS_NS_LAUNCH
    return check_launch("k_node_stats_bits");
}


// one wave per row
__global__ __launch_bounds__(256) void k_select_better_rows(uint8_t* __restrict__ xs0, int64_t* __restrict__ vs0,
                                                            const uint8_t* __restrict__ xs1,
                                                            const int64_t* __restrict__ vs1, int64_t B, int64_t N,
                                                            int if_max) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (b >= B) return;
    const int64_t v0 = vs0[b], v1 = vs1[b];
    const bool take = if_max ? (v1 >= v0) : (v1 <= v0);
    if (!take) return;
    const uint8_t* s = xs1 + b * N;
    uint8_t* d = xs0 + b * N;
    if ((N & 15) == 0 && ((((uintptr_t)xs0) | ((uintptr_t)xs1)) & 15) == 0) {
        const uint4* sv = reinterpret_cast<const uint4*>(s);
        uint4* dv = reinterpret_cast<uint4*>(d);
        for (int64_t i = lane; i < (N >> 4); i += kWave) dv[i] = sv[i];
    } else {
        for (int64_t i = lane; i < N; i += kWave) d[i] = s[i];
    }
    if (lane == 0) vs0[b] = v1;
}

__global__ __launch_bounds__(256) void k_pick_best_of_repeats(const uint8_t* __restrict__ xs,
                                                              const int64_t* __restrict__ vs, int64_t R, int64_t S,
                                                              int64_t N, int if_max, uint8_t* __restrict__ gx,
                                                              int64_t* __restrict__ gv) {
    const int lane = threadIdx.x & 63;
    const int64_t s = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (s >= S) return;
    int64_t best = vs[s];
    int64_t br = 0;
    for (int64_t r = 1; r < R; ++r) {  // first extremum wins (torch.argmax/argmin tie rule)
        const int64_t v = vs[r * S + s];
        if (if_max ? (v > best) : (v < best)) { best = v; br = r; }
    }
    const uint8_t* src = xs + (br * S + s) * N;
    uint8_t* dst = gx + s * N;
    for (int64_t i = lane; i < N; i += kWave) dst[i] = src[i];
    if (lane == 0) gv[s] = best;
}

// K14: spin(b, n) = bit (n & 127) of Philox(seed; ctr = (b_lo, b_hi, n >> 7, 'SPIN')), node 0 := 0
// A thread owns 16 spins = one 16-byte store (VEC16: rows start 16-byte aligned), consecutive lanes consecutive
// pieces, so a wave's store covers 1 KB of a row.  (One byte store per spin ran at 0.8 TB/s.)
template <bool VEC16>
// (repeat_seeds != nullptr: row b is repeat b / S of env b % S and takes that repeat's seed -- LocalSearch.reset_search's
// num_sims x num_sims candidates in one launch)
__global__ __launch_bounds__(256) void k_rand_spins(uint8_t* __restrict__ x, int64_t B, int64_t N, uint64_t seed, int64_t env_offset,
                                                    const uint64_t* __restrict__ repeat_seeds, int64_t S) {
    const int64_t chunks = (N + 15) >> 4;  // 16 spins per thread
    const int64_t total = B * chunks;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / chunks, ch = t - b * chunks;
        const int64_t rep = repeat_seeds ? b / S : 0;
        const Philox ph(repeat_seeds ? repeat_seeds[rep] : seed);
        const uint64_t gb = (uint64_t)(b - rep * S + env_offset);
        uint32_t r[4];
        ph((uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)(ch >> 3), 0x5350494Eu, r);
        uint32_t bits = (r[(ch & 7) >> 1] >> (((ch & 7) & 1) * 16)) & 0xffffu;
        if (ch == 0) bits &= ~1u;  // xs[:, 0] = 0, env_L2A.py:84
        const int64_t n0 = ch << 4;
        uint8_t* row = x + b * N;
        if constexpr (VEC16) {
            // nibble -> 4 bytes of 0/1: the multiply puts bit j at bit 8j (+ copies the mask drops)
            uint4 v;
            v.x = ((bits & 0xFu) * 0x00204081u) & 0x01010101u;
            v.y = (((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
            v.z = (((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
            v.w = (((bits >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
            *reinterpret_cast<uint4*>(row + n0) = v;
        } else {
            for (int k = 0; k < 16 && n0 + k < N; ++k) row[n0 + k] = (uint8_t)((bits >> k) & 1u);
        }
    }
}

// The same spins with ONE Philox call per 128 of them: a wave takes a row, lane l draws the row's l-th block of 128 spins,
// parks its 16 bytes in LDS, and the wave then writes the row as 16-byte pieces (lane p of trip i = piece 64 i + p, whose
// 16 bits sit in block p / 8) -- the per-piece kernel above recomputes the block's 10 Philox rounds for each of its 8
// pieces, ~100 VALU per 16 bytes written, and is bound by that.
// CH = bytes (= spins) per store: 16, or 8 / 4 for rows that are multiples of 8 / 4 bytes only (N = 1000, 3000, 5000, 7000 of the
// Gset sizes: the per-piece kernel took 158 us for 2^16 x 2000-ish rows there, this one 22)
template <int CH>
__global__ __launch_bounds__(256) void k_rand_spins_rows(uint8_t* __restrict__ x, int64_t B, int64_t N, uint64_t seed, int64_t env_offset,
                                                         const uint64_t* __restrict__ repeat_seeds, int64_t S) {
    __shared__ uint4 sh[4][kWave];
    constexpr int CPB = 128 / CH;                   // chunks per 128-spin block
    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    const int64_t chunks = N / CH;                  // N % CH == 0 here
    const int64_t blocks = (chunks + CPB - 1) / CPB;
    for (int64_t b = (int64_t)blockIdx.x * 4 + w; b < B; b += (int64_t)gridDim.x * 4) {
        const int64_t rep = repeat_seeds ? b / S : 0;
        const Philox ph(repeat_seeds ? repeat_seeds[rep] : seed);
        const uint64_t gb = (uint64_t)(b - rep * S + env_offset);
        uint8_t* row = x + b * N;
        for (int64_t g0 = 0; g0 < blocks; g0 += kWave) {
            uint32_t r[4] = {0, 0, 0, 0};
            if (g0 + lane < blocks) ph((uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)(g0 + lane), 0x5350494Eu, r);
            if (g0 + lane == 0) r[0] &= ~1u;        // xs[:, 0] = 0, env_L2A.py:84
            sh[w][lane] = make_uint4(r[0], r[1], r[2], r[3]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const uint32_t* shw = reinterpret_cast<const uint32_t*>(sh[w]);   // the 64 blocks as one array of 8192 bits
#pragma unroll
            for (int i = 0; i < CPB; ++i) {
                const int p = i * kWave + lane;     // chunk p of the window: bits [CH p, CH p + CH)
                const int64_t ch = g0 * CPB + p;
                if (ch < chunks) {
                    const uint32_t bits = (shw[(p * CH) >> 5] >> ((p * CH) & 31)) & ((1u << CH) - 1u);
                    // nibble -> 4 bytes of 0/1: the multiply puts bit j at bit 8j (+ copies the mask drops)
                    const uint32_t v0 = ((bits & 0xFu) * 0x00204081u) & 0x01010101u;
                    if constexpr (CH == 4) {
                        *reinterpret_cast<uint32_t*>(row + ch * 4) = v0;
                    } else {
                        const uint32_t v1 = (((bits >> 4) & 0xFu) * 0x00204081u) & 0x01010101u;
                        if constexpr (CH == 8) {
                            *reinterpret_cast<uint2*>(row + ch * 8) = make_uint2(v0, v1);
                        } else {
                            uint4 v;
                            v.x = v0;
                            v.y = v1;
                            v.z = (((bits >> 8) & 0xFu) * 0x00204081u) & 0x01010101u;
                            v.w = (((bits >> 12) & 0xFu) * 0x00204081u) & 0x01010101u;
                            *reinterpret_cast<uint4*>(row + ch * 16) = v;
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// The same spins, RP ROWS per wave pass (round 6): the row-per-wave kernel above runs its Philox call on `blocks` of 64 lanes (16
// at G22's 2000 nodes, 79 = 64 + 15 at G70's 10^4: its time was ten Philox rounds at a quarter / 62 % of the lanes, 0.60-0.66 of
// HBM where plain stores reach 0.84).  The RP * blocks Philox calls of a pass are dealt over the lanes as one list (RP = 64 /
// blocks for short rows, up to 4 rows of long ones: 316 calls in 5 full wave passes at N = 10^4); the RP rows are ONE contiguous
// run of RP * N bytes, written as 16 / 8 / 4-byte pieces c = 64 k + lane of that run -- whole 1 KB wave stores that do not care
// where a row ends.  Plain stores: G22 / 2^16 21.0 us vs 23.0 nontemporal (a fresh batch is the next kernel's input).
constexpr int kSpinMaxBlk = 320;     // 128-spin blocks a wave pass may hold (5 KB of LDS per wave)
template <int CH>
__global__ __launch_bounds__(256) void k_rand_spins_multi(uint8_t* __restrict__ x, int64_t B, int64_t N, uint64_t seed, int64_t env_offset,
                                                          const uint64_t* __restrict__ repeat_seeds, int64_t S, int nb, int RP) {
    __shared__ uint32_t sh[4][kSpinMaxBlk * 4];
    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    const int cpr = (int)(N / CH);                  // pieces per row (N % CH == 0 here)
    // (row, block) of lane l's first Philox call and (row, piece) of its first store, and how both move per 64 lanes
    const int br_first = lane / nb, bk_first = lane - br_first * nb, bq = kWave / nb, brem = kWave - bq * nb;
    const int r_first = lane / cpr, p_first = lane - r_first * cpr, q = kWave / cpr, rem = kWave - q * cpr;
    const int64_t npass = (B + RP - 1) / RP;
    for (int64_t ps = (int64_t)blockIdx.x * 4 + w; ps < npass; ps += (int64_t)gridDim.x * 4) {
        const int64_t b0 = ps * RP;
        const int nr = (int)((B - b0) < RP ? (B - b0) : RP);
        {
            int br = br_first, bk = bk_first;
            for (int blk = lane; blk < nr * nb; blk += kWave) {
                const int64_t b = b0 + br;
                const int64_t rep = repeat_seeds ? b / S : 0;
                const Philox ph(repeat_seeds ? repeat_seeds[rep] : seed);
                const uint64_t gb = (uint64_t)(b - rep * S + env_offset);
                uint32_t r[4];
                ph((uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)bk, 0x5350494Eu, r);
                if (bk == 0) r[0] &= ~1u;           // xs[:, 0] = 0, env_L2A.py:84
                *reinterpret_cast<uint4*>(&sh[w][blk * 4]) = make_uint4(r[0], r[1], r[2], r[3]);
                br += bq; bk += brem;
                if (bk >= nb) { bk -= nb; ++br; }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        uint8_t* run = x + b0 * N;
        const int total = nr * cpr;
        int rr = r_first, p = p_first;
        for (int c = lane; c < total; c += kWave) {
            const int bit = p * CH;                 // first spin of the piece within its row
            const uint32_t bits = (sh[w][(rr * nb + (bit >> 7)) * 4 + ((bit >> 5) & 3)] >> (bit & 31)) & ((1u << CH) - 1u);
            // nibble -> 4 bytes of 0/1: the multiply puts bit j at bit 8j (+ copies the mask drops); 24-bit operands: a full-rate multiply
            const uint32_t v0 = __umul24(bits & 0xFu, 0x00204081u) & 0x01010101u;
            if constexpr (CH == 4) {
                *reinterpret_cast<uint32_t*>(run + (int64_t)c * 4) = v0;
            } else {
                const uint32_t v1 = __umul24((bits >> 4) & 0xFu, 0x00204081u) & 0x01010101u;
                if constexpr (CH == 8) {
                    *reinterpret_cast<uint2*>(run + (int64_t)c * 8) = make_uint2(v0, v1);
                } else {
                    const uint32_t v2 = __umul24((bits >> 8) & 0xFu, 0x00204081u) & 0x01010101u;
                    const uint32_t v3 = __umul24((bits >> 12) & 0xFu, 0x00204081u) & 0x01010101u;
                    *reinterpret_cast<uint4*>(run + (int64_t)c * 16) = make_uint4(v0, v1, v2, v3);
                }
            }
            rr += q; p += rem;
            while (p >= cpr) { p -= cpr; ++rr; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void k_rand_actions(int64_t* __restrict__ action, int64_t B, int64_t N, uint64_t seed, uint64_t step,
                               int64_t env_offset) {
    const Philox ph(seed);
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t gb = (uint64_t)(b + env_offset);
        uint32_t r[4];
        ph((uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)step, (uint32_t)(step >> 32) ^ 0x41435431u, r);
        action[b] = (int64_t)(((uint64_t)r[0] * (uint64_t)N) >> 32);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Row kernels: graphs whose 64-env bit tile (8 N bytes) does not fit LDS (N > ~19 000: G81-class Gset graphs).
// One wave owns ONE env and keeps its row in LDS as plain BYTES (N bytes: up to N = 160 000 with one wave per
// workgroup), so K1 / K6 / K5 stay exact and on chip; throughput is per-env (no 64-env bit parallelism) -- a
// functional path for sizes the tile kernels cannot take, not a roofline one.
template <typename T>
__device__ __forceinline__ void row_load_bytes(const T* __restrict__ xr, int64_t N, uint8_t* row, int lane) {
    for (int64_t i = lane; i < N; i += kWave) row[i] = spin_is_set(xr[i]) ? 1 : 0;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int64_t row_cut_count(const uint8_t* row, const int32_t* __restrict__ eu, const int32_t* __restrict__ ev,
                                                 int64_t E, int lane) {
    int cnt = 0;
    for (int64_t e = lane; e < E; e += kWave) cnt += row[eu[e]] != row[ev[e]];
    return (int64_t)wave_sum_i32(cnt);
}

template <typename T>
__global__ __launch_bounds__(256) void k_maxcut_obj_rows(const T* __restrict__ x, int64_t B, int64_t N, const int32_t* __restrict__ eu,
                                                         const int32_t* __restrict__ ev, int64_t E, int halve, int64_t* __restrict__ obj) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + wib;
    if (b >= B) return;
    uint8_t* row = smem + (size_t)wib * (((size_t)N + 15) & ~(size_t)15);
    row_load_bytes<T>(x + b * N, N, row, lane);
    int64_t total = row_cut_count(row, eu, ev, E, lane);
    if (halve) total >>= 1;
    if (lane == 0) obj[b] = total;
}

__global__ __launch_bounds__(256) void k_maxcut_propose_accept_rows(uint8_t* __restrict__ x, const uint8_t* __restrict__ mask, int64_t B,
                                                                    int64_t N, const int32_t* __restrict__ eu,
                                                                    const int32_t* __restrict__ ev, int64_t E, int halve,
                                                                    int64_t* __restrict__ obj) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + wib;
    if (b >= B) return;
    uint8_t* row = smem + (size_t)wib * (((size_t)N + 15) & ~(size_t)15);
    const uint8_t* xr = x + b * N;
    const uint8_t* mr = mask + b * N;
    for (int64_t i = lane; i < N; i += kWave) row[i] = (uint8_t)((xr[i] != 0) ^ (mr[i] != 0));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int64_t total = row_cut_count(row, eu, ev, E, lane);
    if (halve) total >>= 1;
    if (total >= obj[b]) {                                               // vs1.ge(vs0), util_read_data.py:199
        for (int64_t i = lane; i < N; i += kWave) x[b * N + i] = row[i];
        if (lane == 0) obj[b] = total;
    }
}

// K5 on a byte row: strictly sequential over i = 0 .. N-1; the lanes split node i's CSR row.
template <bool WEIGHTED>
__global__ __launch_bounds__(256) void k_maxcut_greedy_sweep_rows(uint8_t* __restrict__ x, int64_t B, int64_t N,
                                                                  const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                  const int32_t* __restrict__ wgt, int halve_unused,
                                                                  int64_t* __restrict__ obj) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + wib;
    if (b >= B) return;
    uint8_t* row = smem + (size_t)wib * (((size_t)N + 15) & ~(size_t)15);
    row_load_bytes<uint8_t>(x + b * N, N, row, lane);
    int64_t gain_total = 0;
    for (int64_t i = 0; i < N; ++i) {
        const int r0 = rowptr[i], r1 = rowptr[i + 1];
        const uint8_t xi = row[i];
        int acc = 0;
        for (int j = r0 + lane; j < r1; j += kWave) {
            const bool same = row[col[j]] == xi;
            if constexpr (WEIGHTED) acc += same ? wgt[j] : -wgt[j];
            else acc += same ? 1 : -1;
        }
        const int gain = wave_sum_i32(acc);                               // cut(after the flip) - cut(before)
        if (gain >= 0) {                                                  // ties accept (update_xs_by_vs uses ge)
            if (lane == 0) row[i] = xi ^ 1;
            gain_total += gain;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    }
    for (int64_t i = lane; i < N; i += kWave) x[b * N + i] = row[i];
    if (lane == 0) obj[b] += gain_total;
}

static inline int rows_waves(int64_t N) {   // waves (= envs) per workgroup of a row kernel, 0 when even one row does not fit
    const size_t per = ((size_t)N + 15) & ~(size_t)15;
    if (per > (size_t)kLdsBytes) return 0;
    int w = (int)((size_t)kLdsBytes / per);
    return w > 4 ? 4 : w;
}

}  // namespace rls

using namespace rls;

// ---- narrow-tile launches (16 or 8 envs per workgroup: rls_tile32.h).  `want`: 0 = the widest that fits, 16 / 8 = that one.
// kNarrowNo: not applicable here (nothing was launched, no error recorded).
constexpr int kNarrowNo = 1;
// Which tile a launch of B envs should take when several fit: a tile's load / sweep / store is a chain of its own whatever it holds,
// so a batch of few tiles wants NARROWER ones until the chip is full -- 4096 envs are 64 / 128 / 256 / 512 tiles of 64 / 32 / 16 / 8
// envs on 256 CUs.  Returns 0 (the wide tiles), 16 or 8.  RLS_NARROW_TILE: 0 never, 1 automatic, 2 / 3 force 16 / 8 where they fit.
static int narrow_policy(int64_t N, int64_t B, bool rows_vec, Knob wide_knob) {
    const int64_t k = knob(KN_NARROW_TILE, 1);
    if (k == 0 || (k == 1 && knob(wide_knob, -1) >= 0)) return 0;      // (a forced 64-env / half-tile form is honoured)
    if (k == 2) return 16;
    if (k == 3) return 8;
    // automatic (tools/timing/narrow_policy.py, K1 / K6 / K5 in us, wide -> narrow): N = 10^4, 4096 envs 21.8 / 31.6 / 47.6 -> 11.7 /
    // 22.4 / 39.2 (16 envs), 256 envs 19.4 / 27.8 / 42.4 -> 7.6 / 13.2 / 24.9 (8 envs); N = 39 936, 4096 envs 47 / 120 / 356 -> 36 / 86 /
    // 135; N = 2000, 4096 envs 10.4 / 11.6 / 47 -> 7.1 / 9.1 / 46; N = 800: nothing.  From 16 384 envs on the wide tiles win
    // (N = 10^4: 29.7 / 77.5 / 69.8 vs 32.8 / 89.6 / 110) -- the chip is full and a narrow tile walks the edge list / schedule per 16 envs.
    if (!rows_vec || (N & 7) != 0 || N < 1536) return 0;      // (the narrow loader's fast path: byte rows of 16- or 8-byte multiples)
    const int64_t cus = num_cus();
    if (ceil_div(B, 8) <= cus) return 8;
    if (ceil_div(B, 16) <= cus) return 16;
    return 0;
}

static int launch_obj_narrow(const rls_graph* g, const void* x, int spin_bytes, int64_t B, int64_t* obj, int want, void* stream) {
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    const int Pn = pick_planes(E) == 12 ? 16 : pick_planes(E);
    const bool vecn = tile_rows_aligned(x, N, spin_bytes);
    const int hv = g->if_bidirectional ? 1 : 0;
    const size_t l16 = narrow_words_bytes<uint16_t>(N) + (size_t)kNarrowWaves * kWave * 8, l8 = narrow_words_bytes<uint8_t>(N) + (size_t)kNarrowWaves * kWave * 8;
    if (Pn == 0 || l8 > (size_t)kLdsBytes) return kNarrowNo;
    const bool w16 = want != 8 && l16 <= (size_t)kLdsBytes;
    const size_t ln = w16 ? l16 : l8;
    const dim3 gn((unsigned)ceil_div(B, (int64_t)(w16 ? 16 : 8))), bn(kNarrowWaves * kWave);
#define LAUNCH_OBJN(T, VEC, PP)                                                                                         \
    do {                                                                                                                \
        if (w16) { auto kern = k_maxcut_obj_n<T, VEC, PP, uint16_t>; ensure_dyn_lds((const void*)kern, ln);             \
                   hipLaunchKernelGGL(kern, gn, bn, ln, as_stream(stream), (const T*)x, B, N, g->eu, g->ev, E, hv, obj); } \
        else     { auto kern = k_maxcut_obj_n<T, VEC, PP, uint8_t>; ensure_dyn_lds((const void*)kern, ln);              \
                   hipLaunchKernelGGL(kern, gn, bn, ln, as_stream(stream), (const T*)x, B, N, g->eu, g->ev, E, hv, obj); } \
    } while (0)
#define DISPATCH_PN(T, VEC)                        \
    switch (Pn) {                                  \
        case 16: LAUNCH_OBJN(T, VEC, 16); break;   \
        case 20: LAUNCH_OBJN(T, VEC, 20); break;   \
        default: LAUNCH_OBJN(T, VEC, 24); break;   \
    }
    if (spin_bytes == 1) { if (vecn) { DISPATCH_PN(uint8_t, true) } else { DISPATCH_PN(uint8_t, false) } }
    else { DISPATCH_PN(float, false) }
#undef DISPATCH_PN
#undef LAUNCH_OBJN
    return check_launch("k_maxcut_obj_n");
}

static int launch_propose_accept_narrow(const rls_graph* g, uint8_t* x, int64_t B, const uint8_t* mask, int32_t mask_bits, int64_t* obj, int want,
                                        void* stream) {
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    const int Pn = pick_planes(E) == 12 ? 16 : pick_planes(E);
    const bool vecn = tile_rows_aligned(x, N, 1) && (mask_bits || tile_rows_aligned(mask, N, 1));
    const int hv = g->if_bidirectional ? 1 : 0;
    const size_t l16 = narrow_words_bytes<uint16_t>(N) + (size_t)kNarrowWaves * kWave * 8, l8 = narrow_words_bytes<uint8_t>(N) + (size_t)kNarrowWaves * kWave * 8;
    if (Pn == 0 || l8 > (size_t)kLdsBytes) return kNarrowNo;
    const bool w16 = want != 8 && l16 <= (size_t)kLdsBytes;
    const size_t ln = w16 ? l16 : l8;
    const dim3 gn((unsigned)ceil_div(B, (int64_t)(w16 ? 16 : 8))), bn(kNarrowWaves * kWave);
#define LAUNCH_PAN(VEC, PP, MB)                                                                                          \
    do {                                                                                                                \
        if (w16) { auto kern = k_maxcut_propose_accept_n<VEC, PP, uint16_t, MB>; ensure_dyn_lds((const void*)kern, ln); \
                   hipLaunchKernelGGL(kern, gn, bn, ln, as_stream(stream), x, mask, B, N, g->eu, g->ev, E, hv, obj); }  \
        else     { auto kern = k_maxcut_propose_accept_n<VEC, PP, uint8_t, MB>; ensure_dyn_lds((const void*)kern, ln);  \
                   hipLaunchKernelGGL(kern, gn, bn, ln, as_stream(stream), x, mask, B, N, g->eu, g->ev, E, hv, obj); }  \
    } while (0)
#define DISPATCH_PN(VEC, MB)                       \
    switch (Pn) {                                  \
        case 16: LAUNCH_PAN(VEC, 16, MB); break;   \
        case 20: LAUNCH_PAN(VEC, 20, MB); break;   \
        default: LAUNCH_PAN(VEC, 24, MB); break;   \
    }
    if (mask_bits) { if (vecn) { DISPATCH_PN(true, true) } else { DISPATCH_PN(false, true) } }
    else { if (vecn) { DISPATCH_PN(true, false) } else { DISPATCH_PN(false, false) } }
#undef DISPATCH_PN
#undef LAUNCH_PAN
    return check_launch("k_maxcut_propose_accept_n");
}

static int launch_sweep_narrow(const rls_graph* g, uint8_t* x, int64_t B, int64_t* obj, int want, void* stream) {
    const int64_t N = g->num_nodes, E = g->num_stored_edges, G = g->num_sweep_groups;
    if (g->wgt || !g->sweep_lv_ptr || !g->sweep_lv_data || G <= 0 || knob_on(KN_SWEEP_NO_LEVELS)) return kNarrowNo;
    const int Pn = pick_planes(E) == 12 ? 16 : pick_planes(E);
    const bool vec = tile_rows_aligned(x, N, 1);
    auto ldsn_of = [&](size_t wb) { return wb + (((size_t)(G + 1) * 4 + 15) & ~(size_t)15) + (size_t)kNarrowWaves * kWave * 8; };
    const size_t l16 = ldsn_of(narrow_words_bytes<uint16_t>(N)), l8 = ldsn_of(narrow_words_bytes<uint8_t>(N));
    if (Pn == 0 || l8 > (size_t)kLdsBytes) return kNarrowNo;
    const bool w16 = want != 8 && l16 <= (size_t)kLdsBytes;
    const size_t ln = w16 ? l16 : l8;
    const dim3 gn((unsigned)ceil_div(B, (int64_t)(w16 ? 16 : 8))), bn(kNarrowWaves * kWave);
    const int hv = g->if_bidirectional ? 1 : 0;
    hipStream_t s = as_stream(stream);
#define LAUNCH_SWLN(VEC, PP)                                                                                                       \
    do {                                                                                                                           \
        if (w16) { auto kern = k_maxcut_greedy_sweep_levels_n<VEC, PP, uint16_t>; ensure_dyn_lds((const void*)kern, ln);           \
                   hipLaunchKernelGGL(kern, gn, bn, ln, s, x, B, N, g->sweep_lv_ptr, g->sweep_lv_data, G, g->eu, g->ev, E, hv, obj); } \
        else     { auto kern = k_maxcut_greedy_sweep_levels_n<VEC, PP, uint8_t>; ensure_dyn_lds((const void*)kern, ln);            \
                   hipLaunchKernelGGL(kern, gn, bn, ln, s, x, B, N, g->sweep_lv_ptr, g->sweep_lv_data, G, g->eu, g->ev, E, hv, obj); } \
    } while (0)
#define DISPATCH_SWLN(VEC)                        \
    switch (Pn) {                                 \
        case 16: LAUNCH_SWLN(VEC, 16); break;     \
        case 20: LAUNCH_SWLN(VEC, 20); break;     \
        default: LAUNCH_SWLN(VEC, 24); break;     \
    }
    if (vec) { DISPATCH_SWLN(true) } else { DISPATCH_SWLN(false) }
#undef DISPATCH_SWLN
#undef LAUNCH_SWLN
    return check_launch("k_maxcut_greedy_sweep_levels_n");
}

extern "C" {

#ifdef RLS_PROF
// dev builds only (RLS_EXTRA_CFLAGS=-DRLS_PROF): read (reset != 0: clear) the phase counters of this translation unit
int rls_dev_prof(unsigned long long* out, int reset) {
    if (reset) {
        unsigned long long z[8] = {0, 0, 0, 0, 0, ~0ull, 0, 0};
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z));
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 8);
}
int rls_dev_prof_waves(unsigned long long* out, int64_t nwaves) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof_w), sizeof(unsigned long long) * 5 * (size_t)nwaves);
}
#endif

int rls_maxcut_obj(const rls_graph* g, const void* x, int spin_bytes, int64_t B, int64_t* obj, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x && obj, RLS_EINVAL, "x/obj is NULL");
    RLS_REQUIRE(spin_bytes == 1 || spin_bytes == 4, RLS_EINVAL, "spin_bytes must be 1 or 4");
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    if (const int nw = narrow_policy(N, B, tile_rows_aligned(x, N, spin_bytes), KN_K1_TILE32))
        if (const int rc = launch_obj_narrow(g, x, spin_bytes, B, obj, nw, stream); rc != kNarrowNo) return rc;
    int tw = tile_waves_for(N);
    size_t lds = (size_t)N * 8 + (size_t)tw * kWave * 8;
    if (lds > (size_t)kLdsBytes && (size_t)N * 8 + (size_t)kTileWaves * kWave * 8 <= (size_t)kLdsBytes) {
        // the tile alone still fits (N <= 20 224: Gset's 20 000-node G81): 4 waves, no row-piece stage (lane-per-env loads at
        // ~2.5 TB/s) -- an order of magnitude ahead of the one-env-per-wave form below
        tw = kTileWaves;
        lds = (size_t)N * 8 + (size_t)tw * kWave * 8;
    }
    // Half tiles (32 envs, 32-bit words: rls_tile32.h).  Where the 64-env tile does not fit but N * 4 bytes do (20 224 < N <=
    // 40 448) -- and, for byte rows of 16-byte multiples (their fast loader), where they measure faster (tools/timing/k1_tile32.py):
    // rows past 8192 nodes, whose 64-env tile leaves one workgroup per CU or no room for the row-piece stage (G70-sized 2^17:
    // 280 -> 262 us, N = 20 000 2^16: 374 -> 282), and launches of at most two 64-env tiles per CU (G22-sized 2^14: 14.8 -> 13.4 us;
    // at 2^16 the half tiles LOSE, 34.4 -> 36.3: twice the edge-list reads per env).  Dev knob RLS_K1_TILE32 = 0 | 1 forces the choice.
    const int knob32 = (int)knob(KN_K1_TILE32, -1);
    {
        auto lds32 = [&](int ww) { return (((size_t)N * 4 + 15) & ~(size_t)15) + (size_t)ww * kWave * 8; };
        // 8 waves once two 4-wave workgroups (with their stages) no longer share a CU: one 4-wave workgroup per CU cannot keep
        // enough loads in flight (N = 15 984 .. 16 384 ran at 0.43 of HBM beside 0.61 at 15 872, where two still fit)
        int w32 = 2 * (lds32(kTileWaves) + (size_t)kTileWaves * kStageBytes) > (size_t)kLdsBytes ? kTileWavesMax : kTileWaves;
        if (lds32(w32) > (size_t)kLdsBytes) w32 = kTileWaves;
        size_t l32 = lds32(w32);
        const int P32 = pick_planes(E);
        const bool vec = tile_rows_aligned(x, N, spin_bytes);
        const bool fast32 = spin_bytes == 1 && vec && (N & 7) == 0;      // (rows of 16- or 8-byte multiples: rls_tile32.h)
        const bool want32 = knob32 >= 0 ? knob32 != 0
                                        : fast32 && (N >= 3000 || ceil_div(B, kWave) <= 2 * (int64_t)num_cus());
        // (N >= 3000, round 5: until then "rows past 8192 nodes" -- at 2^16 envs the half tile also wins from 3000 nodes on, K1 N = 3008 /
        // 5008 / 7008: 42.6 / 77.5 / 103.8 -> 38.6 / 75.1 / 102.2 us, rows of 8-byte multiples 49.8 / 89.4 / 124.0 -> 43.1 / 83.1 / 116.3;
        // at G22's 2000 it loses, 36.3 vs 34.4: tools/sweeps/align_sweep.py with RLS_K1_TILE32 = 1)
        if ((want32 || lds > (size_t)kLdsBytes) && l32 <= (size_t)kLdsBytes && P32 != 0) {
            const int st_off = tile_stage_offset(&l32, w32, spin_bytes == 1 && vec && (N & 7) == 0);
            const dim3 g32((unsigned)ceil_div(B, (int64_t)kHalf)), b32(w32 * kWave);
            const int hv = g->if_bidirectional ? 1 : 0;
            hipStream_t s32 = as_stream(stream);
#define LAUNCH_OBJ32(T, VEC, PP)                                                                                        \
    do {                                                                                                                \
        auto kern = w32 == kTileWavesMax ? k_maxcut_obj32<T, VEC, PP, kTileWavesMax> : k_maxcut_obj32<T, VEC, PP, kTileWaves>; \
        if (l32 > 64 * 1024) ensure_dyn_lds((const void*)kern, l32); \
        hipLaunchKernelGGL(kern, g32, b32, l32, s32, (const T*)x, B, N, g->eu, g->ev, E, hv, obj, st_off);              \
    } while (0)
#define DISPATCH_P32(T, VEC)                        \
    switch (P32) {                                  \
        case 12: LAUNCH_OBJ32(T, VEC, 12); break;   \
        case 16: LAUNCH_OBJ32(T, VEC, 16); break;   \
        case 20: LAUNCH_OBJ32(T, VEC, 20); break;   \
        default: LAUNCH_OBJ32(T, VEC, 24); break;   \
    }
            if (spin_bytes == 1) {
                if (vec) { DISPATCH_P32(uint8_t, true) } else { DISPATCH_P32(uint8_t, false) }
            } else {
                DISPATCH_P32(float, false)
            }
#undef DISPATCH_P32
#undef LAUNCH_OBJ32
            return check_launch("k_maxcut_obj32");
        }
    }
    if (lds > (size_t)kLdsBytes && knob(KN_NARROW_TILE, 1) != 0)      // neither the 64-env nor the half tile fits: 16 or 8 envs per workgroup
        if (const int rc = launch_obj_narrow(g, x, spin_bytes, B, obj, 0, stream); rc != kNarrowNo) return rc;
    if (lds > (size_t)kLdsBytes) {   // neither tile fits: one env per wave on a byte row
        const int rw = rows_waves(N);
        RLS_REQUIRE(rw > 0, RLS_EUNSUPPORTED, "N=%lld: a row of %lld bytes does not fit LDS (max %d)", (long long)N, (long long)N, kLdsBytes);
        const size_t lr = (size_t)rw * (((size_t)N + 15) & ~(size_t)15);
        const dim3 gr((unsigned)ceil_div(B, rw)), br(rw * kWave);
        const int hv = g->if_bidirectional ? 1 : 0;
        if (spin_bytes == 1) {
            if (lr > 64 * 1024) ensure_dyn_lds((const void*)k_maxcut_obj_rows<uint8_t>, lr);
            hipLaunchKernelGGL(k_maxcut_obj_rows<uint8_t>, gr, br, lr, as_stream(stream), (const uint8_t*)x, B, N, g->eu, g->ev, E, hv, obj);
        } else {
            if (lr > 64 * 1024) ensure_dyn_lds((const void*)k_maxcut_obj_rows<float>, lr);
            hipLaunchKernelGGL(k_maxcut_obj_rows<float>, gr, br, lr, as_stream(stream), (const float*)x, B, N, g->eu, g->ev, E, hv, obj);
        }
        return check_launch("k_maxcut_obj_rows");
    }
    const int P = pick_planes(E);
    RLS_REQUIRE(P != 0, RLS_EUNSUPPORTED, "E'=%lld too large", (long long)E);
    const bool vec = tile_rows_aligned(x, N, spin_bytes);
    const int stage_off = tile_stage_offset(&lds, tw, spin_bytes == 1);   // (unaligned byte rows use it too)
    const dim3 grid((unsigned)ceil_div(B, kWave)), block(tw * kWave);
    hipStream_t s = as_stream(stream);
    const int halve = g->if_bidirectional ? 1 : 0;
    // dev knob: ask for more LDS than the tile needs, i.e. fewer resident workgroups per CU and a second round of them whose loads
    // could hide the first round's counting (RLS_K1_LDS_KB = kilobytes per workgroup)
    const int pad_kb = (int)knob(KN_K1_LDS_KB, 0);
    if (pad_kb > 0 && (size_t)pad_kb * 1024 > lds && (size_t)pad_kb * 1024 <= (size_t)kLdsBytes) lds = (size_t)pad_kb * 1024;
#define LAUNCH_OBJ(T, VEC, PP)                                                                             \
    do {                                                                                                   \
        auto kern = tw == kTileWavesMax ? k_maxcut_obj<T, VEC, PP, kTileWavesMax> : k_maxcut_obj<T, VEC, PP, kTileWaves>; \
        if (lds > 64 * 1024)                                                                               \
            ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, (const T*)x, B, N, g->eu, g->ev, E, halve, obj, stage_off); \
    } while (0)
#define DISPATCH_P(T, VEC)                       \
    switch (P) {                                 \
        case 12: LAUNCH_OBJ(T, VEC, 12); break;  \
        case 16: LAUNCH_OBJ(T, VEC, 16); break;  \
        case 20: LAUNCH_OBJ(T, VEC, 20); break;  \
        default: LAUNCH_OBJ(T, VEC, 24); break;  \
    }
    if (spin_bytes == 1) {
        if (vec) { DISPATCH_P(uint8_t, true) } else { DISPATCH_P(uint8_t, false) }
    } else {
        if (vec) { DISPATCH_P(float, true) } else { DISPATCH_P(float, false) }
    }
#undef DISPATCH_P
#undef LAUNCH_OBJ
    return check_launch("k_maxcut_obj");
}

int rls_maxcut_propose_accept(const rls_graph* g, uint8_t* x, int64_t B, const void* mask_v, int32_t mask_bits, int64_t* obj,
                              void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    if (B == 0) return RLS_OK;
    const uint8_t* mask = static_cast<const uint8_t*>(mask_v);
    RLS_REQUIRE(x && mask && obj, RLS_EINVAL, "x/mask/obj is NULL");
    RLS_REQUIRE(!mask_bits || (((uintptr_t)mask) & 7) == 0, RLS_EINVAL, "a bit-packed mask is uint64 words: 8-byte aligned");
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    if (const int nw = narrow_policy(N, B, tile_rows_aligned(x, N, 1) && (mask_bits || tile_rows_aligned(mask, N, 1)), KN_K6_TILE32))
        if (const int rc = launch_propose_accept_narrow(g, x, B, mask, mask_bits, obj, nw, stream); rc != kNarrowNo) return rc;
    int tw = tile_waves_for(N);
    size_t lds = (size_t)N * 8 + (size_t)tw * kWave * 8;
    if (lds > (size_t)kLdsBytes && (size_t)N * 8 + (size_t)kTileWaves * kWave * 8 <= (size_t)kLdsBytes) {
        tw = kTileWaves;              // (as in rls_maxcut_obj: the tile alone fits, 4 waves without the row-piece stage)
        lds = (size_t)N * 8 + (size_t)tw * kWave * 8;
    }
    // Half tiles (rls_tile32.h): where the 64-env tile does not fit and N * 4 bytes do, for rows past 8192 nodes (G70-sized 2^17,
    // byte mask: 608 -> 481 us; N = 20 000 2^15, where the 64-env tile has no room for its stage: 387 -> 299), and in launches of
    // at most one 64-env tile per CU (G22-sized 2^12: 21.8 -> 11.4 us, 2^14: 26.8 -> 20.9; at 2^16 they lose, 63.6 -> 67.3).
    // tools/timing/k5_tile32.py.
    const int knob32 = (int)knob(KN_K6_TILE32, -1);   // dev knob: 0 | 1 forces the choice
    const bool no_stage64 = lds + (size_t)tw * kStageBytes > (size_t)kLdsBytes;
    // (the half tile's fast loader wants byte rows of 16-byte multiples on a 16-byte base; other rows keep the 64-env forms)
    const bool fast32 = (N & 7) == 0 && tile_rows_aligned(x, N, 1) && (mask_bits || tile_rows_aligned(mask, N, 1));
    const bool want32 = knob32 >= 0 ? knob32 != 0
                                    : fast32 && (no_stage64 || N >= 3000 || ceil_div(B, kWave) <= (int64_t)num_cus());
    // (N >= 3000, round 5, as for K1: K6 with a byte mask at 2^16 envs, N = 3008 / 5008 / 7008: 101 / 173 / 236 -> 95 / 158 / 212 us,
    // rows of 8-byte multiples 116 / 194 / 271 -> 101 / 165 / 236)
    if (want32 || lds > (size_t)kLdsBytes) {
        int w32 = kTileWavesMax;
        auto lds32 = [&](int ww) { return (((size_t)N * 4 + 15) & ~(size_t)15) + (size_t)ww * kWave * 8; };
        if (lds32(w32) > (size_t)kLdsBytes) w32 = kTileWaves;
        size_t l32 = lds32(w32);
        const int P32 = pick_planes(E);
        if (l32 <= (size_t)kLdsBytes && P32 != 0) {
            const bool vec = tile_rows_aligned(x, N, 1) && (mask_bits || tile_rows_aligned(mask, N, 1));
            const int st32 = tile_stage_offset(&l32, w32, vec && (N & 7) == 0);    // (row-piece stages when they fit beside the tile)
            const dim3 g32((unsigned)ceil_div(B, (int64_t)kHalf)), b32(w32 * kWave);
            hipStream_t s32 = as_stream(stream);
            const int hv = g->if_bidirectional ? 1 : 0;
#define LAUNCH_PA32(VEC, PP)                                                                                            \
    do {                                                                                                                \
        auto kern = mask_bits ? (w32 == kTileWavesMax ? k_maxcut_propose_accept32<VEC, PP, kTileWavesMax, true>         \
                                                      : k_maxcut_propose_accept32<VEC, PP, kTileWaves, true>)           \
                              : (w32 == kTileWavesMax ? k_maxcut_propose_accept32<VEC, PP, kTileWavesMax, false>        \
                                                      : k_maxcut_propose_accept32<VEC, PP, kTileWaves, false>);         \
        if (l32 > 64 * 1024) ensure_dyn_lds((const void*)kern, l32); \
        hipLaunchKernelGGL(kern, g32, b32, l32, s32, x, mask, B, N, g->eu, g->ev, E, hv, obj, st32);                    \
    } while (0)
#define DISPATCH_P32(VEC)                      \
    switch (P32) {                             \
        case 12: LAUNCH_PA32(VEC, 12); break;  \
        case 16: LAUNCH_PA32(VEC, 16); break;  \
        case 20: LAUNCH_PA32(VEC, 20); break;  \
        default: LAUNCH_PA32(VEC, 24); break;  \
    }
            if (vec) { DISPATCH_P32(true) } else { DISPATCH_P32(false) }
#undef DISPATCH_P32
#undef LAUNCH_PA32
            return check_launch("k_maxcut_propose_accept32");
        }
    }
    if (lds > (size_t)kLdsBytes && knob(KN_NARROW_TILE, 1) != 0)      // neither the 64-env nor the half tile fits: 16 or 8 envs per workgroup
        if (const int rc = launch_propose_accept_narrow(g, x, B, mask, mask_bits, obj, 0, stream); rc != kNarrowNo) return rc;
    if (lds > (size_t)kLdsBytes) {   // neither tile fits: one env per wave on a byte row
        const int rw = rows_waves(N);
        RLS_REQUIRE(rw > 0, RLS_EUNSUPPORTED, "N=%lld: a row does not fit LDS (max %d)", (long long)N, kLdsBytes);
        RLS_REQUIRE(!mask_bits, RLS_EUNSUPPORTED, "N=%lld: beyond the tiles the mask must be bytes [B, N]", (long long)N);
        const size_t lr = (size_t)rw * (((size_t)N + 15) & ~(size_t)15);
        if (lr > 64 * 1024) ensure_dyn_lds((const void*)k_maxcut_propose_accept_rows, lr);
        hipLaunchKernelGGL(k_maxcut_propose_accept_rows, dim3((unsigned)ceil_div(B, rw)), dim3(rw * kWave), lr, as_stream(stream), x, mask, B, N,
                           g->eu, g->ev, E, g->if_bidirectional ? 1 : 0, obj);
        return check_launch("k_maxcut_propose_accept_rows");
    }
    const int P = pick_planes(E);
    RLS_REQUIRE(P != 0, RLS_EUNSUPPORTED, "E'=%lld too large", (long long)E);
    const bool vec = tile_rows_aligned(x, N, 1) && (mask_bits || tile_rows_aligned(mask, N, 1));
    const int stage_off = tile_stage_offset(&lds, tw, true);
    const dim3 grid((unsigned)ceil_div(B, kWave)), block(tw * kWave);
    hipStream_t s = as_stream(stream);
    const int halve = g->if_bidirectional ? 1 : 0;
#define LAUNCH_PA(VEC, PP)                                                                                 \
    do {                                                                                                   \
        auto kern = mask_bits ? (tw == kTileWavesMax ? k_maxcut_propose_accept<VEC, PP, kTileWavesMax, true>   \
                                                     : k_maxcut_propose_accept<VEC, PP, kTileWaves, true>)     \
                              : (tw == kTileWavesMax ? k_maxcut_propose_accept<VEC, PP, kTileWavesMax>         \
                                                     : k_maxcut_propose_accept<VEC, PP, kTileWaves>);          \
        if (lds > 64 * 1024)                                                                               \
            ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, x, mask, B, N, g->eu, g->ev, E, halve, obj, stage_off); \
    } while (0)
#define DISPATCH_P(VEC)                      \
    switch (P) {                             \
        case 12: LAUNCH_PA(VEC, 12); break;  \
        case 16: LAUNCH_PA(VEC, 16); break;  \
        case 20: LAUNCH_PA(VEC, 20); break;  \
        default: LAUNCH_PA(VEC, 24); break;  \
    }
    if (vec) { DISPATCH_P(true) } else { DISPATCH_P(false) }
#undef DISPATCH_P
#undef LAUNCH_PA
    return check_launch("k_maxcut_propose_accept");
}

int rls_maxcut_greedy_sweep(const rls_graph* g, uint8_t* x, int64_t B, int64_t* obj, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x && obj, RLS_EINVAL, "x/obj is NULL");
    const int64_t N = g->num_nodes;
    const bool vec = tile_rows_aligned(x, N, 1);
    if (const int nw = narrow_policy(N, B, vec, KN_K5_TILE32))
        if (const int rc = launch_sweep_narrow(g, x, B, obj, nw, stream); rc != kNarrowNo) return rc;
    const dim3 grid((unsigned)ceil_div(B, kWave)), block(kWave);
    hipStream_t s = as_stream(stream);
    const size_t lds_fast = (size_t)(N + 2) * 8 + (size_t)((N + 1 + 3) & ~3ll) * 4 + (size_t)kRing * 4;
    const bool fast = !g->wgt && g->max_degree < kSweepMaxDeg && lds_fast <= (size_t)kLdsBytes &&
                      (((uintptr_t)g->col) & 3) == 0;
    {   // level-parallel sweep: needs the lane-per-node schedule (N < 2^20, degrees < 256) and the tile in LDS
        const bool no_levels = knob_on(KN_SWEEP_NO_LEVELS);   // dev knob
        const int64_t G = g->num_sweep_groups, E = g->num_stored_edges;
        const int P = pick_planes(E);
        const int force_lw = (int)knob(KN_SWEEP_WAVES, 0);
        // one group per level (G22: 44 nodes per group): a level is ONE wave's pass and the others only prefetch -- few
        // waves, more tiles per CU; well-filled groups (G70: 9 levels of ~17 groups): 8 waves share a level
        int sw = force_lw == 2 || force_lw == 4 || force_lw == 8 || force_lw == 16 ? force_lw : (N >= 56 * G ? 8 : 4);
        auto lds_of = [&](int waves, bool stage) {
            return (size_t)(N + 2) * 8 + (((size_t)(G + 1) * 4 + 15) & ~(size_t)15) + (size_t)waves * kWave * 8 +
                   (stage ? (size_t)kSweepLoadWaves * kStageBytes : 0);
        };
        int has_stage = 1;
        if (lds_of(sw, true) > (size_t)kLdsBytes) {   // the tile nearly fills LDS (N ~ 20 000): no row-piece stage, fewer waves
            has_stage = 0;
            if (lds_of(sw, false) > (size_t)kLdsBytes) sw = 4;
            if (lds_of(sw, false) > (size_t)kLdsBytes) sw = 2;
        }
        const size_t lds_l = lds_of(sw, has_stage != 0);
        // half tiles (rls_tile32.h) where the 64-env tile does not fit (dev knob RLS_K5_TILE32 = 1: at any size)
        const int knob32 = (int)knob(KN_K5_TILE32, -1);
        // ... and, for byte rows of 16-byte multiples (the half tile's staged loader / store), where they measure faster
        // (tools/timing/k5_tile32.py): rows past 8192 nodes (G70-sized 2^17: 780 -> 705 us; N = 20 000, where the 64-env tile has no
        // room for its stage, 4096 envs: 171 -> 100) and launches of at most one 64-env tile per CU (G22-sized 2^14: 63 -> 53 us;
        // at 2^16 the half tiles lose, 113 -> 128: twice the schedule reads per env)
        const bool prefer32 = knob32 < 0 && vec && (N & 7) == 0 &&
                              (!has_stage || (size_t)N * 8 > 64 * 1024 || ceil_div(B, kWave) <= (int64_t)num_cus());
        if ((knob32 > 0 || prefer32 || lds_l > (size_t)kLdsBytes) && !no_levels && !g->wgt && g->sweep_lv_ptr && g->sweep_lv_data && G > 0 && P != 0) {
            int sw32 = force_lw == 2 || force_lw == 4 || force_lw == 8 ? force_lw : (N >= 56 * G ? 8 : 4);
            auto lds32_of = [&](int waves, bool stage) {
                return (((size_t)(N + 2) * 4 + 15) & ~(size_t)15) + (((size_t)(G + 1) * 4 + 15) & ~(size_t)15) + (size_t)waves * kWave * 8 +
                       (stage ? (size_t)kSweepLoadWaves * kStageBytes : 0);
            };
            int stage32 = vec && (N & 7) == 0 ? 1 : 0;
            if (stage32 && lds32_of(sw32, true) > (size_t)kLdsBytes) stage32 = 0;
            if (lds32_of(sw32, stage32 != 0) > (size_t)kLdsBytes) sw32 = 4;
            if (lds32_of(sw32, stage32 != 0) > (size_t)kLdsBytes) sw32 = 2;
            // a half tile that leaves room for two waves only (N ~ 40 000: the words fill LDS) sweeps slower than 16-env tiles with
            // eight at every batch size (N = 39 936: 2^12 envs 356 -> 135 us, 2^16 3779 -> 2332)
            if (sw32 == 2 && knob(KN_NARROW_TILE, 1) == 1 && knob32 < 0)
                if (const int rc = launch_sweep_narrow(g, x, B, obj, 16, stream); rc != kNarrowNo) return rc;
            const size_t l32 = lds32_of(sw32, stage32 != 0);
            if (l32 <= (size_t)kLdsBytes) {
                const dim3 g32((unsigned)ceil_div(B, (int64_t)kHalf)), b32(sw32 * kWave);
                const int hv = g->if_bidirectional ? 1 : 0;
#define LAUNCH_SWL32(VEC, SWV, PP)                                                                                          \
    do {                                                                                                                    \
        auto kern = k_maxcut_greedy_sweep_levels32<VEC, SWV, PP>;                                                           \
        if (l32 > 64 * 1024) ensure_dyn_lds((const void*)kern, l32); \
        hipLaunchKernelGGL(kern, g32, b32, l32, s, x, B, N, g->sweep_lv_ptr, g->sweep_lv_data, G, g->eu, g->ev, E, hv, obj, stage32); \
    } while (0)
#define DISPATCH_SWL32_P(VEC, SWV)                       \
    switch (P) {                                         \
        case 12: LAUNCH_SWL32(VEC, SWV, 12); break;      \
        case 16: LAUNCH_SWL32(VEC, SWV, 16); break;      \
        case 20: LAUNCH_SWL32(VEC, SWV, 20); break;      \
        default: LAUNCH_SWL32(VEC, SWV, 24); break;      \
    }
#define DISPATCH_SWL32(VEC)                                     \
    do {                                                        \
        if (sw32 == 4) { DISPATCH_SWL32_P(VEC, 4) }             \
        else if (sw32 == 2) { DISPATCH_SWL32_P(VEC, 2) }        \
        else { DISPATCH_SWL32_P(VEC, 8) }                       \
    } while (0)
                if (vec) DISPATCH_SWL32(true); else DISPATCH_SWL32(false);
#undef DISPATCH_SWL32
#undef DISPATCH_SWL32_P
#undef LAUNCH_SWL32
                return check_launch("k_maxcut_greedy_sweep_levels32");
            }
            // the half tile does not fit either: 16 or 8 envs per workgroup (the same level schedule)
            if (knob(KN_NARROW_TILE, 1) != 0)
                if (const int rc = launch_sweep_narrow(g, x, B, obj, 0, stream); rc != kNarrowNo) return rc;
        }
        if (!no_levels && !g->wgt && g->sweep_lv_ptr && g->sweep_lv_data && G > 0 && P != 0 && lds_l <= (size_t)kLdsBytes) {
            const dim3 blockl(sw * kWave);
            const int halve = g->if_bidirectional ? 1 : 0;
#define LAUNCH_SWL(VEC, SWV, PP)                                                                             \
    do {                                                                                                     \
        auto kern = k_maxcut_greedy_sweep_levels<VEC, SWV, PP>;                                              \
        if (lds_l > 64 * 1024)                                                                               \
            ensure_dyn_lds((const void*)kern, lds_l); \
        hipLaunchKernelGGL(kern, grid, blockl, lds_l, s, x, B, N, g->sweep_lv_ptr, g->sweep_lv_data, G, g->eu, g->ev, \
                           E, halve, obj, has_stage);                                                         \
    } while (0)
#define DISPATCH_SWL_P(VEC, SWV)                       \
    switch (P) {                                       \
        case 12: LAUNCH_SWL(VEC, SWV, 12); break;      \
        case 16: LAUNCH_SWL(VEC, SWV, 16); break;      \
        case 20: LAUNCH_SWL(VEC, SWV, 20); break;      \
        default: LAUNCH_SWL(VEC, SWV, 24); break;      \
    }
#define DISPATCH_SWL(VEC)                                       \
    do {                                                        \
        if (sw == 16) { DISPATCH_SWL_P(VEC, 16) }               \
        else if (sw == 4) { DISPATCH_SWL_P(VEC, 4) }            \
        else if (sw == 2) { DISPATCH_SWL_P(VEC, 2) }            \
        else { DISPATCH_SWL_P(VEC, 8) }                         \
    } while (0)
            if (vec) DISPATCH_SWL(true); else DISPATCH_SWL(false);
#undef DISPATCH_SWL
#undef DISPATCH_SWL_P
#undef LAUNCH_SWL
            return check_launch("k_maxcut_greedy_sweep_levels");
        }
    }
    const bool unbatched = knob_on(KN_SWEEP_UNBATCHED);   // dev knob
    if (fast && g->sweep_rowptr && g->sweep_stream && !unbatched) {
        const int force_sw = (int)knob(KN_SWEEP_WAVES, 0);   // dev knob
        const int sw = force_sw == 4 || force_sw == 8 || force_sw == 16 ? force_sw
                                                                         : (ceil_div(B, kWave) <= (int64_t)num_cus() ? 16 : 8);
        const size_t lds_b = lds_fast + (size_t)sw * kWave * 8;
        const dim3 blockw(sw * kWave);
#define LAUNCH_SWB(VEC)                                                                                    \
    do {                                                                                                   \
        auto kern = sw == 16 ? k_maxcut_greedy_sweep_batched<VEC, 16>                                      \
                             : (sw == 8 ? k_maxcut_greedy_sweep_batched<VEC, 8> : k_maxcut_greedy_sweep_batched<VEC, 4>); \
        if (lds_b > 64 * 1024)                                                                             \
            ensure_dyn_lds((const void*)kern, lds_b); \
        hipLaunchKernelGGL(kern, grid, blockw, lds_b, s, x, B, N, g->sweep_rowptr, g->sweep_stream, g->nnz + N, obj); \
    } while (0)
        if (lds_b <= (size_t)kLdsBytes) {
            if (vec) LAUNCH_SWB(true); else LAUNCH_SWB(false);
            return check_launch("k_maxcut_greedy_sweep_batched");
        }
#undef LAUNCH_SWB
    }
    if (fast) {
#define LAUNCH_SWF(VEC)                                                                                    \
    do {                                                                                                   \
        auto kern = k_maxcut_greedy_sweep<VEC>;                                                            \
        if (lds_fast > 64 * 1024)                                                                          \
            ensure_dyn_lds((const void*)kern, lds_fast); \
        hipLaunchKernelGGL(kern, grid, block, lds_fast, s, x, B, N, g->rowptr, g->col, g->nnz, obj);       \
    } while (0)
        if (vec) LAUNCH_SWF(true); else LAUNCH_SWF(false);
#undef LAUNCH_SWF
        return check_launch("k_maxcut_greedy_sweep");
    }
    const size_t lds = (size_t)N * 8;
    if (lds > (size_t)kLdsBytes) {   // the 64-env bit tile does not fit: one env per wave on a byte row
        const int rw = rows_waves(N);
        RLS_REQUIRE(rw > 0, RLS_EUNSUPPORTED, "N=%lld: a row does not fit LDS (max %d)", (long long)N, kLdsBytes);
        const size_t lr = (size_t)rw * (((size_t)N + 15) & ~(size_t)15);
        const dim3 gr((unsigned)ceil_div(B, rw)), br(rw * kWave);
        // the symmetric CSR counts every undirected edge once per endpoint: the gain of a flip needs no halving
        if (g->wgt) {
            if (lr > 64 * 1024) ensure_dyn_lds((const void*)k_maxcut_greedy_sweep_rows<true>, lr);
            hipLaunchKernelGGL(k_maxcut_greedy_sweep_rows<true>, gr, br, lr, s, x, B, N, g->rowptr, g->col, g->wgt, 0, obj);
        } else {
            if (lr > 64 * 1024) ensure_dyn_lds((const void*)k_maxcut_greedy_sweep_rows<false>, lr);
            hipLaunchKernelGGL(k_maxcut_greedy_sweep_rows<false>, gr, br, lr, s, x, B, N, g->rowptr, g->col, g->wgt, 0, obj);
        }
        return check_launch("k_maxcut_greedy_sweep_rows");
    }
#define LAUNCH_SW(VEC, W)                                                                                  \
    do {                                                                                                   \
        auto kern = k_maxcut_greedy_sweep_generic<VEC, W>;                                                 \
        if (lds > 64 * 1024)                                                                               \
            ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, x, B, N, g->rowptr, g->col, g->wgt, obj);            \
    } while (0)
    if (g->wgt) { if (vec) LAUNCH_SW(true, true); else LAUNCH_SW(false, true); }
    else        { if (vec) LAUNCH_SW(true, false); else LAUNCH_SW(false, false); }
#undef LAUNCH_SW
    return check_launch("k_maxcut_greedy_sweep_generic");
}

int rls_maxcut_edge_cut_mask(const rls_graph* g, const uint8_t* x, int64_t B, uint8_t* cutmask, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    const int64_t E = g->num_stored_edges;
    if (B == 0 || E == 0) return RLS_OK;
    RLS_REQUIRE(x && cutmask, RLS_EINVAL, "x/cutmask is NULL");
    hipLaunchKernelGGL(k_edge_cut_mask, dim3(grid_for(B * E, 256)), dim3(256), 0, as_stream(stream), x, B,
                       g->num_nodes, g->eu, g->ev, E, cutmask);
    return check_launch("k_edge_cut_mask");
}

static inline size_t node_stats_lds(int64_t N) { return (size_t)(N + 2) * 8 + (size_t)kTileWaves * kWave * 144; }
// the lane = env tile kernels (weighted graphs, degrees >= 65536) walk every node and entry of the graph once per tile, 0.10 us
// per node + 0.008 us per entry whatever the batch (K3 on a +-1-weighted G22-sized graph: 530 us from 2048 to 16 384 envs), the
// element-parallel kernels 2.75e-6 us per (env, node + entry) (250 / 480 / 1890 us at 2048 / 4096 / 16 384): the tile form from
// the batch where it is the cheaper one (tools/sweeps/node_stats_forms.py)
static inline bool node_stats_use_tile(const rls_graph* g, int64_t B) {
    const bool off = knob_on(KN_NODE_STATS_NO_TILE);   // dev knob
    const int64_t N = g->num_nodes;
    return !off && node_stats_lds(N) <= (size_t)kLdsBytes &&
           2.75e-6 * (double)B * (double)(N + g->nnz) > 0.103 * (double)N + 0.008 * (double)g->nnz;
}

// which kernel family K2 / K3 / the local-search weights take for a batch of B envs: 1 bit-sliced (lane = node), 2 lane = env
// tile, 0 element-parallel -- the launchers' own tests (what = 0: K2 / weights, the adjacency as stored; 1: K3, symmetric)
int rls_maxcut_node_stats_form(const rls_graph* g, int64_t B, int32_t what) {
    if (!g || g->num_nodes <= 0 || B <= 0) return 0;
    if (what ? node_stats_use_bits(g, g->ell_sym_ptr, g->ell_sym, B) : node_stats_use_bits(g, g->ell_st_ptr, g->ell_st, B)) return 1;
    return node_stats_use_tile(g, B) ? 2 : 0;
}

int rls_maxcut_node_cutdeg(const rls_graph* g, const uint8_t* x, int64_t B, int64_t* cutdeg, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x && cutdeg, RLS_EINVAL, "x/cutdeg is NULL");
    if (node_stats_use_bits(g, g->ell_st_ptr, g->ell_st, B))
        return launch_node_stats_bits<0>(g, x, B, g->erowptr, g->ell_st_ptr, g->ell_st, 0, cutdeg, stream);
    const int64_t N = g->num_nodes;
    const size_t lds = node_stats_lds(N);
    if (node_stats_use_tile(g, B)) {
        const dim3 grid((unsigned)ceil_div(B, kWave)), block(kTileWaves * kWave);
        hipStream_t s = as_stream(stream);
#define LAUNCH_NS(VEC)                                                                                           \
    do {                                                                                                         \
        auto kern = k_node_stats_tile<int64_t, false, false, VEC>;                                               \
        if (lds > 64 * 1024)                                                                                     \
            ensure_dyn_lds((const void*)kern, lds);  \
        hipLaunchKernelGGL(kern, grid, block, lds, s, x, B, N, g->erowptr, g->ev, (const int32_t*)nullptr, cutdeg); \
    } while (0)
        if (tile_rows_aligned(x, N, 1)) LAUNCH_NS(true); else LAUNCH_NS(false);
#undef LAUNCH_NS
        return check_launch("k_node_stats_tile<cutdeg>");
    }
    hipLaunchKernelGGL(k_node_cutdeg, dim3(grid_for(B * N, 256)), dim3(256), 0, as_stream(stream), x, B, N, g->erowptr,
                       g->ev, cutdeg);
    return check_launch("k_node_cutdeg");
}

int rls_maxcut_delta_all(const rls_graph* g, const uint8_t* x, int64_t B, int32_t* delta, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x && delta, RLS_EINVAL, "x/delta is NULL");
    if (node_stats_use_bits(g, g->ell_sym_ptr, g->ell_sym, B))
        return launch_node_stats_bits<1>(g, x, B, g->rowptr, g->ell_sym_ptr, g->ell_sym, 0, delta, stream);
    const int64_t N = g->num_nodes;
    const size_t lds = node_stats_lds(N);
    if (node_stats_use_tile(g, B)) {
        const dim3 grid((unsigned)ceil_div(B, kWave)), block(kTileWaves * kWave);
        hipStream_t s = as_stream(stream);
#define LAUNCH_ND(W, VEC)                                                                                        \
    do {                                                                                                         \
        auto kern = k_node_stats_tile<int32_t, true, W, VEC>;                                                    \
        if (lds > 64 * 1024)                                                                                     \
            ensure_dyn_lds((const void*)kern, lds);  \
        hipLaunchKernelGGL(kern, grid, block, lds, s, x, B, N, g->rowptr, g->col, g->wgt, delta);               \
    } while (0)
        const bool vec = tile_rows_aligned(x, N, 1);
        if (g->wgt) { if (vec) LAUNCH_ND(true, true); else LAUNCH_ND(true, false); }
        else        { if (vec) LAUNCH_ND(false, true); else LAUNCH_ND(false, false); }
#undef LAUNCH_ND
        return check_launch("k_node_stats_tile<delta>");
    }
    const dim3 grid(grid_for(B * N, 256)), block(256);
    if (g->wgt)
        hipLaunchKernelGGL(k_delta_all<true>, grid, block, 0, as_stream(stream), x, B, N, g->rowptr, g->col, g->wgt,
                           delta);
    else
        hipLaunchKernelGGL(k_delta_all<false>, grid, block, 0, as_stream(stream), x, B, N, g->rowptr, g->col, g->wgt,
                           delta);
    return check_launch("k_delta_all");
}

}  // extern "C" (a template follows)

template <typename WT>
static int ls_weights_typed(const rls_graph* g, const uint8_t* x, int64_t B, int32_t mult, WT* ws, int64_t pitch, int32_t* minmax,
                            void* stream) {
    // (a lane = env tile kernel used to take the small batches: 850 us per call on a G22-sized graph at any batch size, against
    // 30 us for the bit-sliced one and 20 - 100 us for the element-parallel one: tools/sweeps/ls_weights_forms.py)
    if (node_stats_use_bits(g, g->ell_st_ptr, g->ell_st, B))
        return launch_node_stats_bits<2, WT>(g, x, B, g->erowptr, g->ell_st_ptr, g->ell_st, (int)mult, ws, stream, minmax, pitch);
    hipLaunchKernelGGL(k_ls_weights_elem<WT>, dim3(grid_for(B * g->num_nodes, 256)), dim3(256), 0, as_stream(stream), x, B, g->num_nodes,
                       g->erowptr, g->ev, (int)mult, ws, pitch, minmax);
    return check_launch("k_ls_weights_elem");
}

extern "C" {

int rls_maxcut_ls_weights(const rls_graph* g, const uint8_t* x, int64_t B, int32_t mult, void* ws, int32_t ws_bytes,
                          int64_t ws_pitch, int32_t* ws_minmax, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0 && mult >= 0, RLS_EINVAL, "bad arguments");
    if (ws_pitch == 0) ws_pitch = g->num_nodes;
    RLS_REQUIRE(ws_pitch >= g->num_nodes, RLS_EINVAL, "ws_pitch %lld < N", (long long)ws_pitch);
    RLS_REQUIRE(ws_bytes == 1 || ws_bytes == 2 || ws_bytes == 4, RLS_EINVAL, "ws_bytes must be 1, 2 or 4");
    // ws lies in [-(mult - 1) deg, deg] with deg <= the graph's largest degree
    const int64_t span = (int64_t)g->max_degree * (mult > 1 ? mult - 1 : 1);
    RLS_REQUIRE(ws_bytes == 4 || span <= (ws_bytes == 1 ? 127 : 32767), RLS_EINVAL,
                "weights up to %lld do not fit %d-byte entries", (long long)span, (int)ws_bytes);
    if (ws_minmax)   // also for B == 0: the caller reads max - min
        hipLaunchKernelGGL(k_fill_minmax, dim3((unsigned)ceil_div(g->num_nodes, 256)), dim3(256), 0, as_stream(stream), ws_minmax,
                           g->num_nodes);
    if (B == 0) return ws_minmax ? check_launch("k_fill_minmax") : RLS_OK;
    RLS_REQUIRE(x && ws, RLS_EINVAL, "NULL pointer");
    if (ws_bytes == 1) return ls_weights_typed<int8_t>(g, x, B, mult, (int8_t*)ws, ws_pitch, ws_minmax, stream);
    if (ws_bytes == 2) return ls_weights_typed<int16_t>(g, x, B, mult, (int16_t*)ws, ws_pitch, ws_minmax, stream);
    // 4-byte entries: only graphs with degrees beyond 32767 need them (and only the decomposed local search reads them):
    // the element-parallel kernel, no tile variants
    hipLaunchKernelGGL(k_ls_weights_elem<int32_t>, dim3(grid_for(B * g->num_nodes, 256)), dim3(256), 0, as_stream(stream), x, B,
                       g->num_nodes, g->erowptr, g->ev, (int)mult, (int32_t*)ws, ws_pitch, ws_minmax);
    return check_launch("k_ls_weights_elem");
}

int rls_select_better_rows(uint8_t* xs0, int64_t* vs0, const uint8_t* xs1, const int64_t* vs1, int64_t B,
                           int64_t N, int if_maximize, void* stream) {
    RLS_REQUIRE(B >= 0 && N >= 0, RLS_EINVAL, "negative size");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(xs0 && vs0 && xs1 && vs1, RLS_EINVAL, "NULL pointer");
    hipLaunchKernelGGL(k_select_better_rows, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, as_stream(stream), xs0,
                       vs0, xs1, vs1, B, N, if_maximize);
    return check_launch("k_select_better_rows");
}

int rls_pick_best_of_repeats(const uint8_t* xs, const int64_t* vs, int64_t R, int64_t S, int64_t N,
                             int if_maximize, uint8_t* good_xs, int64_t* good_vs, void* stream) {
    RLS_REQUIRE(R > 0 && S >= 0 && N >= 0, RLS_EINVAL, "bad sizes R=%lld S=%lld N=%lld", (long long)R, (long long)S,
                (long long)N);
    if (S == 0) return RLS_OK;
    RLS_REQUIRE(xs && vs && good_xs && good_vs, RLS_EINVAL, "NULL pointer");
    hipLaunchKernelGGL(k_pick_best_of_repeats, dim3((unsigned)ceil_div(S, 4)), dim3(256), 0, as_stream(stream), xs,
                       vs, R, S, N, if_maximize, good_xs, good_vs);
    return check_launch("k_pick_best_of_repeats");
}

static int rand_spins_launch(uint8_t* x, int64_t B, int64_t N, uint64_t seed, int64_t env_offset, const uint64_t* repeat_seeds, int64_t S,
                             void* stream) {
    const bool vec16 = (N % 16 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    if (N >= 128 && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ((N + 127) >> 7) <= kSpinMaxBlk) {
        // several rows per wave pass, every lane draws blocks (k_rand_spins_multi): RP = as many rows as fill one wave of calls
        // (short rows), else the 1..4 rows that leave the fewest idle lanes in the last wave of calls
        const int nb = (int)((N + 127) >> 7);
        int RP = nb <= kWave ? kWave / nb : 1;
        if (nb > kWave) {
            double best = 2.0;
            for (int r = 1; r <= 4 && r * nb <= kSpinMaxBlk; ++r) {
                const double waste = (double)(ceil_div((int64_t)r * nb, kWave) * kWave) / (double)(r * nb);
                if (waste < best - 1e-9) { best = waste; RP = r; }
            }
        }
        const int ch = (N & 15) == 0 ? 16 : ((N & 7) == 0 ? 8 : 4);      // (pieces are cut on the run of RP rows: RP * N is a multiple too)
        const dim3 grid(grid_for(ceil_div(B, (int64_t)RP) * kWave, 256)), block(256);
        if (ch == 16) hipLaunchKernelGGL(k_rand_spins_multi<16>, grid, block, 0, as_stream(stream), x, B, N, seed, env_offset, repeat_seeds, S, nb, RP);
        else if (ch == 8) hipLaunchKernelGGL(k_rand_spins_multi<8>, grid, block, 0, as_stream(stream), x, B, N, seed, env_offset, repeat_seeds, S, nb, RP);
        else hipLaunchKernelGGL(k_rand_spins_multi<4>, grid, block, 0, as_stream(stream), x, B, N, seed, env_offset, repeat_seeds, S, nb, RP);
        return check_launch("k_rand_spins_multi");
    }
    if (N >= 512 && (N & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        // rows of at least four 128-spin blocks, 4-byte aligned: one Philox call per block (a wave per row)
        const dim3 grid(grid_for(B * kWave, 256)), block(256);
        if ((N & 15) == 0) hipLaunchKernelGGL(k_rand_spins_rows<16>, grid, block, 0, as_stream(stream), x, B, N, seed, env_offset, repeat_seeds, S);
        else if ((N & 7) == 0) hipLaunchKernelGGL(k_rand_spins_rows<8>, grid, block, 0, as_stream(stream), x, B, N, seed, env_offset, repeat_seeds, S);
        else hipLaunchKernelGGL(k_rand_spins_rows<4>, grid, block, 0, as_stream(stream), x, B, N, seed, env_offset, repeat_seeds, S);
        return check_launch("k_rand_spins_rows");
    }
    hipLaunchKernelGGL(vec16 ? k_rand_spins<true> : k_rand_spins<false>, dim3(grid_for(B * ((N + 15) >> 4), 256)),
                       dim3(256), 0, as_stream(stream), x, B, N, seed, env_offset, repeat_seeds, S);
    return check_launch("k_rand_spins");
}

int rls_rand_spins(uint8_t* x, int64_t B, int64_t N, uint64_t seed, int64_t env_offset, void* stream) {
    RLS_REQUIRE(B >= 0 && N > 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(x, RLS_EINVAL, "x is NULL");
    return rand_spins_launch(x, B, N, seed, env_offset, nullptr, 1, stream);
}

int rls_rand_spins_repeats(uint8_t* x, int64_t R, int64_t S, int64_t N, const uint64_t* repeat_seeds, int64_t env_offset, void* stream) {
    RLS_REQUIRE(R >= 0 && S >= 0 && N > 0, RLS_EINVAL, "bad sizes");
    if (R == 0 || S == 0) return RLS_OK;
    RLS_REQUIRE(x && repeat_seeds, RLS_EINVAL, "x / repeat_seeds is NULL");
    return rand_spins_launch(x, R * S, N, 0, env_offset, repeat_seeds, S, stream);
}

int rls_rand_actions(int64_t* action, int64_t B, int64_t N, uint64_t seed, uint64_t step, int64_t env_offset,
                     void* stream) {
    RLS_REQUIRE(B >= 0 && N > 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(action, RLS_EINVAL, "action is NULL");
    hipLaunchKernelGGL(k_rand_actions, dim3(grid_for(B, 256)), dim3(256), 0, as_stream(stream), action, B, N, seed,
                       step, env_offset);
    return check_launch("k_rand_actions");
}

}  // extern "C"
