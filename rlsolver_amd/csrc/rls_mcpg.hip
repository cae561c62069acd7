// MCPG sampling kernels (K7, K8, K9) for gfx950.
//
// Layout at the boundary is the reference's: node-major x[N, C] (chains are the fast axis), f32
// holding 0/1 (or uint8 0/1).  A wavefront owns 64 consecutive chains and keeps them as a bit tile
// in LDS (words[n] bit c = x[n, c0+c]); node-major rows make the tile load perfectly coalesced
// (64 lanes read 64 consecutive chains of one node) and one ballot per node builds the word.
//
//   K9 metro rounds    : lane = chain; per round one random node, MH accept, flip = atomic XOR of
//                        the lane's bit in LDS.  Per-round accept counts let the host-side wrapper
//                        apply the reference's global early-stop rule without a host sync.
//   K7 local search    : sequential over nodes in the given order (as the reference's semantics
//                        require), wave-uniform CSR row, broadcast LDS reads, lane = chain.
//   K8 expected cut    : the K1 bit-sliced edge counter on the same tile; expected = E - 2*cut.
//
// f32 arithmetic of the reference is reproduced exactly: neighbour sums are multiples of 0.5
// (exact in any order), `s + u * 0.25f` and the division `(1 - p) / p` are single IEEE f32 ops
// (the library is built with -ffp-contract=off).
#include "rls_cutcount.h"
#include "rls_ring.h"
#include <type_traits>

namespace rls {

// wave w of W transposes every W-th group of 64 nodes; 32 row loads are in flight per wave (the loop
// is latency-bound: each load is one coalesced 64-chain row segment)
template <typename T>
__device__ __forceinline__ void tile_load_bits_nodemajor(const T* __restrict__ x, int64_t N, int64_t C, int64_t c0,
                                                         uint64_t* __restrict__ words, int lane, int w = 0, int W = 1) {
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    const int64_t cc = valid ? c : C - 1;
    constexpr int DEPTH = 32;   // row loads in flight per wave (256 B each): the loop is pure HBM latency
    for (int64_t n0 = (int64_t)w * kWave; n0 < N; n0 += (int64_t)W * kWave) {
        const int lim = (int)((N - n0) < kWave ? (N - n0) : kWave);
        uint64_t mine = 0;
        for (int k0 = 0; k0 < lim; k0 += DEPTH) {
            // clamped, unconditional loads (a guarded load per row makes hipcc wait for each one in turn); rows and
            // chains past the end are masked after the fact
            T v[DEPTH];
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                const int64_t nn = (k0 + q < lim) ? n0 + k0 + q : N - 1;
                v[q] = x[nn * C + cc];
            }
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
                const uint64_t wd = ballot64(valid && spin_is_set(v[q]));
                if (lane == k0 + q) mine = wd;
            }
        }
        if (lane < lim) words[n0 + lane] = mine;
    }
}

template <typename T>
__device__ __forceinline__ void tile_store_nodemajor(T* __restrict__ x, int64_t N, int64_t C, int64_t c0,
                                                     const uint64_t* __restrict__ words, int lane, int w = 0, int W = 1) {
    const int64_t c = c0 + lane;
    if (c >= C) return;
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(words);
    const int half = lane >> 5, sh = lane & 31;
#pragma unroll 8
    for (int64_t n = w; n < N; n += W) x[n * C + c] = (T)((w32[(n << 1) + half] >> sh) & 1u);
}

// ---- bit-packed chains ("spin_bytes = 0"): tile-major uint64 [ceil(C / 64), N]; word (t, n) holds node n of the chains
// 64 t .. 64 t + 63 (bit e = chain 64 t + e).  A 64-chain tile is N CONSECUTIVE words: the LDS tile is a straight
// copy (16-byte lanes), 1/32 of the f32 node-major surface's bytes.  tiles_in < tiles broadcasts: tile t reads
// tile t % tiles_in (the reference's  xs_bool = temp_max_info.repeat(1, repeat_times)  without materialising it).
struct Packed64 {};
template <typename T> struct ChainStore { using type = T; };
template <> struct ChainStore<Packed64> { using type = uint64_t; };

__device__ __forceinline__ void tile_load_packed(const uint64_t* __restrict__ x, int64_t N, int64_t C, int64_t tile,
                                                 int64_t tiles_in, uint64_t* __restrict__ words, int tid, int nthreads) {
    const uint64_t* src = x + (tile % tiles_in) * N;
    const int64_t left = C - tile * kWave;
    const uint64_t live = left >= kWave ? ~0ull : ((1ull << left) - 1ull);   // chains past C read as 0
    if ((N & 1) == 0 && (((uintptr_t)src) & 15) == 0) {
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        const u64x2* s2 = reinterpret_cast<const u64x2*>(src);
        u64x2* d2 = reinterpret_cast<u64x2*>(words);
        const u64x2 m{live, live};
#pragma unroll 4
        for (int64_t i = tid; i < N / 2; i += nthreads) d2[i] = s2[i] & m;
    } else {
        for (int64_t i = tid; i < N; i += nthreads) words[i] = src[i] & live;
    }
}

__device__ __forceinline__ void tile_store_packed(uint64_t* __restrict__ x, int64_t N, int64_t tile,
                                                  const uint64_t* __restrict__ words, int tid, int nthreads) {
    uint64_t* dst = x + tile * N;
    if ((N & 1) == 0 && (((uintptr_t)dst) & 15) == 0) {
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        const u64x2* s2 = reinterpret_cast<const u64x2*>(words);
        u64x2* d2 = reinterpret_cast<u64x2*>(dst);
#pragma unroll 4
        for (int64_t i = tid; i < N / 2; i += nthreads) d2[i] = s2[i];
    } else {
        for (int64_t i = tid; i < N; i += nthreads) dst[i] = words[i];
    }
}

// ------------------------------------------------------------------------------------- K9
constexpr int kMetroWaves = 16;   // the walk is one wave; the other 15 only move the tile (4 waves: 1.1 ms per 2.5 MB tile load)

// murmur3's 32-bit finaliser: the counter-based generator of the production paths of K7 and K9 (two hashes
// per draw pair instead of a 10-round Philox, which was 70 % of a K9 round on its single walking wave)
__device__ __forceinline__ uint32_t k7_fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// Global id of local chain c (rls_chain_ids): the key of every counter-based draw, so that a rank's shard of the chains
// draws exactly what those chains draw in an unsharded run.  period > 0: the local batch is repeats of `period` kept
// chains (chain c = repeat c / period of kept chain c % period) cut out of a global batch whose repeats are period + skip
// chains apart.  No division in the kernels: in that mode the launch is TWO-dimensional -- blockIdx.y = the repeat,
// blockIdx.x = the 64-chain tile inside it (period a multiple of 64) -- and a kernel's linear tile index is mcpg_tile().
struct ChainIds {
    int64_t offset, skip;
    __device__ __forceinline__ int64_t operator()(int64_t c) const { return offset + c + (int64_t)blockIdx.y * skip; }
};
__device__ __forceinline__ int64_t mcpg_tile() { return (int64_t)blockIdx.y * gridDim.x + blockIdx.x; }

template <typename T, bool PROBS_LDS>
__global__ __launch_bounds__(kMetroWaves * kWave) void k_mcpg_metro(T* samples, const T* samples_in, int64_t N, int64_t C,
                                                      const float* __restrict__ probs, int64_t T_rounds,
                                                      const int64_t* __restrict__ index,
                                                      const float* __restrict__ u, uint64_t seed,
                                                      const int64_t* __restrict__ t_limit_dev, int write_back,
                                                      unsigned long long* __restrict__ accepts_all, int64_t accept_rows, int64_t t_offset,
                                                      ChainIds ids) {
    // accept counts go to row (workgroup % accept_rows) of [accept_rows][T_rounds]: thousands of workgroups adding into ONE row
    // serialise at the L2 atomic units (measured: 3/4 of the packed walk's time)
    unsigned long long* accepts = accepts_all ? accepts_all + (int64_t)((uint64_t)mcpg_tile() % (uint64_t)accept_rows) * T_rounds : nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    // probs staged in LDS when it fits: a per-round random gather from global memory would put an
    // L2 round trip (~2 us under load) on every round of every chain
    float* probs_l = reinterpret_cast<float*>(words + N);
    uint32_t* acc_cnt = reinterpret_cast<uint32_t*>(probs_l + (PROBS_LDS ? N : 0));   // per-round accept counts
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t c0 = mcpg_tile() * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    // nothing to do for this chunk (stop rule already met) -- unless the chains still have to be moved to `samples`
    if (t_limit_dev && *t_limit_dev <= 0 && samples_in == samples) return;
    if constexpr (PROBS_LDS)
        for (int64_t n = threadIdx.x; n < N; n += kMetroWaves * kWave) probs_l[n] = probs[n];
    tile_load_bits_nodemajor<T>(samples_in, N, C, c0, words, lane, w, kMetroWaves);
    __syncthreads();
    int64_t t_end = T_rounds;
    if (t_limit_dev) {
        const int64_t lim = *t_limit_dev;
        t_end = lim < T_rounds ? (lim > 0 ? lim : 0) : T_rounds;
    }
    const int64_t gc = ids(c);                                  // the chain's global id
    const uint32_t chain_key = k7_fmix32((uint32_t)seed ^ k7_fmix32((uint32_t)(seed >> 32) ^ k7_fmix32((uint32_t)gc) ^
                                                                    ((uint32_t)((uint64_t)gc >> 32) * 0x9E3779B1u)));
    const uint64_t mybit = 1ull << lane;
    const BitXpose xc = bit_xpose_consts(lane);
    uint32_t alo = 0, ahi = 0;                        // this chain's accept bits of the current 64 rounds
    for (int64_t t = 0; w == 0 && t < t_end; ++t) {   // the chain walk itself is one wave (64 chains = 64 lanes)
        int64_t i = 0;
        float uu = 2.0f;
        if (valid) {
            if (index) {
                i = index[(t_offset + t) * C + c];
                uu = u[(t_offset + t) * C + c];
            } else {
                const uint32_t k = chain_key ^ ((uint32_t)(t_offset + t) * 0x9E3779B1u);
                const uint32_t r0 = k7_fmix32(k ^ 0x4D455452u), r1 = k7_fmix32(k + 0x7FEB352Du);
                i = (int64_t)(((uint64_t)r0 * (uint64_t)N) >> 32);
                uu = u32_to_unit_float(r1);
            }
        }
        const bool val = (words[i] >> lane) & 1ull;
        const float base = PROBS_LDS ? probs_l[i] : probs[i];
        const float chosen = val ? base : 1.0f - base;            // torch.where(chosen_value, p, 1 - p)
        const float accept_rate = (1.0f - chosen) / chosen;       // MCPG.py:107
        const bool acc = valid && (uu < accept_rate);
        // One wave, and the LDS executes a wave's operations in issue order: every lane's read of this round
        // precedes the flips, and the flips precede the next round's reads -- no waits needed, so the next
        // round's hash and address math overlap this round's LDS round trips.
        if (acc) atomicXor(reinterpret_cast<unsigned long long*>(&words[i]), (unsigned long long)mybit);
        if (accepts) {
            // one bit per (chain, round) in registers and ONE 64 x 64 bit transpose + popcount per 64 rounds: a ballot +
            // popcount per round is a VALU -> SALU -> VALU round trip on the walk's critical path
            const uint32_t a = acc ? 1u : 0u;
            if ((t & 32) == 0) alo |= a << (t & 31);
            else ahi |= a << (t & 31);
            if ((t & 63) == 63 || t + 1 == t_end) {
                bit_transpose64(alo, ahi, xc);                   // lane r now holds round (t & ~63) + r by chain
                const int64_t tr = (t & ~(int64_t)63) + lane;
                if (tr <= t) acc_cnt[tr] = (uint32_t)(__builtin_popcount(alo) + __builtin_popcount(ahi));
                alo = ahi = 0;
            }
        }
    }
    __syncthreads();
    if (accepts) {   // one coalesced burst of atomics per workgroup instead of one contended atomic per round
        for (int64_t t = threadIdx.x; t < t_end; t += kMetroWaves * kWave)
            if (acc_cnt[t]) atomicAdd(&accepts[t], (unsigned long long)acc_cnt[t]);
    }
    if (write_back) tile_store_nodemajor<T>(samples, N, C, c0, words, lane, w, kMetroWaves);
}

// K9 on bit-packed chains.  The tile moves as a straight 16-byte-lane copy (no transposition) and the workgroup is split
// into ONE walking wave and kMetroPW - 1 draw producers, because a lone wave is instruction-issue bound (~5 cycles per
// VALU instruction): of the ~100 instructions of a round, ~85 do not depend on the chain state -- the counter hash, the
// node id, the probs gather and BOTH possible Metropolis tests "u < (1 - q) / q" (q = p if the bit is set, 1 - p if not;
// the same two IEEE operations as MCPG.py:105-107).  Producers fill a window of 64 rounds in LDS, one dword per (round,
// chain): node | accept-if-0 << 30 | accept-if-1 << 31, while the walker consumes the previous window: read the word
// of the drawn node, pick the pre-computed verdict by the chain's bit, flip with an LDS XOR.  The walker issues the
// read of round r + 1 BEFORE the flip of round r (the LDS executes a wave's operations in issue order, so flips of
// rounds <= r - 1 are visible) and patches the one hazard in registers: its own flip of round r on the same node.
// Per-round accept counts ride in registers (lane r keeps round r of the window) and leave as one coalesced atomic
// per window.  Same draws as the f32 kernel for the same seed.
constexpr int kMetroPW = 8;       // waves per workgroup
constexpr int kMetroWin = 64;     // rounds per window

#ifdef RLS_K7_PROF   // dev build (RLS_EXTRA_CFLAGS=-DRLS_K7_PROF): cycles per wave at a level barrier / waiting for a header / in a group
static __device__ unsigned long long g_k7_prof[2048 * 16 * 6];   // (k_mcpg_metro_packed: [workgroup][0] = total, walking, at the window barrier, windows)
#define K7_NOW() __builtin_readcyclecounter()
#endif

// QG: the draw windows live in GLOBAL scratch (32 KB per workgroup, L2-resident) instead of LDS -- at N = 10^4 the tile is 80 KB and the
// 32 KB of windows beside it left ONE workgroup per CU, i.e. one walking wave per CU; without them two fit (round 5: 1.00 -> 0.5x ms).
template <bool GIVEN, bool QG = false>   // GIVEN: the reference's recorded draws (tests); else the counter hash
__global__ __launch_bounds__(kMetroPW * kWave) void k_mcpg_metro_packed(uint64_t* __restrict__ samples,
                                                             const uint64_t* __restrict__ samples_in, int64_t tiles_in,
                                                             int64_t N, int64_t C, const float* __restrict__ probs,
                                                             int64_t T_rounds, const int64_t* __restrict__ index,
                                                             const float* __restrict__ u, uint64_t seed,
                                                             const int64_t* __restrict__ t_limit_dev, int write_back,
                                                             unsigned long long* __restrict__ accepts_all, int64_t accept_rows, int64_t t_offset,
                                                             ChainIds ids, uint32_t* __restrict__ gqueue) {
    // accept counts go to row (workgroup % accept_rows) of [accept_rows][T_rounds]: thousands of workgroups adding into ONE row
    // serialise at the L2 atomic units (measured: 3/4 of the packed walk's time)
    unsigned long long* accepts = accepts_all ? accepts_all + (int64_t)((uint64_t)mcpg_tile() % (uint64_t)accept_rows) * T_rounds : nullptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t tile = mcpg_tile();
    uint32_t* queue = QG ? gqueue + tile * (2 * kMetroWin * kWave)
                         : reinterpret_cast<uint32_t*>(smem + (size_t)((N + 1) & ~(int64_t)1) * 8);   // [2][kMetroWin][64]
    const int64_t c = tile * kWave + lane;
    const bool valid = c < C;
    const bool in_place = (samples_in == samples) && tiles_in == (int64_t)gridDim.x * gridDim.y;
    int64_t t_end = T_rounds;
    if (t_limit_dev) {
        const int64_t lim = *t_limit_dev;
        t_end = lim < T_rounds ? (lim > 0 ? lim : 0) : T_rounds;
    }
    if (t_end == 0 && in_place) return;           // stop rule already met and nothing to move
    tile_load_packed(samples_in, N, C, tile, tiles_in, words, threadIdx.x, kMetroPW * kWave);
    const int64_t gc = ids(c);                                  // the chain's global id
    const uint32_t chain_key = k7_fmix32((uint32_t)seed ^ k7_fmix32((uint32_t)(seed >> 32) ^ k7_fmix32((uint32_t)gc) ^
                                                                    ((uint32_t)((uint64_t)gc >> 32) * 0x9E3779B1u)));
    const int64_t nwin = (t_end + kMetroWin - 1) / kMetroWin;
    auto produce = [&](int64_t win) {             // waves 1 .. PW-1 share the window's rounds
        // A producer's rounds are a chain draw -> probs[] gather (an L2 round trip) -> verdicts; the NEXT round's draw and gather are
        // issued before this round's verdicts, so a round trip hides behind the arithmetic of its neighbours.  (All ten gathers
        // first, verdicts after, was measured in round 5 and is slower: the bursts take issue slots from the walker, 139 -> 171
        // cycles per round; so is a per-node table of the two thresholds in place of the divisions: the 8-byte gather costs more
        // than the divisions it saves.)
        uint32_t* q = queue + (win & 1) * (kMetroWin * kWave);
        auto draw = [&](int r, int64_t& i, float& uu, bool& live) {
            const int64_t t = win * kMetroWin + r;
            live = valid && r < kMetroWin && t < t_end;
            if constexpr (GIVEN) {
                const int64_t at = live ? (t_offset + t) * C + c : 0;
                i = index[at];
                uu = u[at];
            } else {
                const uint32_t k = chain_key ^ ((uint32_t)(t_offset + t) * 0x9E3779B1u);
                const uint32_t r0 = k7_fmix32(k ^ 0x4D455452u), r1 = k7_fmix32(k + 0x7FEB352Du);
                i = (int64_t)(((uint64_t)r0 * (uint64_t)N) >> 32);
                uu = u32_to_unit_float(r1);
            }
            i = live ? i : 0;
        };
        int64_t i, i_n;
        float uu, uu_n;
        bool live, live_n;
        draw(w - 1, i, uu, live);
        float p = probs[i];
        for (int r = w - 1; r < kMetroWin; r += kMetroPW - 1) {
            draw(r + kMetroPW - 1, i_n, uu_n, live_n);
            const float p_n = probs[i_n];
            const float q1 = p, q0 = 1.0f - p;                    // torch.where(chosen_value, p, 1 - p)
            const bool a1 = uu < (1.0f - q1) / q1;                // MCPG.py:107 for a set bit
            const bool a0 = uu < (1.0f - q0) / q0;                //             for a clear bit
            // byte offset of the node's word | verdict for a set bit << 30 | verdict for a clear bit << 31: the walker
            // shifts left by the current bit and reads the sign
            q[r * kWave + lane] = live ? (((uint32_t)i << 3) | ((uint32_t)a1 << 30) | ((uint32_t)a0 << 31)) : 0u;
            i = i_n; uu = uu_n; live = live_n; p = p_n;
        }
    };
    if (w > 0 && nwin > 0) produce(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const BitXpose xc = bit_xpose_consts(lane);
#ifdef RLS_K7_PROF
    unsigned long long pf_walk = 0, pf_sync = 0;
    const unsigned long long pf_t0 = K7_NOW();
#endif
    for (int64_t win = 0; win < nwin; ++win) {
#ifdef RLS_K7_PROF
        const unsigned long long pf_a = K7_NOW();
#endif
        if (w == 0) {
            const uint32_t* q = queue + (win & 1) * (kMetroWin * kWave);
            const int rounds = (int)((t_end - win * kMetroWin) < kMetroWin ? (t_end - win * kMetroWin) : kMetroWin);
            uint32_t mycnt = 0;
            __builtin_amdgcn_s_setprio(3);                        // the walker is the critical path of the workgroup
            // entries come 8 rounds at a time (they do not depend on the chain state); the word of round r + 1 is
            // requested before the flip of round r is issued.  A lane works on the 32-bit half of the word its chain
            // lives in; the flip is an UNCONDITIONAL ds_xor with a zero mask for a rejected proposal (no exec
            // juggling, and the wait for the early read never has to cover the atomic behind it); rounds past the stop
            // rule carry entry 0 = no verdict.  ~14 VALU per round -- the walk is one wave's issue rate.
            const uint32_t half4 = (uint32_t)(lane >> 5) * 4u, sh = (uint32_t)lane & 31u, mybit32 = 1u << sh;
            auto addr_of = [&](uint32_t e) { return (e & 0x3fffffffu) | half4; };
            auto half_at = [&](uint32_t a) { return *reinterpret_cast<const uint32_t*>(smem + a); };
            uint32_t eb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) eb[k] = q[k * kWave + lane];
            // Two rounds of look-ahead: the word of round r + 2 is requested in round r, BEFORE that round's flip is
            // issued, so it misses the flips of rounds r and r + 1; those two are patched in registers (m1 collects them
            // for the word that is next in line) -- other lanes' flips touch other bits of the word.
            uint32_t a0 = addr_of(eb[0]), a1 = addr_of(eb[1]);
            uint32_t w0 = half_at(a0), w1 = half_at(a1), m1 = 0;
            uint32_t alo = 0, ahi = 0;
            for (int r0 = 0; r0 < rounds; r0 += 8) {
                uint32_t en8[8], blk = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) en8[k] = (r0 + 8 + k < kMetroWin) ? q[(r0 + 8 + k) * kWave + lane] : 0u;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // (opaque from here on: otherwise the compiler folds the pending mask into the word the moment it is
                    // read, i.e. waits for the look-ahead read right after issuing it)
                    asm volatile("" : "+v"(m1));
                    const uint32_t bit = (w0 >> sh) & 1u;
                    const bool acc = (int32_t)(eb[k] << bit) < 0;
                    const uint32_t flip = acc ? mybit32 : 0u;
                    const uint32_t a2 = addr_of((k + 2 < 8) ? eb[(k + 2) & 7] : en8[(k + 2) & 7]);
                    const uint32_t w2 = half_at(a2);
                    __hip_atomic_fetch_xor(static_cast<uint32_t*>(__builtin_assume_aligned(smem + a0, 4)), flip, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t m2 = (a2 == a0) ? flip : 0u;
                    m1 ^= (a1 == a0) ? flip : 0u;
                    blk |= (flip >> sh) << k;                     // accept bit of round r0 + k (pure VALU: a ballot + popcount
                                                                   // per round is a VALU -> SALU -> VALU round trip, 240 of the
                                                                   // walker's 330 cycles per round when it was done that way)
                    w0 = w1 ^ m1;
                    a0 = a1;
                    w1 = w2;
                    m1 = m2;
                    a1 = a2;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) eb[k] = en8[k];
                if (r0 < 32) alo |= blk << r0;
                else ahi |= blk << (r0 - 32);
            }
            // lane l holds its chain's accept bits by round; transposed, lane r holds round r's accepts by chain
            if (accepts) {
                bit_transpose64(alo, ahi, xc);
                mycnt = (uint32_t)(__builtin_popcount(alo) + __builtin_popcount(ahi));
            }
            __builtin_amdgcn_s_setprio(0);
            if (accepts && lane < rounds && mycnt) atomicAdd(&accepts[win * kMetroWin + lane], (unsigned long long)mycnt);
        } else if (win + 1 < nwin) {
            produce(win + 1);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifdef RLS_K7_PROF
        const unsigned long long pf_b = K7_NOW();
        pf_walk += pf_b - pf_a;
#endif
        __syncthreads();
#ifdef RLS_K7_PROF
        pf_sync += K7_NOW() - pf_b;
#endif
    }
#ifdef RLS_K7_PROF
    if (lane == 0 && blockIdx.x < 2048 && blockIdx.y == 0) {
        unsigned long long* q = g_k7_prof + ((size_t)blockIdx.x * 16 + w) * 6;
        q[0] = K7_NOW() - pf_t0; q[1] = pf_sync; q[2] = 0; q[3] = pf_walk; q[4] = (unsigned long long)nwin; q[5] = 0;
    }
#endif
    if (write_back) tile_store_packed(samples, N, tile, words, threadIdx.x, kMetroPW * kWave);
}

// ------------------------------------------------------------------------------------- K7 + K8
template <typename TI, int P>
__global__ __launch_bounds__(kWave) void k_mcpg_local_search(const TI* __restrict__ xs_in, float* __restrict__ xs_out,
                                                             int64_t N, int64_t C,
                                                             const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const int32_t* __restrict__ order, int64_t num_ls,
                                                             const float* __restrict__ uniforms, uint64_t seed,
                                                             const int32_t* __restrict__ eu,
                                                             const int32_t* __restrict__ ev, int64_t E,
                                                             float* __restrict__ expected, ChainIds ids) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    uint32_t* updated = reinterpret_cast<uint32_t*>(words + N);  // 1 bit per node: visited in pass 0
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(smem);
    const int lane = threadIdx.x;
    const int64_t c0 = mcpg_tile() * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    tile_load_bits_nodemajor<TI>(xs_in, N, C, c0, words, lane);
    for (int64_t n = lane; n < (N + 31) / 32; n += kWave) updated[n] = 0;
    __syncthreads();
    const int half = lane >> 5, sh = lane & 31;
    const Philox ph(seed);
    const int64_t gc = ids(c);                                  // the chain's global id
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        for (int64_t pos = 0; pos < N; ++pos) {
            const int node = order[pos];
            const int r0 = rowptr[node], r1 = rowptr[node + 1];
            // neighbour sum in units of 0.5: updated nbr contributes 2*b, fresh nbr (pass 0) 4*b - 1
            int s2 = 0;
            for (int base = r0; base < r1; base += kWave) {
                const int n_here = (r1 - base) < kWave ? (r1 - base) : kWave;
                const int my_nb = (lane < n_here) ? col[base + lane] : 0;
                for (int j = 0; j < n_here; ++j) {
                    const int nb = __builtin_amdgcn_readlane(my_nb, j);
                    const int b = (int)((w32[((int64_t)nb << 1) + half] >> sh) & 1u);
                    const bool upd = (cnt > 0) || ((updated[nb >> 5] >> (nb & 31)) & 1u);
                    s2 += upd ? 2 * b : 4 * b - 1;
                }
            }
            float uu;
            if (uniforms) {
                uu = valid ? uniforms[(cnt * N + pos) * C + c] : 0.0f;
            } else {
                uint32_t r[4];
                ph((uint32_t)gc, (uint32_t)((uint64_t)gc >> 32), (uint32_t)(cnt * N + pos), 0x4C4F4353u, r);
                uu = u32_to_unit_float(r[0]);
            }
            const float rv = (float)s2 * 0.5f + uu * 0.25f;                   // MCPG.py:139-141
            const float thr = ((float)(r1 - r0) + 0.25f) / 2.0f;             // (weighted_degree + k) / 2
            const uint64_t nw = ballot64(rv < thr);
            if (lane == 0) {
                words[node] = nw;
                if (cnt == 0) updated[node >> 5] |= 1u << (node & 31);
            }
            __syncthreads();
        }
    }
    // K8: expected[c] = sum_e (2x_u - 1)(2x_v - 1) = E - 2 * cut
    const int64_t cut = tile_cut_count<P>(words, eu, ev, E, lane);
    if (valid) {
        expected[c] = (float)(E - 2 * cut);
        for (int64_t n = 0; n < N; ++n) xs_out[n * C + c] = (float)((w32[(n << 1) + half] >> sh) & 1u);
    }
}

// K7 fast path.  The host flattens the visiting order into ONE int32 "visit stream": visiting positions
// level-scheduled (methods/MCPG.py: build_visit_stream; same argument as rls_sweep.h) and cut into batches of
// pairwise NON-adjacent nodes (<= 32 nodes, <= 400 entries) whose earlier-visited neighbours all sit in earlier
// batches, so a batch may be decided in any order with results identical to the sequential pass:
//     per batch:  m, next_batch_offset, 0, off_0 .. off_{m-1}
//     per node (at stream offset off_k):  node, deg, nfresh, visiting position, then deg entries  nb | (fresh << 31)
// (fresh = nb is visited later than node in pass 0, i.e. still holds -0.5|1.5 there; nfresh = their
// count).  The stream is consumed in order through an LDS ring (rls_ring.h).  The waves of the
// workgroup take the nodes of a batch round-robin, one barrier per batch; per node: ONE lane-parallel
// ring read (header in lanes 0-3, first 60 entries behind it), v_readlane + broadcast word read +
// v_bfe + v_mad per neighbour, one ballot.  No global-memory latency per node (the generic kernel pays
// 2-3 dependent L2 round trips per node).  Production noise: one counter-based hash per (chain, pass,
// visiting position); test mode reads the reference's torch.rand draws.
constexpr int kK7Waves = 16;   // one workgroup per CU (80 KB bit tile at N = 10^4): 4 waves 27.5, 8 waves 20.1, 16 waves 18.1 ms

// WEIGHTED (upstream MCPG's weighted MaxCut sampler, rlsolver/methods/MCPG/sampling.py:89-127): records carry
//     node, deg, Wfresh, visiting position, Wdeg, then deg PAIRS (nb | fresh << 31, weight)
// (Wfresh = sum of the weights of not-yet-visited neighbours, Wdeg = weighted degree = sum of all weights, integers);
// the test becomes  sum_j w_j v_j + u / 4 < Wdeg / 2 + 0.125  and the expected value sum_e w_e (2x_u - 1)(2x_v - 1).
// `gauge_node` >= 0 applies the sampler's gauge fix first: every chain is XORed with its value at that node (:101-104).
template <typename TI, int P, bool WEIGHTED>
__global__ __launch_bounds__(kK7Waves * kWave) void k_mcpg_local_search_stream(
    const TI* __restrict__ xs_in, float* __restrict__ xs_out, int64_t N, int64_t C,
    const int32_t* __restrict__ vstream, int64_t vlen, int64_t num_ls, const float* __restrict__ uniforms,
    uint64_t seed, const int32_t* __restrict__ eu, const int32_t* __restrict__ ev, const int32_t* __restrict__ ew, int64_t E,
    int gauge_node, float* __restrict__ expected, ChainIds ids) {
    constexpr int HDR = WEIGHTED ? 5 : 4;                 // header words of a node record
    constexpr int FIRST = WEIGHTED ? 28 : 56;             // neighbours whose entries sit in the record's first 64 words
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int32_t* ring = reinterpret_cast<int32_t*>(smem + (size_t)(N + 2) * 8);
    int64_t* scratch = reinterpret_cast<int64_t*>(ring + kRing);
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(smem);
    const unsigned char* wbytes = smem;
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t c0 = mcpg_tile() * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    if (threadIdx.x == 0) words[N] = 0;   // sentinel word for lanes past a row's end
    tile_load_bits_nodemajor<TI>(xs_in, N, C, c0, words, lane, w, kK7Waves);
    if (gauge_node >= 0) {                // graph_probs = (graph_probs + graph_probs[hub]) % 2
        __syncthreads();
        const uint64_t hub = words[gauge_node];
        __syncthreads();
        for (int64_t n = threadIdx.x; n < N; n += kK7Waves * kWave) words[n] ^= hub;
    }
    const int sh = lane & 31;
    const uint32_t half4 = (uint32_t)(lane >> 5) * 4u;
    const uint32_t sentinel = (uint32_t)N;
    const int64_t gc = ids(c);                                  // the chain's global id
    const uint32_t chain_key = k7_fmix32((uint32_t)seed ^ k7_fmix32((uint32_t)(seed >> 32) ^ k7_fmix32((uint32_t)gc) ^
                                                                    ((uint32_t)((uint64_t)gc >> 32) * 0x9E3779B1u)));
    auto bit_of = [&](uint32_t e) -> uint32_t {
        return (*reinterpret_cast<const uint32_t*>(wbytes + ((e & 0x7fffffffu) * 8u + half4)) >> sh) & 1u;
    };
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        int64_t F;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        F = 0;
        if (w == 0) ring_prime(vstream, vlen, F, ring, lane);   // wave 0 alone feeds the ring (rls_sweep.h)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int64_t cur = 0;
        while (cur < vlen) {
            const uint32_t hb = (uint32_t)ring[(cur + lane) & (kRing - 1)];
            const int m = __builtin_amdgcn_readlane((int)hb, 0);
            const int64_t nxt = (uint32_t)__builtin_amdgcn_readlane((int)hb, 1);
            for (int k = w; k < m; k += kK7Waves) {
                const int64_t off = (uint32_t)__builtin_amdgcn_readlane((int)hb, 3 + k);
                const uint32_t blk = (uint32_t)ring[(off + lane) & (kRing - 1)];
                const int node = __builtin_amdgcn_readlane((int)blk, 0);
                const int deg = __builtin_amdgcn_readlane((int)blk, 1);
                const int nfresh = (cnt == 0) ? __builtin_amdgcn_readlane((int)blk, 2) : 0;   // WEIGHTED: sum of fresh weights
                const int64_t pos = (uint32_t)__builtin_amdgcn_readlane((int)blk, 3);   // visiting position: RNG key / draw index
                const int wdeg = WEIGHTED ? __builtin_amdgcn_readlane((int)blk, 4) : deg;
                const int64_t row = off + HDR;
                float uu;
                if (uniforms) uu = valid ? uniforms[(cnt * N + pos) * C + c] : 0.0f;
                else uu = u32_to_unit_float(k7_fmix32(chain_key ^ ((uint32_t)pos * 0x9E3779B1u) ^
                                                      ((uint32_t)cnt * 0x7FEB352Du + 0x165667B1u)));
                // s2 = sum over neighbours of weight * value in units of 0.5: 2 * bit, or in pass 0 for a neighbour not
                // visited yet 4 * bit - 1 (the -1s are nfresh).  The first entries of the row sit behind the header in blk
                // (lanes past the row end read the sentinel, whose word is 0); longer rows continue from the ring.
                int acc = 0;
                if constexpr (WEIGHTED) {
                    const int first = deg < FIRST ? deg : FIRST;
                    for (int j = 0; j < first; ++j) {
                        const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)blk, HDR + 2 * j);
                        const int wj = __builtin_amdgcn_readlane((int)blk, HDR + 2 * j + 1);
                        acc += wj * (int)(bit_of(e) << (1u + ((cnt == 0) ? (e >> 31) : 0u)));
                    }
                    for (int j = FIRST; j < deg; ++j) {
                        const uint32_t e = (uint32_t)ring[(row + 2 * j) & (kRing - 1)];
                        const int wj = ring[(row + 2 * j + 1) & (kRing - 1)];
                        acc += wj * (int)(bit_of(e) << (1u + ((cnt == 0) ? (e >> 31) : 0u)));
                    }
                } else {
                    const uint32_t mine = (lane >= 4 && lane - 4 < deg) ? blk : sentinel;
                    const int first = deg < FIRST ? deg : FIRST;
                    if (cnt == 0) {
                        for (int j = 0; j < first; j += 8) {
                            uint32_t wv[8], ee[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                ee[q] = (uint32_t)__builtin_amdgcn_readlane((int)mine, j + q + 4);
                                wv[q] = *reinterpret_cast<const uint32_t*>(wbytes + ((ee[q] & 0x7fffffffu) * 8u + half4));
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc += (int)(((wv[q] >> sh) & 1u) << (1u + (ee[q] >> 31)));
                        }
                        for (int j = FIRST; j < deg; ++j) {
                            const uint32_t e = (uint32_t)ring[(row + j) & (kRing - 1)];
                            acc += (int)(bit_of(e) << (1u + (e >> 31)));
                        }
                    } else {
                        for (int j = 0; j < first; j += 8) {
                            uint32_t wv[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const uint32_t e = (uint32_t)__builtin_amdgcn_readlane((int)mine, j + q + 4);
                                wv[q] = *reinterpret_cast<const uint32_t*>(wbytes + ((e & 0x7fffffffu) * 8u + half4));
                            }
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc += (int)((wv[q] >> sh) & 1u);
                        }
                        for (int j = FIRST; j < deg; ++j) acc += (int)bit_of((uint32_t)ring[(row + j) & (kRing - 1)]);
                        acc <<= 1;
                    }
                }
                const int s2 = acc - nfresh;                                      // units of 0.5
                const float rv = (float)s2 * 0.5f + uu * 0.25f;                   // MCPG.py:139-141 / sampling.py:114-116
                const float thr = ((float)wdeg + 0.25f) / 2.0f;                   // (weighted_degree + k) / 2
                const uint64_t nw = ballot64(rv < thr);
                if (lane == 0) words[node] = nw;
            }
            if (w == 0) ring_advance(vstream, vlen, F, nxt, ring, lane);
            __syncthreads();   // the batch's updates are visible before any wave reads the next batch's neighbours
            cur = nxt;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    // K8: expected[c] = sum_e w_e (2x_u - 1)(2x_v - 1) = W - 2 * (weight of the cut)
    int64_t total;
    if constexpr (WEIGHTED) {
        int64_t part = 0;                                                         // lane = chain, the waves split the edges
        for (int64_t e = w; e < E; e += kK7Waves) {
            const uint32_t d = bit_of((uint32_t)eu[e]) ^ bit_of((uint32_t)ev[e]);
            part += d ? -(int64_t)ew[e] : (int64_t)ew[e];
        }
        total = block_sum_partials<kK7Waves>(part, scratch, lane, w);             // = sum_e w_e (+1 | -1)
    } else {
        const int64_t cut = block_sum_partials<kK7Waves>(tile_cut_count<P>(words, eu, ev, E, lane, w, kK7Waves), scratch, lane, w);
        total = E - 2 * cut;
    }
    if (valid) {
        if (w == 0) expected[c] = (float)total;
        const int half = lane >> 5;
        for (int64_t n = w; n < N; n += kK7Waves) xs_out[n * C + c] = (float)((w32[(n << 1) + half] >> sh) & 1u);
    }
}

// K7 level-parallel (lane = node): include/rlsolver_hip.h, rls_mcpg_visit_levels / rls_mcpg_local_search_levels.
// A wave decides the 64 nodes of a group at once on 64-chain words: vertical counters of the neighbours' ones
// (pass 0: visited and not-yet-visited neighbours counted apart, C = cV + 2 cF), a bit-sliced compare with the
// per-node constant K and  new word = [C < K] | ([C == K] & tie & coin).  Nodes of degree > 64 get a group of
// their own and the lanes share the NEIGHBOURS instead (per-lane counters, then transpose + popcount per plane).
constexpr int kLvWaves = 16;      // f32 / uint8 node-major surface: 16 waves mostly to move the tile
constexpr int kLvWavesPacked = 8; // bit-packed chains: the tile is a straight copy; 8 waves, two workgroups per CU at N = 10^4

// 8 words into a (ones, twos, fours, c[0..NC-1] = planes 3..) vertical counter.  NC follows the group's number of
// rounds (a count <= 7 never carries out of `fours`, <= 15 needs one more plane, ...): the ripple is 4 VALU per plane.
template <int NC>
__device__ __forceinline__ void lv_add8(const uint64_t (&d)[8], uint64_t& ones, uint64_t& twos, uint64_t& fours,
                                        uint64_t (&c)[5]) {
    uint64_t twosA, twosB, foursA, foursB, carry;
    csa(twosA, ones, ones, d[0], d[1]);
    csa(twosB, ones, ones, d[2], d[3]);
    csa(foursA, twos, twos, twosA, twosB);
    csa(twosA, ones, ones, d[4], d[5]);
    csa(twosB, ones, ones, d[6], d[7]);
    csa(foursB, twos, twos, twosA, twosB);
    csa(carry, fours, fours, foursA, foursB);
#pragma unroll
    for (int p = 0; p < NC; ++p) {
        const uint64_t t = c[p] & carry;
        c[p] ^= carry;
        carry = t;
    }
}

// rls_tile.h: lv_merge_planes, on nine planes
template <int X>
__device__ __forceinline__ void lv_merge_planes9(uint64_t (&pl)[9], uint64_t take) {
    uint64_t carry = 0;
#pragma unroll
    for (int p = 0; p < 9; ++p) csa(carry, pl[p], pl[p], lv_lane_xor<X>(pl[p]) & take, carry);
}

// The same eight words at weight TWO: they enter at the `twos` plane (pass 0 counts a not-yet-visited neighbour twice, C = cV + 2 cF:
// one counter takes both kinds, each at its own weight -- a second set of planes for cF and the plane-by-plane sum at the end cost
// 16 registers and NP more adders).  NCW = planes above `fours` (>= 1).
template <int NCW>
__device__ __forceinline__ void lv_add8_x2(const uint64_t (&d)[8], uint64_t& twos, uint64_t& fours, uint64_t (&c)[5]) {
    uint64_t fA, fB, eA, eB, carry;
    csa(fA, twos, twos, d[0], d[1]);
    csa(fB, twos, twos, d[2], d[3]);
    csa(eA, fours, fours, fA, fB);
    csa(fA, twos, twos, d[4], d[5]);
    csa(fB, twos, twos, d[6], d[7]);
    csa(eB, fours, fours, fA, fB);
    csa(carry, c[0], c[0], eA, eB);
#pragma unroll
    for (int p = 1; p < NCW; ++p) {
        const uint64_t t = c[p] & carry;
        c[p] ^= carry;
        carry = t;
    }
}

// One lane = node group: the vertical counters over its rounds, C = cV (+ 2 cF in pass 0) plane by plane, and the
// bit-sliced compare with K.  NC = counter planes above `fours`, NP = planes of C that can be set; both follow the
// number of rounds (wave-uniform), see the dispatch in the kernel.  Neighbour entries are LDS BYTE offsets of the
// neighbour's word (| fresh << 31).
// A node of long degree occupies L = 2, 4 or 8 ADJACENT lanes (lcode = log2 L; lane j of them holds neighbours j, j + L,
// ...): the partial counters of those lanes are added across them before the compare, so a level's longest row
// costs deg / L rounds instead of deg -- the group's rounds are what a level waits for.
typedef const uint64_t __attribute__((address_space(3))) lds_cu64;
// the tile word at LDS byte address `a` (the kernel's dynamic LDS starts at address 0: checked at its top) -- a table entry goes
// into the read instruction as it was loaded, where `words + (entry & mask)` cost an AND and the add of the LDS base per neighbour
__device__ __forceinline__ uint64_t lds_word_at(uint32_t a) { return *(lds_cu64*)(uintptr_t)a; }

// eight neighbour words of a lane into its counter (pass 0: a not-yet-visited neighbour counts twice)
template <int NC, int NCW>
__device__ __forceinline__ void lv_count_block(const uint32_t (&e)[8], bool pass0, uint64_t& vo, uint64_t& vt, uint64_t& vf, uint64_t (&vc)[5]) {
    constexpr uint32_t M31 = 0x7fffffffu;
    uint64_t d[8];
    if (pass0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = lds_word_at(e[q] & M31);
        uint64_t dv[8], df[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint64_t fm = 0ull - (uint64_t)(e[q] >> 31);
            df[q] = d[q] & fm;
            dv[q] = d[q] & ~fm;
        }
        lv_add8<NCW>(dv, vo, vt, vf, vc);
        lv_add8_x2<NCW>(df, vt, vf, vc);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = lds_word_at(e[q]);   // (the flag-free copy of the table)
        lv_add8<NC>(d, vo, vt, vf, vc);
    }
}

// NB = the group's blocks of 8 rounds when that is 1 or 2 -- what most groups have, straight-line code on the registers the
// header came with --, 0: any number, a loop that requests block b + 2 while block b is counted
template <int NB, int NC, int NP>
__device__ __forceinline__ void lv_node_group(const int32_t* __restrict__ blk, int rounds, const uint32_t (&e0)[8], const uint32_t (&e1)[8],
                                              bool pass0, uint32_t S, uint32_t lcode, uint64_t coin, uint64_t& nw) {
    constexpr int NCW = NP - 3;                                   // pass 0: C = cV + 2 cF <= 3 rounds needs every plane of C
    static_assert(NCW >= 1 && NCW <= 5 && NC <= NCW, "planes");
    uint64_t vo = 0, vt = 0, vf = 0, vc[5] = {0, 0, 0, 0, 0};      // ones among the neighbours (pass 0: visited x 1 + not-yet-visited x 2)
    // blocks 0 and 1 came with the header (requested before the level boundary): a group of 8 or 16 rounds, which is what a level
    // usually waits for, issues no load of its own
    if constexpr (NB == 1) {
        lv_count_block<NC, NCW>(e0, pass0, vo, vt, vf, vc);
    } else if constexpr (NB == 2) {
        lv_count_block<NC, NCW>(e0, pass0, vo, vt, vf, vc);
        lv_count_block<NC, NCW>(e1, pass0, vo, vt, vf, vc);
    } else {
        uint32_t e[8], nx[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { e[q] = e0[q]; nx[q] = e1[q]; }
        for (int r0 = 0; r0 < rounds; r0 += 8, blk += 512) {      // blk: this lane's slab of block r0 / 8 + 2
            u32x4 na, nb;                                         // (read below only where they were loaded)
            if (r0 + 16 < rounds) {                               // (rounds are a multiple of 8: a block exists whole)
                na = *reinterpret_cast<const u32x4*>(blk);
                nb = *reinterpret_cast<const u32x4*>(blk + 256);
            }
            lv_count_block<NC, NCW>(e, pass0, vo, vt, vf, vc);
#pragma unroll
            for (int q = 0; q < 8; ++q) e[q] = nx[q];
            nx[0] = na.x; nx[1] = na.y; nx[2] = na.z; nx[3] = na.w;
            nx[4] = nb.x; nx[5] = nb.y; nx[6] = nb.z; nx[7] = nb.w;
        }
    }
    const int gl = __builtin_amdgcn_readlane((int)lcode, 0);      // lanes are sorted by L: lane 0 has the group's largest
    if (gl > 0) {
        // (a lane holds <= 64 entries: its own count fits the 8 planes, C <= 128; the node's -- up to 8 lanes, degree <= 128,
        // C <= 256 in pass 0 -- needs the ninth)
        uint64_t pv[9] = {vo, vt, vf, vc[0], vc[1], vc[2], vc[3], vc[4], 0};
        lv_merge_planes9<1>(pv, 0ull - (uint64_t)(lcode >= 1u));
        if (gl > 1) lv_merge_planes9<2>(pv, 0ull - (uint64_t)(lcode >= 2u));
        if (gl > 2) lv_merge_planes9<4>(pv, 0ull - (uint64_t)(lcode >= 3u));
        nw = lv_le_const_x2<9, 9>(pv, ~coin, S);
        return;
    }
    const uint64_t pl[9] = {vo, vt, vf, vc[0], vc[1], vc[2], vc[3], vc[4], 0};
    nw = lv_le_const_x2<NP, 9>(pl, ~coin, S);
}

// One node of high degree, lanes share its neighbours: per-lane vertical counters over the node's <= 2^NP - 1 rounds
// (ones among visited / not-yet-visited neighbours), then every plane is transposed across the wave and
// popcounted -- afterwards lane = chain holds that chain's counts.  NP follows the number of rounds, so a node of
// degree <= 64 (one round) costs one transpose per counter.
template <int NP>
__device__ __forceinline__ void lv_hub_counts(const uint64_t* words, const int32_t* __restrict__ data, int64_t p0,
                                              int rounds, const uint32_t (&e0)[8], bool pass0, int lane,
                                              const BitXpose& xc, int& cV, int& cF) {
    constexpr uint32_t M31 = 0x7fffffffu;
    uint64_t cv[NP], cf[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) { cv[p] = 0; cf[p] = 0; }
    auto add = [&](uint32_t en) {
        const uint64_t d = *reinterpret_cast<const uint64_t*>(reinterpret_cast<const unsigned char*>(words) + (en & M31));
        const uint64_t fm = pass0 ? 0ull - (uint64_t)(en >> 31) : 0ull;
        uint64_t carry = d & ~fm;
#pragma unroll
        for (int p = 0; p < NP; ++p) { const uint64_t t = cv[p] & carry; cv[p] ^= carry; carry = t; }
        if (pass0) {
            carry = d & fm;
#pragma unroll
            for (int p = 0; p < NP; ++p) { const uint64_t t = cf[p] & carry; cf[p] ^= carry; carry = t; }
        }
    };
#pragma unroll
    for (int q = 0; q < 8; ++q)
        if (q < rounds) add(e0[q]);
    for (int r = 8; r < rounds; ++r)      // (lane-major record: include/rlsolver_hip.h)
        add((uint32_t)data[p0 + 128 + (int64_t)(r >> 3) * 512 + ((r >> 2) & 1) * 256 + lane * 4 + (r & 3)]);
    cV = 0;
    cF = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        uint32_t r0 = (uint32_t)cv[p], r1 = (uint32_t)(cv[p] >> 32);
        bit_transpose64(r0, r1, xc);
        cV += (__builtin_popcount(r0) + __builtin_popcount(r1)) << p;
    }
    if (pass0) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            uint32_t r0 = (uint32_t)cf[p], r1 = (uint32_t)(cf[p] >> 32);
            bit_transpose64(r0, r1, xc);
            cF += (__builtin_popcount(r0) + __builtin_popcount(r1)) << p;
        }
    }
}

// LDS: the bit tile (N + 2 words), 64 int32 slots for the cut reduction and the group offsets lv_ptr[] (G + 2 dwords; a
// wave reads them 64 at a time into a register and walks them with v_readlane) -- small enough that two workgroups fit
// a CU at N = 10^4 (2 x ~81.3 KB).
// (two 8-wave workgroups per CU at N = 10^4 are 4 waves per SIMD: the second launch bound keeps the kernel at <= 128 registers --
// without it the compiler spent 129 on the same code once a 64-bit division appeared in the prologue, and one workgroup per CU ran)
// (Requesting the next group's header BEFORE this group's work instead of after it -- ten more live registers, no spills once pass 0
// kept one counter -- is SLOWER, 4.19 -> 4.40 ms: the per-wave cycle stamps of -DRLS_K7_PROF show a wave waiting 16 cycles per group
// for its header, i.e. the request already hides under the level barrier, where a wave spends half its cycles.)
template <typename TI, typename TO, int P, int W>
__global__ __launch_bounds__(W * kWave, 4) void k_mcpg_local_search_levels(
    const typename ChainStore<TI>::type* __restrict__ xs_in, typename ChainStore<TO>::type* __restrict__ xs_out, int64_t N,
    int64_t C, int64_t tiles_in, const int32_t* __restrict__ lv_ptr, const int32_t* __restrict__ data, int64_t G,
    int64_t num_ls, const uint64_t* __restrict__ coins, uint64_t seed, const int32_t* __restrict__ eu,
    const int32_t* __restrict__ ev, int64_t E, float* __restrict__ expected, ChainIds ids) {
    constexpr uint32_t M30 = 0x3fffffffu;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* words = reinterpret_cast<uint64_t*>(smem);
    int* cut_slots = reinterpret_cast<int*>(smem + (size_t)(N + 2) * 8);          // [64]
    int32_t* lvl = cut_slots + kWave;                                             // [G + 2]
    const uint32_t* w32 = reinterpret_cast<const uint32_t*>(smem);
    const int lane = threadIdx.x & (kWave - 1);
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const int64_t c0 = mcpg_tile() * kWave;
    const int64_t c = c0 + lane;
    const bool valid = c < C;
    const int64_t CB = (C + kWave - 1) / kWave;               // 64-chain blocks = words per coins row
    // table entries are used as LDS addresses: the tile must sit at LDS address 0 (it does: this kernel has no static LDS)
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)smem != 0u) __builtin_trap();
    const int64_t clean_off = (int64_t)((uint32_t)lv_ptr[G] & M30) + 1024;      // the flag-free copy of the table (rls_mcpg_visit_levels)
    if (threadIdx.x == 0) words[N] = 0;                       // padding / idle lanes point here
    if (threadIdx.x < kWave) cut_slots[threadIdx.x] = 0;
    for (int64_t i = threadIdx.x; i <= G + 1; i += W * kWave) lvl[i] = i <= G ? lv_ptr[i] : lv_ptr[G];
    if constexpr (std::is_same<TI, Packed64>::value) tile_load_packed(xs_in, N, C, mcpg_tile(), tiles_in, words, threadIdx.x, W * kWave);
    else tile_load_bits_nodemajor<TI>(xs_in, N, C, c0, words, lane, w, W);
    const uint32_t blk_key = k7_fmix32((uint32_t)seed ^ k7_fmix32((uint32_t)(seed >> 32) ^
                                                                  k7_fmix32((uint32_t)(ids(c0) >> 6) * 0x9E3779B1u + 0x632BE5ABu)));   // the tile's global id
    const BitXpose xc = bit_xpose_consts(lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // the tile is complete before any wave reads a neighbour word
    auto lvp = [&](int64_t k) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane(lvl[k]); };
    int num_levels = 0;                                       // level boundaries per pass (wave-uniform)
    for (int64_t k0 = 0; k0 < G; k0 += kWave)
        num_levels += __builtin_popcountll(ballot64(k0 + lane < G && (lvl[k0 + lane < G ? k0 + lane : G] >> 31) != 0));
    auto coin_word = [&](int64_t cnt, uint32_t pos) -> uint64_t {   // bit e: "u < 1/2" for chain c0 + e at (pass, pos)
        if (coins) return coins[((int64_t)cnt * N + (pos < (uint32_t)N ? pos : 0u)) * CB + mcpg_tile()];   // idle lanes carry pos = N: past the last row
        const uint32_t k = blk_key ^ (pos * 0x9E3779B1u) ^ ((uint32_t)cnt * 0x7FEB352Du + 0x165667B1u);
        return ((uint64_t)k7_fmix32(k ^ 0x4C4F4353u) << 32) | k7_fmix32(k + 0x27D4EB2Fu);
    };
#ifdef RLS_K7_PROF
    unsigned long long pf_bar = 0, pf_hdr = 0, pf_comp = 0, pf_n = 0, pf_hub = 0;
    const unsigned long long pf_t0 = K7_NOW();
#endif
    for (int64_t cnt = 0; cnt < num_ls; ++cnt) {
        const bool pass0 = cnt == 0;
        int64_t mine = w;
        uint32_t h0 = (uint32_t)N, h1 = (uint32_t)N, e0[8], e1[8];
        auto prefetch = [&](int64_t k) {
            if (k < G) {
                // this lane's two header words and its first TWO blocks of eight rounds: five wide loads, unguarded (a record is whole
                // blocks and the table ends in sixteen spare rows).  Passes >= 1 read the entries of a lane = node group from the
                // flag-free copy of the table
                const uint32_t lp = lvp(k);
                const int32_t* rec = data + (lp & M30);
                const int32_t* ent = (pass0 || ((lp >> 30) & 1u)) ? rec : rec + clean_off;
                const uint2 hh = *reinterpret_cast<const uint2*>(rec + 2 * lane);
                const u32x4 a = *reinterpret_cast<const u32x4*>(ent + 128 + 4 * lane), b = *reinterpret_cast<const u32x4*>(ent + 384 + 4 * lane);
                const u32x4 c = *reinterpret_cast<const u32x4*>(ent + 640 + 4 * lane), d = *reinterpret_cast<const u32x4*>(ent + 896 + 4 * lane);
                h0 = hh.x;
                h1 = hh.y;
                e0[0] = a.x; e0[1] = a.y; e0[2] = a.z; e0[3] = a.w;
                e0[4] = b.x; e0[5] = b.y; e0[6] = b.z; e0[7] = b.w;
                e1[0] = c.x; e1[1] = c.y; e1[2] = c.z; e1[3] = c.w;
                e1[4] = d.x; e1[5] = d.y; e1[6] = d.z; e1[7] = d.w;
            }
        };
        prefetch(mine);
        // A wave walks ITS groups only (w, w + W, ...) and meets the others once per level boundary it crosses: the level
        // of group k = the number of level-start flags up to k, counted from a ballot over the 64 offsets the wave holds
        // in registers.  (Walking all G groups to find the boundaries cost every wave ~5 instructions per group and pass.)
        int chunk = 0, chunk_next = 0;                        // lvl[k0 + lane] and lvl[k0 + 1 + lane] of the current 64 groups
        int64_t cbase = -1;
        uint64_t lmask = 0;                                   // level-start flags of the current 64 groups
        int lev_base = 0, passed = 0;                         // level starts before the chunk; barriers done in this pass
        for (int64_t k = mine; k < G; k += W) {
            if ((k & ~(int64_t)63) != cbase) {
                cbase = k & ~(int64_t)63;
                lev_base += __builtin_popcountll(lmask);
                const int64_t a0 = cbase + lane <= G ? cbase + lane : G, a1 = cbase + 1 + lane <= G ? cbase + 1 + lane : G;
                chunk = lvl[a0];
                chunk_next = lvl[a1];
                lmask = ballot64(cbase + lane < G && (chunk >> 31) != 0);
            }
            const uint32_t flags = (uint32_t)__builtin_amdgcn_readlane(chunk, (int)(k & 63));
            const int need = lev_base + __builtin_popcountll(lmask & ((2ull << (k & 63)) - 1ull));
#ifdef RLS_K7_PROF
            const unsigned long long pf_a = K7_NOW();
#endif
            for (; passed < need; ++passed) __syncthreads();  // new level (k = 0: new pass): earlier updates are visible
#ifdef RLS_K7_PROF
            const unsigned long long pf_b = K7_NOW();
            pf_bar += pf_b - pf_a;
#endif
            const int64_t p0 = flags & M30, p1 = (uint32_t)__builtin_amdgcn_readlane(chunk_next, (int)(k & 63)) & M30;
            const int rounds = (int)((p1 - p0) >> 6) - 2;
#ifdef RLS_K7_PROF
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long pf_c = K7_NOW();
            pf_hdr += pf_c - pf_b;
#endif
            if (!((flags >> 30) & 1u)) {
                // ---- 64 lanes of nodes (a long row takes 2, 4 or 8 adjacent lanes), K in 8 bits, log2 lanes-per-node above it
                const uint32_t node = h0 & 0xFFFFFu, pos = h1 & 0xFFFFFu;
                const uint32_t K = pass0 ? ((h1 >> 20) & 0xFFu) : ((h0 >> 20) & 0xFFu);
                const uint32_t lcode = (h0 >> 28) & 3u;
                // new bit = [C < K] | ([C == K] & tie & coin) with K = ceil(S / 2), tie = S even (S = the degree; + the not-yet-
                // visited neighbours in pass 0)  <=>  2 C + !coin <= S: ONE bit-sliced "<= constant" on the counter shifted up a
                // plane with the coin below it (carry chain of D + ~S, rls_tile.h) -- the scan that kept "smaller so far" and
                // "equal so far" per plane cost ~2.5x the instructions
                const uint32_t S = 2u * K - 1u + ((pass0 ? h1 : h0) >> 31);
                uint64_t nw;
                const uint64_t coin = coin_word(cnt, pos);
                // counts <= rounds, C = cV + 2 cF <= 2 rounds (the not-yet-visited neighbours count twice in pass 0)
                const int32_t* blk = (pass0 ? data : data + clean_off) + p0 + 128 + 1024 + 4 * lane;    // this lane's slab of block 2
                if (rounds == 8) lv_node_group<1, 1, 6>(blk, rounds, e0, e1, pass0, S, lcode, coin, nw);
                else if (rounds == 16) lv_node_group<2, 2, 7>(blk, rounds, e0, e1, pass0, S, lcode, coin, nw);
                else if (rounds < 8) lv_node_group<0, 1, 6>(blk, rounds, e0, e1, pass0, S, lcode, coin, nw);   // (isolated nodes: no round at all)
                else if (rounds < 32) lv_node_group<0, 2, 7>(blk, rounds, e0, e1, pass0, S, lcode, coin, nw);
                else lv_node_group<0, 4, 8>(blk, rounds, e0, e1, pass0, S, lcode, coin, nw);
                if (node < (uint32_t)N && (lane & ((1 << lcode) - 1)) == 0) words[node] = nw;
            } else {
                // ---- one node of high degree, lane = neighbour
                const uint32_t g0 = (uint32_t)__builtin_amdgcn_readlane((int)h0, 0);
                const uint32_t g1 = (uint32_t)__builtin_amdgcn_readlane((int)h1, 0);
                const uint32_t node = g0 & 0xFFFFFu, pos = g1 & 0xFFFFFu;
                const uint32_t K = pass0 ? ((g1 >> 20) & 0x7FFu) : ((g0 >> 20) & 0x7FFu);
                const bool tie = (pass0 ? g1 : g0) >> 31;
                int cV, cF;                                                   // lane = chain after the call
                const int rounds = ((int)__builtin_amdgcn_readlane((int)h0, 2) + 63) >> 6;     // the hub's own rounds: ceil(deg / 64)
                if (rounds <= 1) lv_hub_counts<1>(words, data, p0, rounds, e0, pass0, lane, xc, cV, cF);
                else if (rounds <= 3) lv_hub_counts<2>(words, data, p0, rounds, e0, pass0, lane, xc, cV, cF);
                else if (rounds <= 7) lv_hub_counts<3>(words, data, p0, rounds, e0, pass0, lane, xc, cV, cF);
                else lv_hub_counts<5>(words, data, p0, rounds, e0, pass0, lane, xc, cV, cF);
                const uint32_t Cc = (uint32_t)(cV + 2 * cF);
                const uint64_t coin = coin_word(cnt, pos);
                const bool bit = (Cc < K) || (Cc == K && tie && ((coin >> lane) & 1ull));
                const uint64_t nw = ballot64(bit);
                if (lane == 0) words[node] = nw;
            }
#ifdef RLS_K7_PROF
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long pf_d = K7_NOW();
            pf_comp += pf_d - pf_c;
            if ((flags >> 30) & 1u) pf_hub += pf_d - pf_c;
            ++pf_n;
#endif
            prefetch(k + W);
        }
#ifdef RLS_K7_PROF
        const unsigned long long pf_e = K7_NOW();
#endif
        for (; passed < num_levels; ++passed) __syncthreads();   // every wave crosses every boundary of the pass
#ifdef RLS_K7_PROF
        pf_bar += K7_NOW() - pf_e;
#endif
    }
#ifdef RLS_K7_PROF
    if (lane == 0 && blockIdx.x < 2048 && blockIdx.y == 0) {
        unsigned long long* q = g_k7_prof + ((size_t)blockIdx.x * 16 + w) * 6;
        q[0] = K7_NOW() - pf_t0; q[1] = pf_bar; q[2] = pf_hdr; q[3] = pf_comp; q[4] = pf_n; q[5] = pf_hub;
    }
#endif
    __syncthreads();
    // K8: expected[c] = sum_e (2x_u - 1)(2x_v - 1) = E - 2 * cut; the W partial counts meet in 64 LDS slots
    const int part = (int)tile_cut_count<P>(words, eu, ev, E, lane, w, W);
    atomicAdd(&cut_slots[lane], part);
    __syncthreads();
    if (valid && w == 0) expected[c] = (float)(E - 2 * (int64_t)cut_slots[lane]);
    if constexpr (std::is_same<TO, Packed64>::value) {
        tile_store_packed(xs_out, N, mcpg_tile(), words, threadIdx.x, W * kWave);
    } else {
        if (valid) {
            const int half = lane >> 5, sh = lane & 31;
            for (int64_t n = w; n < N; n += W) xs_out[n * C + c] = (float)((w32[(n << 1) + half] >> sh) & 1u);
        }
    }
}

// best-of-repeats: index = argmin_r expected[r*M + m] (first on ties), then the column gather as its own
// grid over (node, m) -- one thread per m walking all N rows took 5 ms at N = 10^4, three times the K7 kernel
__global__ void k_mcpg_pick_argmin(const float* __restrict__ expected, int64_t M, int64_t R, float num_edges,
                                   int64_t* __restrict__ best_index, float* __restrict__ vs_good) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float best = expected[m];
    int64_t br = 0;
#pragma unroll 8
    for (int64_t r = 1; r < R; ++r) {
        const float v = expected[r * M + m];
        if (v < best) { best = v; br = r; }
    }
    best_index[m] = m + br * M;
    vs_good[m] = (num_edges - best) / 2.0f;                                   // MCPG.py:160
}

__global__ void k_mcpg_pick_gather(const float* __restrict__ xs, int64_t N, int64_t M, int64_t Ctot,
                                   const int64_t* __restrict__ best_index, float* __restrict__ xs_good) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < N * M; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = t / M, m = t - n * M;                           // t = n * M + m: coalesced writes
        xs_good[t] = xs[n * Ctot + best_index[m]];
    }
}

// best-of-repeats gather on bit-packed chains: kept tile j, node n: bit e = node n of chain best_index[64 j + e].
// lane = node; for each of the 64 kept chains one coalesced read of 64 consecutive words of the winner's tile.
__global__ __launch_bounds__(256) void k_mcpg_pick_gather_packed(const uint64_t* __restrict__ xs, int64_t N, int64_t M,
                                                                 const int64_t* __restrict__ best_index,
                                                                 uint64_t* __restrict__ xs_good) {
    const int64_t j = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ int64_t src[kWave];
    if (threadIdx.x < kWave) {
        const int64_t m = j * kWave + threadIdx.x;
        src[threadIdx.x] = m < M ? best_index[m] : -1;
    }
    __syncthreads();
    if (n >= N) return;
    uint64_t out = 0;
#pragma unroll 8
    for (int e = 0; e < kWave; ++e) {
        const int64_t bi = src[e];
        if (bi < 0) break;
        out |= ((xs[(bi >> 6) * N + n] >> (bi & 63)) & 1ull) << e;
    }
    xs_good[j * N + n] = out;
}

// Best-merge of the MCPG outer loop, methods/MCPG.py:376-391, on bit-packed kept chains (M = total_mcmc_num):
//   (a) per chain m: if temp_max[m] > now_max_res[m] the incumbent takes the value and the chain     (:377-380)
//   (b) the globally best incumbent (first argmax) overwrites the worst (first argmin): value, incumbent
//       column and the column of temp_max_info that seeds the next round                              (:383-391)
// k_mcpg_merge_mask does (a) on the values and leaves one mask word per 64 chains; k_mcpg_merge_apply moves the bits;
// k_mcpg_merge_minmax does (b) in one workgroup.
__global__ void k_mcpg_merge_mask(const float* __restrict__ temp_max, float* __restrict__ now_max_res, int64_t M,
                                  uint64_t* __restrict__ mask) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = m < M;
    const float t = in ? temp_max[m] : 0.0f, r = in ? now_max_res[m] : 0.0f;
    const bool take = in && t > r;
    if (take) now_max_res[m] = t;
    const uint64_t w = ballot64(take);
    if ((threadIdx.x & 63) == 0 && (m >> 6) < ((M + 63) >> 6)) mask[m >> 6] = w;   // exactly ceil(M / 64) words
}

__global__ void k_mcpg_merge_apply(const uint64_t* __restrict__ temp_info, uint64_t* __restrict__ now_info, int64_t N,
                                   int64_t tiles, const uint64_t* __restrict__ mask) {
    const int64_t total = tiles * N;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t mk = mask[t / N];
        if (mk) now_info[t] = (now_info[t] & ~mk) | (temp_info[t] & mk);
    }
}

__global__ __launch_bounds__(1024) void k_mcpg_merge_minmax(float* __restrict__ now_max_res, int64_t M, int64_t N,
                                                            uint64_t* __restrict__ now_info, uint64_t* __restrict__ temp_info,
                                                            float* __restrict__ best_value, int64_t* __restrict__ best_index,
                                                            int replace_worst) {
    __shared__ float s_hi[16], s_lo[16];
    __shared__ int64_t s_hii[16], s_loi[16];
    float hi = -INFINITY, lo = INFINITY;
    int64_t hii = INT64_MAX, loi = INT64_MAX;
    for (int64_t m = threadIdx.x; m < M; m += blockDim.x) {
        const float v = now_max_res[m];
        if (v > hi) { hi = v; hii = m; }
        if (v < lo) { lo = v; loi = m; }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        const float oh = __shfl_xor(hi, s, 64), ol = __shfl_xor(lo, s, 64);
        const int64_t ohi = __shfl_xor(hii, s, 64), oli = __shfl_xor(loi, s, 64);
        if (oh > hi || (oh == hi && ohi < hii)) { hi = oh; hii = ohi; }
        if (ol < lo || (ol == lo && oli < loi)) { lo = ol; loi = oli; }
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_hi[wv] = hi; s_hii[wv] = hii; s_lo[wv] = lo; s_loi[wv] = loi; }
    __syncthreads();
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) {
        if (s_hi[k] > hi || (s_hi[k] == hi && s_hii[k] < hii)) { hi = s_hi[k]; hii = s_hii[k]; }
        if (s_lo[k] < lo || (s_lo[k] == lo && s_loi[k] < loi)) { lo = s_lo[k]; loi = s_loi[k]; }
    }
    if (!replace_worst) {   // a shard of the kept chains: the caller replaces the GLOBAL worst by the GLOBAL best itself
        if (threadIdx.x == 0) {
            best_value[0] = hi; best_value[1] = lo;
            best_index[0] = hii; best_index[1] = loi;
        }
        return;
    }
    if (threadIdx.x == 0) {
        now_max_res[loi] = hi;                                                  // :388
        if (best_value) best_value[0] = hi;
        if (best_index) best_index[0] = hii;
    }
    const uint64_t* srcw = now_info + (hii >> 6) * N;
    uint64_t* d0 = now_info + (loi >> 6) * N;
    uint64_t* d1 = temp_info + (loi >> 6) * N;
    const int sb = (int)(hii & 63), db = (int)(loi & 63);
    const bool same = hii == loi;                                               // one kept chain, or all incumbents equal: :389 is a
    for (int64_t n = threadIdx.x; n < N; n += blockDim.x) {                     // no-op, :390 still seeds the next round from the incumbent
        const uint64_t bit = (srcw[n] >> sb) & 1ull;                            // read before either write: d0 may alias srcw's tile
        const uint64_t a = d0[n], b = d1[n];
        if (!same) d0[n] = (a & ~(1ull << db)) | (bit << db);                   // :389
        d1[n] = (b & ~(1ull << db)) | (bit << db);                              // :390
    }
}

// get_return, methods/MCPG.py:292-302, reduced to what it needs: with s in {0,1},
//   sum_n log(s p + (1 - s)(1 - p)) = sum_n log(1 - p_n) + sum_n s_n (log p_n - log(1 - p_n)),
// so the objective mean_b(log_prob_sum_b * value_b) and its gradient need only  A_n = sum_b value_b s_bn  and
// V = sum_b value_b.  One workgroup per (64-chain tile, 256-node slab): lane = node, the tile's 64 values broadcast
// from LDS, float atomics into A[N] (zeroed by the caller).
__global__ __launch_bounds__(256) void k_mcpg_value_bit_sums(const uint64_t* __restrict__ samples, int64_t N, int64_t C,
                                                             const float* __restrict__ value, float* __restrict__ A) {
    __shared__ float v[kWave];
    const int64_t tile = blockIdx.y;
    if (threadIdx.x < kWave) {
        const int64_t c = tile * kWave + threadIdx.x;
        v[threadIdx.x] = c < C ? value[c] : 0.0f;
    }
    __syncthreads();
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const uint64_t wd = samples[tile * N + n];
    float acc = 0.0f;
#pragma unroll 16
    for (int e = 0; e < kWave; ++e) acc += ((wd >> e) & 1ull) ? v[e] : 0.0f;
    if (acc != 0.0f) atomicAdd(&A[n], acc);
}

// The same sums for N that fits LDS (4 N + 8 KB): a workgroup walks tiles blockIdx.x, + gridDim.x, ... and keeps its
// partial A[] in LDS, so the atomics fall from one per (tile, node) -- 41 M onto 10^4 addresses at BA-10^4 / 2^18, the
// whole cost of the kernel above -- to one per (workgroup, node).  Per tile the 64 values become 8 tables of 256 partial
// sums (table q, entry i = sum of the values of chains 8q + j over the set bits j of i): a word costs 8 lookups + 8 adds
// instead of 64 selects + 64 adds.
constexpr int kBitSumThreads = 512;   // (256: 117 us at BA-1e4 / 2^18; the lookups want more waves per CU than 3 workgroups x 4 gave)
constexpr int kBitSumRegs = 24;       // nodes per thread whose partial sums stay in registers: N <= 24 * 512
template <bool REGS>                  // REGS: the workgroup's partial A[] lives in registers (thread t owns nodes t, t + 512, ...), else in LDS
__global__ __launch_bounds__(kBitSumThreads) void k_mcpg_value_bit_sums_lut(const uint64_t* __restrict__ samples, int64_t N,
                                                                            int64_t C, int64_t tiles,
                                                                            const float* __restrict__ value,
                                                                            float* __restrict__ A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lut = reinterpret_cast<float*>(smem);          // [8][256]
    float* v = lut + 8 * 256;                             // [64]
    float* acc = v + kWave;                               // [N] (not REGS)
    const int t = threadIdx.x;
    float accr[kBitSumRegs];
    if constexpr (REGS) {
#pragma unroll
        for (int u = 0; u < kBitSumRegs; ++u) accr[u] = 0.0f;
    } else {
        for (int64_t n = t; n < N; n += kBitSumThreads) acc[n] = 0.0f;
    }
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        __syncthreads();                                  // the previous tile's lookups are done
        if (t < kWave) {
            const int64_t c = tile * kWave + t;
            v[t] = c < C ? value[c] : 0.0f;
        }
        __syncthreads();
        if (t < 256) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float sum = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; ++j) sum += ((t >> j) & 1) ? v[8 * q + j] : 0.0f;
                lut[q * 256 + t] = sum;
            }
        }
        __syncthreads();
        const uint64_t* row = samples + tile * N;
        if constexpr (REGS) {
            // four words in flight per trip; the sums never leave the registers between tiles (the LDS form read and wrote
            // acc[n] once per word: two of its ten LDS operations)
#pragma unroll
            for (int u0 = 0; u0 < kBitSumRegs; u0 += 4) {
                if ((int64_t)u0 * kBitSumThreads >= N) break;            // (uniform)
                uint64_t wd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t n = t + (int64_t)(u0 + u) * kBitSumThreads;
                    wd[u] = n < N ? row[n] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float sum = 0.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += lut[q * 256 + (int)((wd[u] >> (8 * q)) & 255u)];     // (word 0: entry 0 of every table = 0)
                    accr[u0 + u] += sum;
                }
            }
        } else {
            for (int64_t n0 = t; n0 < N; n0 += 4 * kBitSumThreads) {
                uint64_t wd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t n = n0 + (int64_t)u * kBitSumThreads;
                    wd[u] = n < N ? row[n] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t n = n0 + (int64_t)u * kBitSumThreads;
                    if (n >= N) break;
                    float sum = 0.0f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum += lut[q * 256 + (int)((wd[u] >> (8 * q)) & 255u)];
                    acc[n] += sum;                            // n = t mod the workgroup size: this thread's own slot
                }
            }
        }
    }
    if constexpr (REGS) {
#pragma unroll
        for (int u = 0; u < kBitSumRegs; ++u) {
            const int64_t n = t + (int64_t)u * kBitSumThreads;
            if (n < N && accr[u] != 0.0f) atomicAdd(&A[n], accr[u]);
        }
    } else {
        for (int64_t n = t; n < N; n += kBitSumThreads)
            if (acc[n] != 0.0f) atomicAdd(&A[n], acc[n]);
    }
}

// bit-packed tiles <-> the reference's node-major f32 [N, C] surface (shims for callers that want it)
// unpack: a thread writes four consecutive chains of one node as one 16-byte store (rows of C % 4 == 0 floats are 16-byte
// aligned); grid.y walks the nodes, so no division per element
template <bool V4, bool NT = false, int U = 1>
__global__ __launch_bounds__(256) void k_mcpg_unpack(const uint64_t* __restrict__ packed, int64_t N, int64_t C, float* __restrict__ xs) {
    if constexpr (V4) {
        const int64_t c4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // chains 4 c4 .. 4 c4 + 3
        if (c4 * 4 >= C) return;
        const int64_t tile = c4 >> 4;
        const int sh = (int)(c4 & 15) * 4;
        for (int64_t n0 = (int64_t)blockIdx.y * U; n0 < N; n0 += (int64_t)gridDim.y * U) {     // U consecutive rows: their words are adjacent
            uint64_t wd[U];
#pragma unroll
            for (int k = 0; k < U; ++k) wd[k] = n0 + k < N ? packed[tile * N + n0 + k] : 0ull;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (n0 + k >= N) break;
                const uint32_t b = (uint32_t)(wd[k] >> sh) & 15u;
                f32x4 v;
                v[0] = (float)(b & 1u); v[1] = (float)((b >> 1) & 1u); v[2] = (float)((b >> 2) & 1u); v[3] = (float)(b >> 3);
                if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(xs + (n0 + k) * C + c4 * 4));
                else *reinterpret_cast<f32x4*>(xs + (n0 + k) * C + c4 * 4) = v;
            }
        }
    } else {
        const int64_t total = N * C;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t n = t / C, c = t - n * C;
            xs[t] = (float)((packed[(c >> 6) * N + n] >> (c & 63)) & 1ull);
        }
    }
}

// pack: a workgroup turns 64 nodes x kPackTiles consecutive 64-chain tiles; each of its 4 waves takes 16 of the rows, all
// 16 loads of a tile in flight and the next tile's requested before this one's ballots (one load -> ballot per trip was a
// chain of memory round trips, and one tile per workgroup left the launch rate of its short-lived waves as the bound)
constexpr int kPackTiles = 8;
template <typename T>
__global__ __launch_bounds__(4 * kWave) void k_mcpg_pack(const T* __restrict__ xs, int64_t N, int64_t C, uint64_t* __restrict__ packed) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    const int64_t tiles = (C + kWave - 1) / kWave;
    const int64_t t0 = (int64_t)blockIdx.x * kPackTiles;
    const int64_t n0 = (int64_t)blockIdx.y * kWave + 16 * w;
    auto fetch = [&](int64_t tile, T (&v)[16]) {
        const int64_t c = tile * kWave + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (tile < tiles && c < C && n0 + k < N) ? xs[(n0 + k) * C + c] : T(0);
    };
    T cur[16], nxt[16];
    fetch(t0, cur);
#pragma unroll 1
    for (int i = 0; i < kPackTiles && t0 + i < tiles; ++i) {
        fetch(t0 + i + 1 < t0 + kPackTiles ? t0 + i + 1 : tiles, nxt);
        uint64_t mine = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t wd = ballot64(spin_is_set(cur[k]));
            if (lane == k) mine = wd;
        }
        if (lane < 16 && n0 + lane < N) packed[(t0 + i) * N + n0 + lane] = mine;
#pragma unroll
        for (int k = 0; k < 16; ++k) cur[k] = nxt[k];
    }
}

// ---- round 5: the f32 shims as line-wide streams ----------------------------------------------------------------------------
// What a caller of the reference-shaped surface (f32 [N, C], MCPG.py:88-166) pays beside the bit-packed kernels is 4 N bytes per
// chain each way, so the shims must run at what a plain stream reaches.  pack takes one 16-byte vector per lane: a wave-instruction
// moves 1 KB of ONE row = 256 chains = four 64-chain tiles (the round-4 kernel moved 256 B per instruction).  (unpack already wrote
// 16 bytes per lane; the same 16-rows-per-wave geometry for it, and a row-sequential form, measured SLOWER than one row per thread:
// 2220 / 2100 vs 2020 us -- what it lacked were non-temporal stores.)
//
// pack: lane l holds chains 4 l .. 4 l + 3 of the span as a nibble; the 16 lanes of a DPP row are one tile: the nibble shifted to
// its place in the 64-bit word (lanes 0..7 fill the low dword, 8..15 the high one) and OR-ed over the row in four DPP steps (the two
// quad permutes, rotate by 4, rotate by 8) -- afterwards every lane of the row holds the tile's word.  A wave takes 16 rows; lane
// 16 t + k keeps row k's word of tile t, so the result leaves as 16 consecutive words (128 B) per tile.  ~25 VALU per KB read.
__device__ __forceinline__ uint32_t row_or16(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);   // row_ror:4
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);   // row_ror:8
    return v;
}
__global__ __launch_bounds__(4 * kWave) void k_mcpg_pack_f32x4(const float* __restrict__ xs, int64_t N, int64_t C, uint64_t* __restrict__ packed, int spans) {
    const int lane = threadIdx.x & (kWave - 1);
    const int w = threadIdx.x / kWave;
    const int64_t tiles = (C + kWave - 1) / kWave;
    const int64_t n0 = ((int64_t)blockIdx.y * 4 + w) * 16;
    if (n0 >= N) return;
    const int sh = 4 * (lane & 7);
    const bool hi_half = (lane & 8) != 0;
    auto fetch = [&](int64_t span, int r0, f32x4 (&v)[8]) {          // rows n0 + r0 .. + 7 of the span
        const int64_t c = span * 256 + 4 * lane;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            v[k] = (c < C && n0 + r0 + k < N) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xs + (n0 + r0 + k) * C + c))
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const int64_t s0 = (int64_t)blockIdx.x * spans;
    f32x4 cur[8], nxt[8];
    fetch(s0, 0, cur);
    uint32_t keep_lo = 0, keep_hi = 0;
#pragma unroll 1
    for (int i = 0; i < 2 * spans; ++i) {                               // half-batches: (span, rows 0..7), (span, rows 8..15), ...
        const int64_t span = s0 + (i >> 1);
        if (span * 256 >= C) break;
        const int r0 = (i & 1) * 8;
        if (i + 1 < 2 * spans) fetch(s0 + ((i + 1) >> 1), ((i + 1) & 1) * 8, nxt);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t nib = (cur[k][0] > 0.0f ? 1u : 0u) | (cur[k][1] > 0.0f ? 2u : 0u) | (cur[k][2] > 0.0f ? 4u : 0u) |
                                 (cur[k][3] > 0.0f ? 8u : 0u);
            const uint32_t pl = nib << sh;
            const uint32_t lo = row_or16(hi_half ? 0u : pl), hi = row_or16(hi_half ? pl : 0u);
            if ((lane & 15) == r0 + k) { keep_lo = lo; keep_hi = hi; }
        }
        if (i & 1) {
            const int64_t tile = span * 4 + (lane >> 4), n = n0 + (lane & 15);
            if (tile < tiles && n < N) packed[tile * N + n] = ((uint64_t)keep_hi << 32) | keep_lo;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = nxt[k];
    }
}

// The stop rule of metro_sampling between two chunks of rounds (MCPG.py:103,115: stop after the first round whose cumulative
// accept count reaches C * T), as ONE small launch: column sums of accepts [rows][T], their running total on top of the earlier
// chunks', the first round that reaches the target.  ctl = {accepts so far, walk still live, limit of the next dry pass}.
//   first chunk (applied directly, all T rounds ran):  hit = total >= target
//   later chunks: *apply_limit = min(limit, stop round) for the apply pass of this chunk
// Before, this was ten [1]- to [T]-sized torch ops per chunk -- 70 launches per call, 440 us of dispatch around 20 us of kernels
// on a G22-sized graph.
constexpr int kStopWaves = 16;
__global__ __launch_bounds__(kStopWaves * kWave) void k_metro_stop(const int64_t* __restrict__ accepts, int64_t rows, int64_t T, int64_t target,
                                                                   int first, int64_t next_T, int64_t* __restrict__ ctl,
                                                                   int64_t* __restrict__ apply_limit) {
    // a wave owns a contiguous run of rounds, its lanes consecutive rounds (coalesced reads of every row); the column sums of up to
    // kKeep blocks of 64 rounds stay in registers between the two phases (segment totals, then the running count)
    constexpr int kKeep = 4;
    __shared__ int64_t seg[kStopWaves];
    __shared__ unsigned long long stop_at;
    const int lane = threadIdx.x & (kWave - 1), w = threadIdx.x / kWave;
    const int64_t blocks = (T + kWave - 1) / kWave;                         // blocks of 64 rounds
    const int64_t per = (blocks + kStopWaves - 1) / kStopWaves;             // blocks per wave
    const int64_t b0 = per * w, b1 = b0 + per < blocks ? b0 + per : blocks;
    auto colsum = [&](int64_t blk) {
        const int64_t t = blk * kWave + lane;
        int64_t v = 0;
        if (t < T) {
#pragma unroll 8
            for (int64_t r = 0; r < rows; ++r) v += accepts[r * T + t];
        }
        return v;
    };
    int64_t keep[kKeep];
    int64_t mine = 0;
#pragma unroll
    for (int k = 0; k < kKeep; ++k) {
        keep[k] = b0 + k < b1 ? colsum(b0 + k) : 0;
        mine += keep[k];
    }
    for (int64_t blk = b0 + kKeep; blk < b1; ++blk) mine += colsum(blk);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mine += __shfl_xor(mine, m, kWave);
    if (lane == 0) seg[w] = mine;
    if (threadIdx.x == 0) stop_at = ~0ull;
    __syncthreads();
    int64_t before = first == 1 ? 0 : ctl[0];     // first: 1 = the call's first chunk, 2 = a later chunk that was applied directly
    int64_t total = before;
    for (int k = 0; k < kStopWaves; ++k) {
        if (k < w) before += seg[k];
        total += seg[k];
    }
    // the first round of this wave's run at which the running count reaches the target
    int64_t run = before;
    bool done = false;
    auto step = [&](int64_t blk, int64_t v) {            // wave-uniform result: this block reaches the target
        int64_t inc = v;                                 // inclusive scan over the 64 lanes
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1) {
            const int64_t o = __shfl_up(inc, d, kWave);
            if (lane >= d) inc += o;
        }
        const uint64_t reached = ballot64(run + inc >= target && blk * kWave + lane < T);
        if (reached) {
            if (lane == 0) atomicMin(&stop_at, (unsigned long long)(blk * kWave + __builtin_ctzll(reached) + 1));
            return true;
        }
        run += __shfl(inc, kWave - 1, kWave);
        return false;
    };
#pragma unroll
    for (int k = 0; k < kKeep; ++k)
        if (!done && b0 + k < b1) done = step(b0 + k, keep[k]);
    for (int64_t blk = b0 + kKeep; blk < b1 && !done; ++blk) done = step(blk, colsum(blk));
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool live = first == 1 ? true : ctl[1] != 0;
        const int64_t limit = first ? T : ctl[2];
        const bool hit = first ? total >= target : stop_at != ~0ull;
        const int64_t t_stop = hit && !first ? (int64_t)stop_at : T;
        if (apply_limit) apply_limit[0] = limit < t_stop ? limit : t_stop;
        const bool live_next = live && !hit;
        ctl[0] = total;
        ctl[1] = live_next ? 1 : 0;
        ctl[2] = live_next ? next_T : 0;
    }
}

}  // namespace rls

using namespace rls;

// rls_chain_ids -> the kernels' by-value form + the launch grid for C chains; NULL = the identity (a single-process run, 1-D grid)
static int chain_ids_arg(const rls_chain_ids* in, int64_t C, ChainIds& out, dim3& grid) {
    out = ChainIds{0, 0};
    grid = dim3((unsigned)ceil_div(C, kWave));
    if (!in) return RLS_OK;
    RLS_REQUIRE(in->offset >= 0 && in->period >= 0 && in->skip >= 0, RLS_EINVAL, "chain_ids: negative offset / period / skip");
    out.offset = in->offset;
    if (in->period > 0 && in->skip > 0) {
        // repeats of `period` chains: one grid row per repeat (no division on the device)
        RLS_REQUIRE((in->period & (kWave - 1)) == 0 && C % in->period == 0 && C / in->period < 65536, RLS_EINVAL,
                    "chain_ids: period=%lld must be a multiple of 64 that divides C=%lld (fewer than 65536 repeats)", (long long)in->period,
                    (long long)C);
        out.skip = in->skip;
        grid = dim3((unsigned)(in->period / kWave), (unsigned)(C / in->period));
    }
    return RLS_OK;
}

extern "C" {

int rls_mcpg_metro_rounds(void* samples, const void* samples_in, int64_t C_in, int spin_bytes, int64_t N, int64_t C,
                          const float* probs, int64_t T, int64_t t_offset, const int64_t* index, const float* u,
                          uint64_t seed, const int64_t* t_limit_dev, int write_back, int64_t* accepts, int64_t accept_rows,
                          const rls_chain_ids* chain_ids, void* scratch, int64_t scratch_bytes, void* stream) {
    RLS_REQUIRE(N > 0 && C >= 0 && T >= 0 && t_offset >= 0, RLS_EINVAL, "bad sizes N=%lld C=%lld T=%lld", (long long)N, (long long)C,
                (long long)T);
    if (C == 0) return RLS_OK;
    ChainIds ids;
    dim3 cgrid;
    if (int rc = chain_ids_arg(chain_ids, C, ids, cgrid)) return rc;
    RLS_REQUIRE(samples && probs, RLS_EINVAL, "samples/probs is NULL");
    RLS_REQUIRE((index == nullptr) == (u == nullptr), RLS_EINVAL, "index and u must both be given or both be NULL");
    RLS_REQUIRE(spin_bytes == 0 || spin_bytes == 1 || spin_bytes == 4, RLS_EINVAL, "spin_bytes must be 0 (bit-packed), 1 or 4");
    RLS_REQUIRE(!accepts || (accept_rows >= 1 && accept_rows < (1ll << 31)), RLS_EINVAL, "accept_rows must be >= 1 when accepts is given");
    if (!accepts) accept_rows = 1;
    if (!samples_in) { samples_in = samples; C_in = C; }
    if (C_in <= 0) C_in = C;
    RLS_REQUIRE(samples_in == samples || write_back, RLS_EINVAL, "samples_in != samples needs write_back");
    if (spin_bytes == 0) {
        RLS_REQUIRE(C_in == C || (C_in % kWave == 0 && C_in < C && write_back), RLS_EINVAL,
                    "broadcast start state: C_in=%lld must be a multiple of 64 below C (and write_back set)", (long long)C_in);
        RLS_REQUIRE(samples_in != samples || C_in == C, RLS_EINVAL, "a broadcast start state cannot be updated in place");
        RLS_REQUIRE(N < (1 << 30), RLS_EUNSUPPORTED, "N=%lld: the packed walk keeps node ids in 30 bits", (long long)N);
        const size_t tile_b = (size_t)((N + 1) & ~(int64_t)1) * 8, win_b = (size_t)2 * kMetroWin * kWave * 4;
        // the draw windows go to the caller's scratch where that lets a second workgroup share the CU (rls_mcpg_metro_scratch_bytes)
        const int64_t need = rls_mcpg_metro_scratch_bytes(N, C);
        const bool qg = need > 0 && scratch != nullptr && scratch_bytes >= need && knob(KN_METRO_QG, 1) != 0;
        const size_t lds = tile_b + (qg ? 0 : win_b);
        RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld needs %zu B of LDS (max %d)", (long long)N, lds, kLdsBytes);
#define LAUNCH_MP(GV, QQ)                                                                                                      \
    do {                                                                                                                       \
        auto kern = k_mcpg_metro_packed<GV, QQ>;                                                                               \
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);                                                           \
        hipLaunchKernelGGL(kern, cgrid, dim3(kMetroPW * kWave), lds, as_stream(stream), (uint64_t*)samples,                     \
                           (const uint64_t*)samples_in, ceil_div(C_in, kWave), N, C, probs, T, index, u, seed, t_limit_dev,    \
                           write_back, (unsigned long long*)accepts, accept_rows, t_offset, ids, (uint32_t*)scratch);         \
    } while (0)
        if (index) { if (qg) LAUNCH_MP(true, true); else LAUNCH_MP(true, false); }
        else       { if (qg) LAUNCH_MP(false, true); else LAUNCH_MP(false, false); }
#undef LAUNCH_MP
        return check_launch("k_mcpg_metro_packed");
    }
    RLS_REQUIRE(C_in == C, RLS_EINVAL, "a broadcast start state (C_in != C) needs the bit-packed layout");
    const size_t lds_base = (size_t)N * 8 + (accepts ? (size_t)T * 4 : 0) + 16;
    const bool probs_lds = lds_base + (size_t)N * 4 <= (size_t)kLdsBytes;
    const size_t lds = lds_base + (probs_lds ? (size_t)N * 4 : 0);
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld, T=%lld need %zu B of LDS (max %d)", (long long)N,
                (long long)T, lds, kLdsBytes);
    const dim3 grid = cgrid, block(kMetroWaves * kWave);
    hipStream_t s = as_stream(stream);
#define LAUNCH_METRO(TT, PL)                                                                                        \
    do {                                                                                                            \
        auto kern = k_mcpg_metro<TT, PL>;                                                                           \
        if (lds > 64 * 1024)                                                                                        \
            ensure_dyn_lds((const void*)kern, lds);     \
        hipLaunchKernelGGL(kern, grid, block, lds, s, (TT*)samples, (const TT*)samples_in, N, C, probs, T, index, u, seed, t_limit_dev, \
                           write_back, (unsigned long long*)accepts, accept_rows, t_offset, ids);                   \
    } while (0)
    if (spin_bytes == 1) { if (probs_lds) LAUNCH_METRO(uint8_t, true); else LAUNCH_METRO(uint8_t, false); }
    else                 { if (probs_lds) LAUNCH_METRO(float, true);   else LAUNCH_METRO(float, false); }
#undef LAUNCH_METRO
    return check_launch("k_mcpg_metro");
}

// scratch the bit-packed walk wants for its draw windows (0: they stay in LDS): used when the tile + windows leave one workgroup per
// CU but two tiles alone fit -- 32 KB per 64-chain tile of the launch
int64_t rls_mcpg_metro_scratch_bytes(int64_t N, int64_t C) {
    if (N <= 0 || C <= 0) return 0;
    const size_t tile_b = (size_t)((N + 1) & ~(int64_t)1) * 8, win_b = (size_t)2 * kMetroWin * kWave * 4;
    const bool two_with = 2 * (tile_b + win_b) <= (size_t)kLdsBytes, two_without = 2 * tile_b <= (size_t)kLdsBytes;
    const bool fits_with = tile_b + win_b <= (size_t)kLdsBytes;
    if ((two_with || !two_without) && fits_with) return 0;
    if (tile_b > (size_t)kLdsBytes) return 0;
    return ceil_div(C, kWave) * (int64_t)win_b;
}

// rounds one rls_mcpg_metro_rounds launch can take WITH accept counts (they sit in LDS beside the tile in the node-major
// kernels): 0 = the layout does not fit at all for this N
int64_t rls_mcpg_metro_max_rounds(int64_t N, int32_t spin_bytes) {
    if (N <= 0) return 0;
    if (spin_bytes == 0)
        return (size_t)((N + 1) & ~(int64_t)1) * 8 + (size_t)2 * kMetroWin * kWave * 4 <= (size_t)kLdsBytes ? ((int64_t)1 << 40) : 0;
    const int64_t room = (int64_t)kLdsBytes - N * 8 - 16;
    return room >= 4 ? room / 4 : 0;
}

int rls_mcpg_metro_stop(const int64_t* accepts, int64_t accept_rows, int64_t T, int64_t target, int32_t first, int64_t next_T,
                        int64_t* ctl, int64_t* apply_limit, void* stream) {
    RLS_REQUIRE(accepts && ctl, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(accept_rows >= 1 && T >= 1 && next_T >= 0 && target >= 0, RLS_EINVAL, "bad sizes rows=%lld T=%lld next_T=%lld",
                (long long)accept_rows, (long long)T, (long long)next_T);
    hipLaunchKernelGGL(k_metro_stop, dim3(1), dim3(kStopWaves * kWave), 0, as_stream(stream), accepts, accept_rows, T, target, (int)first, next_T,
                       ctl, apply_limit);
    return check_launch("k_metro_stop");
}

static size_t lv_lds_bytes(int64_t N, int64_t num_groups) {
    return (size_t)(N + 2) * 8 + (size_t)kWave * 4 + (((size_t)(num_groups + 2) * 4 + 15) & ~(size_t)15);
}

#ifdef RLS_K7_PROF
int rls_dev_k7_prof(unsigned long long* out) {   // dev builds only: [2048 workgroups][16 waves][total, barrier, header, group, groups, hub] cycles
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k7_prof), sizeof(unsigned long long) * 2048 * 16 * 6);
}
#endif

int rls_mcpg_local_search_levels_supported(const rls_graph* g, int64_t num_groups) {
    if (!g || g->num_nodes <= 0 || num_groups <= 0) return 0;
    if (g->num_nodes >= (1 << 20) || g->max_degree >= 1024 || g->wgt) return 0;
    if (pick_planes(g->num_stored_edges) == 0) return 0;
    return lv_lds_bytes(g->num_nodes, num_groups) <= (size_t)kLdsBytes;
}

int rls_mcpg_local_search_levels(const rls_graph* g, const void* xs_in, int spin_bytes, int64_t C_in, void* xs_out,
                                 int out_spin_bytes, int64_t C, const int32_t* lv_ptr, const int32_t* lv_data,
                                 int64_t num_groups, int64_t num_ls, const uint64_t* coins, uint64_t seed, float* expected,
                                 const rls_chain_ids* chain_ids, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(C >= 0 && num_ls >= 0 && num_groups > 0, RLS_EINVAL, "bad sizes");
    if (C == 0) return RLS_OK;
    ChainIds ids;
    dim3 cgrid;
    if (int rc = chain_ids_arg(chain_ids, C, ids, cgrid)) return rc;
    RLS_REQUIRE((ids.offset & (kWave - 1)) == 0 && (ids.skip & (kWave - 1)) == 0, RLS_EINVAL,
                "chain_ids: this kernel draws per 64-chain tile; offset, period and skip must be multiples of 64");
    RLS_REQUIRE(xs_in && xs_out && lv_ptr && lv_data && expected, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(spin_bytes == 0 || spin_bytes == 1 || spin_bytes == 4, RLS_EINVAL, "spin_bytes must be 0 (bit-packed), 1 or 4");
    RLS_REQUIRE(out_spin_bytes == 0 || out_spin_bytes == 4, RLS_EINVAL, "out_spin_bytes must be 0 (bit-packed) or 4 (float32)");
    if (C_in <= 0) C_in = C;
    RLS_REQUIRE(C_in == C || (spin_bytes == 0 && C_in % kWave == 0 && C_in < C), RLS_EINVAL,
                "broadcast input (C_in != C) needs the bit-packed layout and C_in a multiple of 64");
    RLS_REQUIRE(xs_in != xs_out || (spin_bytes == 0 && out_spin_bytes == 0 && C_in == C), RLS_EINVAL,
                "in-place local search needs the bit-packed layout on both sides");
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    RLS_REQUIRE(N < (1 << 20) && g->max_degree < 1024 && !g->wgt, RLS_EUNSUPPORTED,
                "level-parallel K7 needs an unweighted graph, N < 2^20, degrees < 1024");
    const int P = pick_planes(E);
    RLS_REQUIRE(P != 0, RLS_EUNSUPPORTED, "E=%lld too large", (long long)E);
    const size_t lds = lv_lds_bytes(N, num_groups);
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld needs %zu B of LDS (max %d)", (long long)N, lds,
                kLdsBytes);
    const int64_t tiles_in = ceil_div(C_in, kWave);
    const dim3 grid = cgrid;
    hipStream_t s = as_stream(stream);
    const int force_w = (int)knob(KN_K7_WAVES, 0);   // dev knob
#define LAUNCH_LVL(TI, TO, PP, WW)                                                                              \
    do {                                                                                                        \
        auto kern = k_mcpg_local_search_levels<TI, TO, PP, WW>;                                                 \
        if (lds > 64 * 1024)                                                                                    \
            ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, dim3(WW * kWave), lds, s, (const typename ChainStore<TI>::type*)xs_in,   \
                           (typename ChainStore<TO>::type*)xs_out, N, C, tiles_in, lv_ptr, lv_data, num_groups,  \
                           num_ls, coins, seed, g->eu, g->ev, E, expected, ids);                                 \
    } while (0)
#define DISPATCH_LVL(TI, TO, WW)                      \
    switch (P) {                                      \
        case 12: LAUNCH_LVL(TI, TO, 12, WW); break;   \
        case 16: LAUNCH_LVL(TI, TO, 16, WW); break;   \
        case 20: LAUNCH_LVL(TI, TO, 20, WW); break;   \
        default: LAUNCH_LVL(TI, TO, 24, WW); break;   \
    }
    if (spin_bytes == 0 && out_spin_bytes == 0) {
        const int ww = force_w ? force_w : kLvWavesPacked;
        if (ww == 4) { DISPATCH_LVL(Packed64, Packed64, 4) }
        else if (ww == 16) { DISPATCH_LVL(Packed64, Packed64, 16) }
        else { DISPATCH_LVL(Packed64, Packed64, kLvWavesPacked) }
    } else if (spin_bytes == 0) { DISPATCH_LVL(Packed64, float, kLvWaves) }
    else if (spin_bytes == 1 && out_spin_bytes == 4) { DISPATCH_LVL(uint8_t, float, kLvWaves) }
    else if (spin_bytes == 4 && out_spin_bytes == 4) { DISPATCH_LVL(float, float, kLvWaves) }
    else if (spin_bytes == 1) { DISPATCH_LVL(uint8_t, Packed64, kLvWaves) }
    else { DISPATCH_LVL(float, Packed64, kLvWaves) }
#undef DISPATCH_LVL
#undef LAUNCH_LVL
    return check_launch("k_mcpg_local_search_levels");
}

int rls_mcpg_local_search(const rls_graph* g, const void* xs_in, int spin_bytes, float* xs_out, int64_t C,
                          const int32_t* order, const int32_t* visit_stream, int64_t visit_len, int64_t num_ls,
                          const float* uniforms, uint64_t seed, const int32_t* edge_weights, int64_t gauge_node,
                          float* expected, const rls_chain_ids* chain_ids, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(C >= 0 && num_ls >= 0, RLS_EINVAL, "bad sizes");
    if (C == 0) return RLS_OK;
    ChainIds ids;
    dim3 cgrid;
    if (int rc = chain_ids_arg(chain_ids, C, ids, cgrid)) return rc;
    RLS_REQUIRE(xs_in && xs_out && order && expected, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(spin_bytes == 1 || spin_bytes == 4, RLS_EINVAL, "spin_bytes must be 1 or 4");
    const int64_t N = g->num_nodes, E = g->num_stored_edges;
    RLS_REQUIRE(gauge_node >= -1 && gauge_node < N, RLS_EINVAL, "gauge_node %lld outside [-1, N)", (long long)gauge_node);
    const int P = pick_planes(E);
    RLS_REQUIRE(P != 0, RLS_EUNSUPPORTED, "E=%lld too large", (long long)E);
    const dim3 grid = cgrid, block(kWave);
    hipStream_t s = as_stream(stream);
    const bool weighted = edge_weights != nullptr;
    const size_t lds_fast = (size_t)(N + 2) * 8 + (size_t)kRing * 4 + (size_t)kK7Waves * kWave * 8;
    const int64_t hdr = weighted ? 5 : 4, ent = weighted ? 2 : 1;
    const bool fast = visit_stream != nullptr && ent * g->max_degree + hdr <= kRingMaxRun && lds_fast <= (size_t)kLdsBytes &&
                      (((uintptr_t)visit_stream) & 3) == 0;
    RLS_REQUIRE(fast || (!weighted && gauge_node < 0), RLS_EUNSUPPORTED,
                "weighted / gauge-fixed K7 needs the batched visit stream (max degree <= %d, N * 8 + 24 KB of LDS)",
                (int)((kRingMaxRun - 5) / 2));
    if (fast) {
        // records ent * nnz + hdr * N, one offset per node, 3 header words per batch (1 <= batches <= N)
        RLS_REQUIRE(visit_len >= ent * g->nnz + (hdr + 1) * N + 3 && visit_len <= ent * g->nnz + (hdr + 4) * N, RLS_EINVAL,
                    "visit_len %lld is not a batched visit stream of this graph", (long long)visit_len);
        const dim3 block(kK7Waves * kWave);
        const int gn = (int)gauge_node;
#define LAUNCH_LSS(TI, PP, WGT)                                                                                      \
    do {                                                                                                             \
        auto kern = k_mcpg_local_search_stream<TI, PP, WGT>;                                                         \
        if (lds_fast > 64 * 1024)                                                                                    \
            ensure_dyn_lds((const void*)kern, lds_fast); \
        hipLaunchKernelGGL(kern, grid, block, lds_fast, s, (const TI*)xs_in, xs_out, N, C, visit_stream, visit_len,   \
                           num_ls, uniforms, seed, g->eu, g->ev, edge_weights, E, gn, expected, ids);                 \
    } while (0)
#define DISPATCH_PS(TI)                              \
    if (weighted) { LAUNCH_LSS(TI, 12, true); }      \
    else switch (P) {                                \
        case 12: LAUNCH_LSS(TI, 12, false); break;   \
        case 16: LAUNCH_LSS(TI, 16, false); break;   \
        case 20: LAUNCH_LSS(TI, 20, false); break;   \
        default: LAUNCH_LSS(TI, 24, false); break;   \
    }
        if (spin_bytes == 1) { DISPATCH_PS(uint8_t) } else { DISPATCH_PS(float) }
#undef DISPATCH_PS
#undef LAUNCH_LSS
        return check_launch("k_mcpg_local_search_stream");
    }
    const size_t lds = (size_t)N * 8 + (size_t)((N + 31) / 32) * 4 + 16;
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld needs %zu B of LDS (max %d)", (long long)N, lds,
                kLdsBytes);
#define LAUNCH_LS(TI, PP)                                                                                       \
    do {                                                                                                        \
        auto kern = k_mcpg_local_search<TI, PP>;                                                                \
        if (lds > 64 * 1024)                                                                                    \
            ensure_dyn_lds((const void*)kern, lds); \
        hipLaunchKernelGGL(kern, grid, block, lds, s, (const TI*)xs_in, xs_out, N, C, g->rowptr, g->col, order, \
                           num_ls, uniforms, seed, g->eu, g->ev, E, expected, ids);                             \
    } while (0)
#define DISPATCH_P(TI)                       \
    switch (P) {                             \
        case 12: LAUNCH_LS(TI, 12); break;   \
        case 16: LAUNCH_LS(TI, 16); break;   \
        case 20: LAUNCH_LS(TI, 20); break;   \
        default: LAUNCH_LS(TI, 24); break;   \
    }
    if (spin_bytes == 1) { DISPATCH_P(uint8_t) } else { DISPATCH_P(float) }
#undef DISPATCH_P
#undef LAUNCH_LS
    return check_launch("k_mcpg_local_search");
}

int rls_mcpg_pick_best(const float* expected, const void* xs, int spin_bytes, int64_t N, int64_t total_mcmc_num,
                       int64_t repeat_times, int64_t num_edges, int64_t* best_index, float* vs_good,
                       void* xs_good, void* stream) {
    RLS_REQUIRE(N > 0 && total_mcmc_num >= 0 && repeat_times > 0, RLS_EINVAL, "bad sizes");
    if (total_mcmc_num == 0) return RLS_OK;
    RLS_REQUIRE(expected && xs && best_index && vs_good && xs_good, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(spin_bytes == 0 || spin_bytes == 4, RLS_EINVAL, "spin_bytes must be 0 (bit-packed) or 4 (float32)");
    hipLaunchKernelGGL(k_mcpg_pick_argmin, dim3((unsigned)ceil_div(total_mcmc_num, 64)), dim3(64), 0,
                       as_stream(stream), expected, total_mcmc_num, repeat_times, (float)num_edges, best_index, vs_good);
    if (spin_bytes == 0)
        hipLaunchKernelGGL(k_mcpg_pick_gather_packed, dim3((unsigned)ceil_div(N, 256), (unsigned)ceil_div(total_mcmc_num, kWave)),
                           dim3(256), 0, as_stream(stream), (const uint64_t*)xs, N, total_mcmc_num, best_index, (uint64_t*)xs_good);
    else
        hipLaunchKernelGGL(k_mcpg_pick_gather, dim3(grid_for(N * total_mcmc_num, 256)), dim3(256), 0, as_stream(stream),
                           (const float*)xs, N, total_mcmc_num, total_mcmc_num * repeat_times, best_index, (float*)xs_good);
    return check_launch("k_mcpg_pick_best");
}

int rls_mcpg_merge_best(const float* temp_max, uint64_t* temp_info, float* now_max_res, uint64_t* now_info, int64_t N,
                        int64_t total_mcmc_num, uint64_t* mask_scratch, float* best_value, int64_t* best_index,
                        int32_t replace_worst, void* stream) {
    RLS_REQUIRE(N > 0 && total_mcmc_num > 0, RLS_EINVAL, "bad sizes");
    RLS_REQUIRE(temp_max && temp_info && now_max_res && now_info && mask_scratch, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(replace_worst || (best_value && best_index), RLS_EINVAL, "replace_worst = 0 reports {max, min}: best_value / best_index [2] are needed");
    const int64_t M = total_mcmc_num, tiles = ceil_div(M, kWave);
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_mcpg_merge_mask, dim3((unsigned)ceil_div(tiles * kWave, 256)), dim3(256), 0, s, temp_max, now_max_res, M,
                       mask_scratch);
    hipLaunchKernelGGL(k_mcpg_merge_apply, dim3(grid_for(tiles * N, 256)), dim3(256), 0, s, temp_info, now_info, N, tiles,
                       mask_scratch);
    hipLaunchKernelGGL(k_mcpg_merge_minmax, dim3(1), dim3(1024), 0, s, now_max_res, M, N, now_info, temp_info, best_value,
                       best_index, (int)(replace_worst != 0));
    return check_launch("k_mcpg_merge_best");
}

int rls_mcpg_value_bit_sums(const uint64_t* samples, int64_t N, int64_t C, const float* value, float* A, void* stream) {
    RLS_REQUIRE(N > 0 && C >= 0, RLS_EINVAL, "bad sizes");
    if (C == 0) return RLS_OK;
    RLS_REQUIRE(samples && value && A, RLS_EINVAL, "NULL pointer");
    const int64_t tiles = ceil_div(C, kWave);
    const bool regs = N <= (int64_t)kBitSumRegs * kBitSumThreads;
    const size_t lds = (size_t)(8 * 256 + kWave) * 4 + (regs ? 0 : (size_t)N * 4);
    if (lds <= (size_t)kLdsBytes / 2) {
        auto kern = regs ? k_mcpg_value_bit_sums_lut<true> : k_mcpg_value_bit_sums_lut<false>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        int64_t per_cu = (int64_t)kLdsBytes / (int64_t)lds;           // resident workgroups per CU by LDS ...
        if (per_cu > 2048 / kBitSumThreads) per_cu = 2048 / kBitSumThreads;   // ... and by threads
        int64_t grid = 256 * per_cu;
        if (grid > tiles) grid = tiles;
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kBitSumThreads), lds, as_stream(stream), samples, N, C, tiles,
                           value, A);
        return check_launch("k_mcpg_value_bit_sums_lut");
    }
    hipLaunchKernelGGL(k_mcpg_value_bit_sums, dim3((unsigned)ceil_div(N, 256), (unsigned)ceil_div(C, kWave)), dim3(256), 0,
                       as_stream(stream), samples, N, C, value, A);
    return check_launch("k_mcpg_value_bit_sums");
}

int rls_mcpg_pack_chains(const void* xs, int spin_bytes, int64_t N, int64_t C, uint64_t* packed, void* stream) {
    RLS_REQUIRE(N > 0 && C >= 0, RLS_EINVAL, "bad sizes");
    if (C == 0) return RLS_OK;
    RLS_REQUIRE(xs && packed, RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(spin_bytes == 1 || spin_bytes == 4, RLS_EINVAL, "spin_bytes must be 1 or 4");
    RLS_REQUIRE(N < (1ll << 22), RLS_EUNSUPPORTED, "N=%lld: the pack grid walks 64-node blocks in grid.y", (long long)N);
    const dim3 grid((unsigned)ceil_div(ceil_div(C, kWave), kPackTiles), (unsigned)ceil_div(N, kWave));   // x = run of chain tiles, y = 64-node block
    const int shim = (int)knob(KN_MCPG_SHIM, 1);     // 0: the round-4 kernels (A/B)
    if (spin_bytes == 4 && (shim & 15) && C % 4 == 0 && (reinterpret_cast<uintptr_t>(xs) & 15) == 0 && N < (1ll << 21)) {
        const int spans = 1 << ((shim >> 4) & 15 ? ((shim >> 4) & 15) - 1 : 0);      // A/B: bits 4..7 of the knob = 1 + log2 spans; 1 span measured best (2020 us; 4 spans 2105, the round-4 kernel 2390)
        const dim3 g2((unsigned)ceil_div(ceil_div(C, 256), spans), (unsigned)ceil_div(N, 64));   // x = run of 256-chain spans, y = 64 rows
        hipLaunchKernelGGL(k_mcpg_pack_f32x4, g2, dim3(4 * kWave), 0, as_stream(stream), (const float*)xs, N, C, packed, spans);
        return check_launch("k_mcpg_pack_f32x4");
    }
    if (spin_bytes == 1) hipLaunchKernelGGL(k_mcpg_pack<uint8_t>, grid, dim3(4 * kWave), 0, as_stream(stream), (const uint8_t*)xs, N, C, packed);
    else hipLaunchKernelGGL(k_mcpg_pack<float>, grid, dim3(4 * kWave), 0, as_stream(stream), (const float*)xs, N, C, packed);
    return check_launch("k_mcpg_pack");
}

int rls_mcpg_unpack_chains(const uint64_t* packed, int64_t N, int64_t C, float* xs, void* stream) {
    RLS_REQUIRE(N > 0 && C >= 0, RLS_EINVAL, "bad sizes");
    if (C == 0) return RLS_OK;
    RLS_REQUIRE(xs && packed, RLS_EINVAL, "NULL pointer");
    if (C % 4 == 0 && (reinterpret_cast<uintptr_t>(xs) & 15) == 0) {
        const dim3 grid((unsigned)ceil_div(C / 4, 256), (unsigned)(N < 32768 ? N : 32768));
        // two consecutive rows per thread (their words are adjacent) and non-temporal stores -- the [N, C] result is far larger than
        // the Infinity Cache and nobody on the chip reads it back: BA-1e4 / 2^18 chains 2022 -> 1737 us (5.19 -> 6.04 TB/s)
        if (knob(KN_MCPG_SHIM, 1) != 0)
            hipLaunchKernelGGL((k_mcpg_unpack<true, true, 2>), dim3(grid.x, (unsigned)ceil_div((int64_t)grid.y, 2)), dim3(256), 0, as_stream(stream),
                               packed, N, C, xs);
        else
        hipLaunchKernelGGL(k_mcpg_unpack<true>, grid, dim3(256), 0, as_stream(stream), packed, N, C, xs);
    } else {
        hipLaunchKernelGGL(k_mcpg_unpack<false>, dim3(grid_for(N * C, 256)), dim3(256), 0, as_stream(stream), packed, N, C, xs);
    }
    return check_launch("k_mcpg_unpack");
}

}  // extern "C"
