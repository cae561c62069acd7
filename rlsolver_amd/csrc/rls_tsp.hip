// TSP kernels (K12, K13, K14-perm) for gfx950.
//
// perm[B, N] int64 is the reference's `sample` (env-major); the distance matrix D[N, N] f32 is shared
// by every env and staged in LDS once per workgroup when it fits (N <= 160: 100 KB; TSP-100 = 40 KB),
// otherwise read through L2.  One wave owns one tour at a time: lanes run along the tour positions,
// so perm reads and the [B, N] outputs are coalesced.
#include "rls_tile.h"
#include "rls_ring.h"
#include "rls_draw.h"

namespace rls {

constexpr int kTspBlock = 1024;     // launch bound; small tours use it whole (32 waves/CU hide the HBM round trip)
constexpr int kTspBlockSmall = 256;  // large N: the per-wave LDS tour scratch limits waves per workgroup

__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

template <bool LDS_D>
__device__ __forceinline__ const float* stage_dist(const float* __restrict__ dist, int64_t N, float* lds) {
    if constexpr (LDS_D) {
        // LDS-DMA, 256 B per wave instruction, every chunk in flight at once (a load->ds_write loop
        // serialises on the L2 round trip and dominated the kernel)
        const int64_t n2 = N * N;
        const int lane = threadIdx.x & 63;
        for (int64_t c = (threadIdx.x >> 6) * (int64_t)kWave; c < n2; c += blockDim.x)
            if (c + lane < n2) glds4(dist + c + lane, lds + c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        return lds;
    } else {
        return dist;
    }
}

// K12: length[b] = sum_k D[p[k], p[k+1]] + D[p[N-1], p[0]]
template <bool LDS_D>
__global__ __launch_bounds__(kTspBlock) void k_tsp_tour_length(const float* __restrict__ dist, int64_t N,
                                                               const int64_t* __restrict__ perm, int64_t B,
                                                               float* __restrict__ length) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const float* D = stage_dist<LDS_D>(dist, N, reinterpret_cast<float*>(smem));
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    constexpr int TPT = 4;   // tours per trip: their (independent) perm loads are all in flight together --
                             // one tour at a time leaves each wave waiting a full HBM round trip per tour
    for (int64_t b = wave * TPT; b < B; b += nwaves * TPT) {
        float acc[TPT];
#pragma unroll
        for (int t = 0; t < TPT; ++t) acc[t] = 0.0f;
        for (int64_t k = lane; k < N; k += kWave) {
            const int64_t k1 = (k + 1 == N) ? 0 : k + 1;
            int64_t a[TPT], c[TPT];
#pragma unroll
            for (int t = 0; t < TPT; ++t) {
                const int64_t bb = (b + t < B) ? b + t : b;
                a[t] = perm[bb * N + k];
                c[t] = perm[bb * N + k1];
            }
#pragma unroll
            for (int t = 0; t < TPT; ++t) acc[t] += D[a[t] * N + c[t]];
        }
#pragma unroll
        for (int t = 0; t < TPT; ++t) {
            const float tot = wave_sum_f32(acc[t]);
            if (lane == t && b + t < B) length[b + t] = tot;
        }
    }
}

// K13: ISCO_TSP.opt_2 (env_ISCO.py:238-335) for every position of every tour.  (What bounds it, round 6: not HBM -- each position
// makes 12 RANDOM LDS gathers (8 matrix entries, 2 tour cities, the inverse, the table entry), and 64 random addresses over 32 banks
// serialise ~4-5 deep: ~31 us of LDS time at TSP-100 / 2^16 whatever the bytes moved.  A "flat" form (T tours per wave pass as one
// run of T * N positions, 98 % of the lane slots busy instead of 78 %, byte-sized tours) was built and measured: 42.4 / 38.6 us vs
// 41.3 / 34.2 for this one -- the same gathers in fewer, fuller passes conflict more; not kept.)  The partner city is drawn IN the
// kernel (DRAW:
// rand < K / (K + 1) picks one of the K nearest, else one of the N - K - 1 others; the generator and the counters of the fused
// ISCO step's iteration 0, so the [B, N] int64 `selected` tensor of a two-op opt_2 -- 8N of the 29N bytes per tour -- never
// exists), or given (selected != NULL: the recorded-draw hook of the golden tests); then the position of the partner, the ban
// mask and the swap delta.
struct TspDraw {
    const int32_t* nearest; const int32_t* random; int32_t K; int32_t random_stride; float near_threshold;
    uint64_t seed; int64_t env_offset; int64_t* selected_out;
    const uint8_t* tables8; int32_t tables8_bytes;      // both tables as bytes, [N, K] (padded to 16 B) then [N, N - K - 1], or NULL
};

// TAB8: the two neighbour tables in LDS as bytes (city ids < 256: TSP-100 = 2 + 7.9 KB beside the 40 KB matrix; the caller's
// tables8 block, LDS-DMA'd with the matrix) -- from global memory the table entry is a dependent load on every position's path
template <bool LDS_D, bool DRAW, bool TAB8 = false>
__global__ __launch_bounds__(kTspBlock) void k_tsp_swap_delta_all(const float* __restrict__ dist, int64_t N,
                                                                  const int64_t* __restrict__ perm, int64_t B,
                                                                  const int64_t* __restrict__ selected, TspDraw dr, float temperature,
                                                                  float* __restrict__ logratio,
                                                                  int64_t* __restrict__ indices,
                                                                  uint8_t* __restrict__ ban) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* dl = reinterpret_cast<float*>(smem);
    // per-wave scratch after the (optional) matrix: tour and its inverse, int32 each
    int32_t* scratch = reinterpret_cast<int32_t*>(smem + (LDS_D ? (size_t)N * N * 4 : 0));
    const int wib = threadIdx.x / kWave;
    int32_t* P = scratch + (int64_t)wib * 2 * N;
    int32_t* INV = P + N;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / kWave) + wib;
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x / kWave);
    const int n = (int)N;
    const int K = dr.K, NR = n - K - 1;
    uint8_t* near8 = reinterpret_cast<uint8_t*>(scratch + (int64_t)(blockDim.x / kWave) * 2 * N);      // [N, K]
    uint8_t* rand8 = near8 + (((size_t)n * K + 15) & ~(size_t)15);                                      // [N, NR]
    if constexpr (DRAW && TAB8) {
        // the byte tables travel with the matrix: LDS-DMA, 256 B per wave instruction, published by stage_dist's barrier
        const int ndw = dr.tables8_bytes >> 2;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(dr.tables8);
        uint32_t* dst = reinterpret_cast<uint32_t*>(near8);
        for (int c = (threadIdx.x >> 6) * kWave; c < ndw; c += blockDim.x)
            if (c + lane < ndw) glds4(src + c + lane, dst + c);
        if constexpr (!LDS_D) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const float* D = stage_dist<LDS_D>(dist, N, dl);
    for (int64_t b = wave; b < B; b += nwaves) {
        const int64_t* p = perm + b * N;
        for (int k = lane; k < n; k += kWave) {
            const int city = (int)p[k];
            P[k] = city;
            INV[city] = k;     // sort + searchsorted of the reference == inverse permutation
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (a wave owns the tour: its global id and the env key of the generator are scalars)
        const uint64_t genv = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)(b + dr.env_offset) >> 32)) << 32) |
                              (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(b + dr.env_offset));
        const uint32_t ekey = DRAW ? isco_env_key(dr.seed, genv) : 0u;
        for (int i = lane; i < n; i += kWave) {
            int sel;
            if constexpr (DRAW) {
                // env_ISCO.py:246-266 with the draws of rls_isco_tsp_step's iteration 0: streams 3 (coin), 4 (near pick), 5 (far
                // pick) -- the pick's stream chosen per lane, so a position costs one position key and two draws
                const int city = P[i];
                const uint32_t pkey = isco_pos_key(ekey, genv, (uint32_t)i);
                const bool near = isco_unit(isco_draw_at(pkey, 0u, 3)) < dr.near_threshold;
                const uint32_t pick = isco_draw_at(pkey, 0u, near ? 4u : 5u);
                const int r = (int)__umulhi(pick, (uint32_t)(near ? K : NR));
                if (near) sel = TAB8 ? (int)near8[__umul24(city, K) + r] : dr.nearest[(int64_t)city * K + r];
                else sel = TAB8 ? (int)rand8[__umul24(city, NR) + r] : dr.random[(int64_t)city * dr.random_stride + r];
                if (dr.selected_out) dr.selected_out[b * N + i] = sel;
            } else {
                sel = (int)selected[b * N + i];
            }
            const int j = INV[sel];
            const int i0 = (i == 0) ? n - 1 : i - 1;
            const int i1 = (i + 1 == n) ? 0 : i + 1;
            const int i2 = (i1 + 1 == n) ? 0 : i1 + 1;
            const int j0 = (j == 0) ? n - 1 : j - 1;
            const int j1 = (j + 1 == n) ? 0 : j + 1;
            const int s_m1 = P[i1], s_m0 = P[i0];
            const bool banned = (s_m1 == sel) || (s_m0 == sel);           // env_ISCO.py:291-293
            const int s_i0 = P[j0], s_i1 = P[j1], s_i = P[j];
            const bool c3 = (s_m1 == s_i0);                                // partner sits at position i+2
            const int nm = P[i], nm1 = s_m1, nm2 = P[i2];
            auto DD = [&](int a, int c) { return D[__umul24(a, n) + c]; };  // (N < 2^24: a full-rate multiply)
            float delta;
            if (banned) {
                delta = 0.0f;
            } else if (c3) {                                               // env_ISCO.py:320-324
                delta = -(DD(nm, nm1) + DD(s_i, s_i1)) + (DD(nm, s_i) + DD(s_i0, s_i1));
            } else {                                                       // env_ISCO.py:326-332
                delta = -(((DD(nm, nm1) + DD(nm1, nm2)) + DD(s_i0, s_i)) + DD(s_i, s_i1)) +
                        (((DD(nm, s_i) + DD(s_i, nm2)) + DD(s_i0, nm1)) + DD(nm1, s_i1));
            }
            logratio[b * N + i] = (-delta) / temperature;                  // -delta_yx / temperature
            indices[b * N + i] = j;
            ban[b * N + i] = banned ? 1 : 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

__global__ void k_tsp_apply_swap(int64_t* __restrict__ perm, int64_t B, int64_t N, const int64_t* __restrict__ pos,
                                 const int64_t* __restrict__ indices) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t q = pos[b];
    if (q < 0) return;
    const int64_t j = indices[b * N + q];
    const int64_t a = (q + 1) % N;                                          // env_ISCO.py:340
    int64_t* p = perm + b * N;
    const int64_t t = p[a];
    p[a] = p[j];
    p[j] = t;
}

__global__ void k_tsp_2opt_delta(const float* __restrict__ dist, int64_t N, const int64_t* __restrict__ perm,
                                 int64_t B, const int64_t* __restrict__ ii, const int64_t* __restrict__ jj,
                                 float* __restrict__ delta) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int64_t i = ii[b], j = jj[b];
    const int64_t* p = perm + b * N;
    const int64_t im = (i == 0) ? N - 1 : i - 1, jp = (j + 1 == N) ? 0 : j + 1;
    if (jp == i) { delta[b] = 0.0f; return; }                              // whole tour reversed
    const int64_t a = p[im], c = p[i], d = p[j], e = p[jp];
    delta[b] = (dist[a * N + d] + dist[c * N + e]) - (dist[a * N + c] + dist[d * N + e]);
}

// Fisher-Yates with counter-based murmur draws (round 6; csrc/rls_draw.h, stream 7): p = identity; for k = N-1..1:
// j = (draw(seed, global env id, k) * (k+1)) >> 32; swap(p[k], p[j]).  (A Philox call per swap -- ten rounds of two 32 x 32 -> 64
// multiplies -- was the kernel's whole time: 0.21 of the write stream; four swaps per call 0.30.)
__device__ __forceinline__ uint32_t perm_draw(uint32_t ekey, uint64_t gb, uint32_t k) { return isco_draw_at(isco_pos_key(ekey, gb, k), 0u, 7u); }

__global__ void k_rand_perms(int64_t* __restrict__ perm, int64_t B, int64_t N, uint64_t seed, int64_t env_offset) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int64_t* p = perm + b * N;
    for (int64_t k = 0; k < N; ++k) p[k] = k;
    const uint64_t gb = (uint64_t)(b + env_offset);
    const uint32_t ekey = isco_env_key(seed, gb);
    for (int64_t k = N - 1; k >= 1; --k) {
        const int64_t j = (int64_t)__umulhi(perm_draw(ekey, gb, (uint32_t)k), (uint32_t)(k + 1));
        const int64_t t = p[k];
        p[k] = p[j];
        p[j] = t;
    }
}

// Same shuffle with the 64 tours of a wave held in LDS as uint16 (lane = env; position k of the 64 envs sits in 66
// halfwords = 33 dwords, so both the per-lane random accesses of the shuffle and the row-wise write-out, where lanes walk
// k, spread over the banks).  The int64 rows then leave as contiguous 512-byte stores -- the in-place global shuffle
// above moved 13x the bytes of its result (every swap a read-modify-write of an 800-byte-strided line).
constexpr int kPermStride = 66;
__global__ __launch_bounds__(kWave) void k_rand_perms_lds(int64_t* __restrict__ perm, int64_t B, int64_t N, uint64_t seed,
                                                           int64_t env_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* p = reinterpret_cast<uint16_t*>(smem);
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * kWave, b = b0 + lane;
    for (int64_t k = 0; k < N; ++k) p[k * kPermStride + lane] = (uint16_t)k;
    const uint64_t gb = (uint64_t)(b + env_offset);
    const uint32_t ekey = isco_env_key(seed, gb);
    // the draw of swap k - 1 is computed while swap k's LDS round trip is in flight (it depends on k only)
    uint32_t r = perm_draw(ekey, gb, (uint32_t)(N - 1));
    for (int64_t k = N - 1; k >= 1; --k) {
        const int64_t j = (int64_t)__umulhi(r, (uint32_t)(k + 1));
        const uint16_t t = p[k * kPermStride + lane];
        const uint16_t u = p[j * kPermStride + lane];
        if (k > 1) r = perm_draw(ekey, gb, (uint32_t)(k - 1));
        p[k * kPermStride + lane] = u;
        p[j * kPermStride + lane] = t;
    }
    __syncthreads();
    const int64_t nb = (B - b0 < kWave) ? B - b0 : kWave;
    if ((N & 1) == 0 && (reinterpret_cast<uintptr_t>(perm) & 15) == 0) {
        // the nb rows are one contiguous run of nb * N int64: 16-byte pieces (two positions of one tour) c = 64 trip + lane of that
        // run, 1 KB per wave store, nontemporal
        typedef int64_t i64x2 __attribute__((ext_vector_type(2)));
        i64x2* run = reinterpret_cast<i64x2*>(perm + b0 * N);
        const int half = (int)(N >> 1), total = (int)nb * half;
        const int q = kWave / half, rem = kWave - q * half;
        int e = lane / half, h = lane - e * half;
        for (int c = lane; c < total; c += kWave) {
            const int64_t k = 2 * h;
            __builtin_nontemporal_store(i64x2{(int64_t)p[k * kPermStride + e], (int64_t)p[(k + 1) * kPermStride + e]}, run + c);
            e += q; h += rem;
            if (h >= half) { h -= half; ++e; }
        }
        return;
    }
    for (int64_t e = 0; e < nb; ++e) {
        int64_t* row = perm + (b0 + e) * N;
        for (int64_t k = lane; k < N; k += kWave) row[k] = (int64_t)p[k * kPermStride + e];
    }
}

// One best-improvement pass of methods_problem_specific/TSP/opt_2.py:27-57: every reversal [i..j], 0 <= i < j <= N - 1, of
// the closed tour is evaluated against the SAME seed tour and the best one wins (the first in (i, j) order among equals:
// the reference keeps a candidate only when it is strictly shorter than the best so far).  A workgroup per tour.
//   EXACT: the reference's own comparison values -- the candidate's whole length as distance_calc sums it (float64,
//          edge after edge from the candidate's first city).  The edges before position i - 1 are the seed's, so the sum
//          resumes from the seed's running sum S[i - 1] (the same additions in the same order) and walks the remaining
//          N - i + 1 edges: candidates that tie in exact arithmetic (a reversal that only turns the cycle around) are
//          ranked by their rounding, exactly as the reference ranks them.  value = that length, compared with cur[b].
//   else:  delta(i, j) = D[a,c] + D[b,e] - D[a,b] - D[c,e] (a = t[i-1], b = t[i], c = t[j], e = t[j+1], cyclic; 0 for the
//          whole tour), O(1) per candidate, for a SYMMETRIC matrix.  value = delta, compared with 0.
// The N(N-1)/2 candidates are one list c = 0.. in (i, j) order, dealt round-robin over the threads of the tour's `slices`
// workgroups (a lone tour -- the reference's use -- still spreads over the chip; rows of different cost interleave).  With
// slices > 1 every workgroup leaves its best (value, key) in row blockIdx.y of the [slices, B] outputs and
// k_tsp_2opt_reduce folds the rows into row 0.
template <bool EXACT>
__global__ __launch_bounds__(256) void k_tsp_2opt_best(const double* __restrict__ dist, int64_t N, const int64_t* __restrict__ perm,
                                                        int64_t B, const double* __restrict__ cur, int64_t* __restrict__ best_i,
                                                        int64_t* __restrict__ best_j, double* __restrict__ best_value) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* rd = reinterpret_cast<double*>(smem);                           // [256]
    int64_t* rk = reinterpret_cast<int64_t*>(rd + 256);                     // [256]
    double* S = reinterpret_cast<double*>(rk + 256);                        // [N + 1] running sums of the seed (EXACT)
    int32_t* t = reinterpret_cast<int32_t*>(S + (EXACT ? N + 1 : 0));       // [N]
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    const int64_t slices = gridDim.y, slice = blockIdx.y;
    for (int64_t k = tid; k < N; k += 256) t[k] = (int32_t)perm[b * N + k];
    __syncthreads();
    if constexpr (EXACT) {                                                  // edges in parallel, then the one sequential sum
        for (int64_t k = tid; k < N; k += 256) S[k + 1] = dist[(int64_t)t[k] * N + t[k + 1 == N ? 0 : k + 1]];
        __syncthreads();
        if (tid == 0) {
            double acc = 0.0;
            S[0] = 0.0;
            for (int64_t k = 0; k < N; ++k) {
                acc = acc + S[k + 1];
                S[k + 1] = acc;
            }
        }
        __syncthreads();
    }
    double best = EXACT ? cur[b] : 0.0;                                     // only strictly better candidates count
    int64_t key = -1;                                                       // i * N + j of the best so far
    const int64_t M = N * (N - 1) / 2;
    const double tn1 = (double)(2 * N - 1);
    for (int64_t c = slice * 256 + tid; c < M; c += slices * 256) {
        // row i holds candidates off(i) .. off(i + 1) - 1, off(i) = i (2N - i - 1) / 2
        int64_t i = (int64_t)((tn1 - sqrt(tn1 * tn1 - 8.0 * (double)c)) * 0.5);
        if (i < 0) i = 0;
        if (i > N - 2) i = N - 2;
        while (i * (2 * N - i - 1) / 2 > c) --i;
        while ((i + 1) * (2 * N - i - 2) / 2 <= c) ++i;
        const int64_t j = i + 1 + (c - i * (2 * N - i - 1) / 2);
        double v;
        if constexpr (EXACT) {
            auto city = [&](int64_t k) -> int64_t {                         // the candidate tour, closed
                if (k == N) k = 0;
                return (k >= i && k <= j) ? t[i + j - k] : t[k];
            };
            const int64_t k0 = i >= 1 ? i - 1 : 0;
            v = S[k0];
            int64_t c0 = city(k0);
            for (int64_t k = k0; k < N; ++k) {
                const int64_t c1 = city(k + 1);
                v = v + dist[c0 * N + c1];
                c0 = c1;
            }
        } else {
            if (i == 0 && j == N - 1) continue;                             // the whole tour reversed: the same cycle
            const int64_t a = t[i == 0 ? N - 1 : i - 1], bb = t[i], cc = t[j], e = t[j + 1 == N ? 0 : j + 1];
            v = (dist[a * N + cc] + dist[bb * N + e]) - (dist[a * N + bb] + dist[cc * N + e]);
        }
        const int64_t kk = i * N + j;
        if (v < best || (v == best && key >= 0 && kk < key)) { best = v; key = kk; }   // the first of equals in (i, j) order
    }
    rd[tid] = best;
    rk[tid] = key;
    __syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (tid < sft) {
            const double o = rd[tid + sft];
            const int64_t ok = rk[tid + sft];
            if (ok >= 0 && (rk[tid] < 0 || o < rd[tid] || (o == rd[tid] && ok < rk[tid]))) { rd[tid] = o; rk[tid] = ok; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int64_t o = slice * B + b;
        best_value[o] = rk[0] >= 0 ? rd[0] : (EXACT ? cur[b] : 0.0);
        if (slices == 1) {
            best_i[o] = rk[0] >= 0 ? rk[0] / N : -1;
            best_j[o] = rk[0] >= 0 ? rk[0] % N : -1;
        } else {
            best_i[o] = rk[0];                                               // the key; k_tsp_2opt_reduce splits the winner's
        }
    }
}

__global__ void k_tsp_2opt_reduce(int64_t N, int64_t B, int64_t slices, int64_t* __restrict__ best_i, int64_t* __restrict__ best_j,
                                  double* __restrict__ best_value) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double v = best_value[b];
    int64_t key = best_i[b];
    for (int64_t s = 1; s < slices; ++s) {
        const double o = best_value[s * B + b];
        const int64_t ok = best_i[s * B + b];
        if (ok >= 0 && (key < 0 || o < v || (o == v && ok < key))) { v = o; key = ok; }
    }
    best_value[b] = v;
    best_i[b] = key >= 0 ? key / N : -1;
    best_j[b] = key >= 0 ? key % N : -1;
}

static inline bool dist_fits_lds(int64_t N, size_t extra) { return (size_t)N * N * 4 + extra <= (size_t)kLdsBytes - 1024; }

static inline int tsp_block(int64_t N) { return N <= 256 ? kTspBlock : kTspBlockSmall; }

static inline int tsp_grid(int64_t B, int block) {
    const int64_t waves = block / kWave;
    const int64_t cap = (int64_t)2 * 256 * kTspBlock / block;   // 32 waves per CU
    int64_t g = ceil_div(B, waves);
    if (g > cap) g = cap;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace rls

using namespace rls;

extern "C" {

int rls_tsp_tour_length(const float* dist, int64_t N, const int64_t* perm, int64_t B, float* length, void* stream) {
    RLS_REQUIRE(N > 0 && N < (1 << 30) && B >= 0, RLS_EINVAL, "bad sizes N=%lld B=%lld", (long long)N, (long long)B);
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(dist && perm && length, RLS_EINVAL, "NULL pointer");
    const dim3 grid(tsp_grid(B, tsp_block(N))), block(tsp_block(N));
    if (dist_fits_lds(N, 0)) {
        const size_t lds = (size_t)N * N * 4;
        auto kern = k_tsp_tour_length<true>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), dist, N, perm, B, length);
    } else {
        hipLaunchKernelGGL(k_tsp_tour_length<false>, grid, block, 0, as_stream(stream), dist, N, perm, B, length);
    }
    return check_launch("k_tsp_tour_length");
}

int64_t rls_tsp_tables8_bytes(int64_t N, int32_t K) {
    if (N < 3 || N > 256 || K < 1 || K >= N) return 0;
    return (int64_t)((((size_t)N * K + 15) & ~(size_t)15) + (((size_t)N * (N - K - 1) + 15) & ~(size_t)15));
}

int rls_tsp_swap_delta_all(const float* dist, int64_t N, const int64_t* perm, int64_t B, const int64_t* selected,
                           const int32_t* nearest, int32_t K, const int32_t* random, int32_t random_stride, const uint8_t* tables8,
                           float near_threshold, uint64_t seed, int64_t env_offset, int64_t* selected_out, float temperature,
                           float* logratio, int64_t* indices, uint8_t* ban, void* stream) {
    RLS_REQUIRE(N > 2 && N < (1 << 24) && B >= 0, RLS_EINVAL, "bad sizes N=%lld B=%lld", (long long)N, (long long)B);
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(dist && perm && logratio && indices && ban, RLS_EINVAL, "NULL pointer");
    const bool draw = selected == nullptr;
    if (draw) {
        RLS_REQUIRE(nearest && random, RLS_EINVAL, "selected == NULL draws the partners in the kernel: nearest / random must be given");
        RLS_REQUIRE(K >= 1 && K < N && random_stride >= N - K - 1, RLS_EINVAL, "bad neighbour tables K=%d stride=%d N=%lld", K,
                    random_stride, (long long)N);
    } else {
        RLS_REQUIRE(!selected_out, RLS_EINVAL, "selected_out is the in-kernel draw's record: it needs selected == NULL");
    }
    const size_t tabs = (draw && tables8 && N <= 256) ? (size_t)rls_tsp_tables8_bytes(N, K) : 0;
    RLS_REQUIRE(!tables8 || (((uintptr_t)tables8) & 3) == 0, RLS_EINVAL, "tables8 must be 4-byte aligned");
    const TspDraw dr{nearest, random, K, random_stride, near_threshold, seed, env_offset, selected_out, tables8, (int32_t)tabs};
    const size_t scratch = (size_t)(tsp_block(N) / kWave) * 2 * N * 4;
    RLS_REQUIRE(scratch <= (size_t)kLdsBytes - 1024, RLS_EUNSUPPORTED, "N=%lld too large for the per-wave tour scratch",
                (long long)N);
    const dim3 grid(tsp_grid(B, tsp_block(N))), block(tsp_block(N));
    // the neighbour tables as bytes in LDS when the ids fit a byte and they fit beside the matrix and the scratch
    const bool tab8 = tabs > 0 && dist_fits_lds(N, scratch + tabs);
    const bool in_lds = dist_fits_lds(N, scratch + (tab8 ? tabs : 0));
    const size_t lds = (in_lds ? (size_t)N * N * 4 : 0) + scratch + (tab8 ? tabs : 0);
#define LAUNCH_K13(LD, DRW)                                                                                              \
    do {                                                                                                                 \
        auto kern = (DRW && tab8) ? k_tsp_swap_delta_all<LD, DRW, true> : k_tsp_swap_delta_all<LD, DRW, false>;          \
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);                                                     \
        hipLaunchKernelGGL(kern, grid, block, lds, as_stream(stream), dist, N, perm, B, selected, dr, temperature, logratio, \
                           indices, ban);                                                                                \
    } while (0)
    if (in_lds) { if (draw) LAUNCH_K13(true, true); else LAUNCH_K13(true, false); }
    else { if (draw) LAUNCH_K13(false, true); else LAUNCH_K13(false, false); }
#undef LAUNCH_K13
    return check_launch("k_tsp_swap_delta_all");
}

int rls_tsp_apply_swap(int64_t* perm, int64_t B, int64_t N, const int64_t* pos, const int64_t* indices, void* stream) {
    RLS_REQUIRE(N > 0 && B >= 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(perm && pos && indices, RLS_EINVAL, "NULL pointer");
    hipLaunchKernelGGL(k_tsp_apply_swap, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, as_stream(stream), perm, B, N,
                       pos, indices);
    return check_launch("k_tsp_apply_swap");
}

int rls_tsp_2opt_delta(const float* dist, int64_t N, const int64_t* perm, int64_t B, const int64_t* i,
                       const int64_t* j, float* delta, void* stream) {
    RLS_REQUIRE(N > 0 && B >= 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(dist && perm && i && j && delta, RLS_EINVAL, "NULL pointer");
    hipLaunchKernelGGL(k_tsp_2opt_delta, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, as_stream(stream), dist, N,
                       perm, B, i, j, delta);
    return check_launch("k_tsp_2opt_delta");
}

int rls_tsp_2opt_best(const double* dist, int64_t N, const int64_t* perm, int64_t B, const double* cur_length, int32_t slices,
                      int64_t* best_i, int64_t* best_j, double* best_value, void* stream) {
    RLS_REQUIRE(N > 2 && N < (1ll << 26) && B >= 0, RLS_EINVAL, "bad sizes N=%lld B=%lld", (long long)N, (long long)B);
    RLS_REQUIRE(slices >= 1 && slices <= 65535, RLS_EINVAL, "slices=%d outside [1, 65535]", slices);
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(dist && perm && best_i && best_j && best_value, RLS_EINVAL, "NULL pointer");
    const size_t lds = 256 * 16 + (cur_length ? (size_t)(N + 1) * 8 : 0) + (size_t)N * 4;
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N=%lld needs %zu B of LDS (max %d)", (long long)N, lds, kLdsBytes);
    if (cur_length) {
        auto kern = k_tsp_2opt_best<true>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)B, (unsigned)slices), dim3(256), lds, as_stream(stream), dist, N, perm, B, cur_length, best_i, best_j, best_value);
    } else {
        auto kern = k_tsp_2opt_best<false>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)B, (unsigned)slices), dim3(256), lds, as_stream(stream), dist, N, perm, B, cur_length, best_i, best_j, best_value);
    }
    if (slices > 1)
        hipLaunchKernelGGL(k_tsp_2opt_reduce, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, as_stream(stream), N, B, (int64_t)slices, best_i,
                           best_j, best_value);
    return check_launch("k_tsp_2opt_best");
}

int rls_rand_perms(int64_t* perm, int64_t B, int64_t N, uint64_t seed, int64_t env_offset, void* stream) {
    RLS_REQUIRE(N > 0 && B >= 0, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(perm, RLS_EINVAL, "perm is NULL");
    const size_t lds = (size_t)N * kPermStride * sizeof(uint16_t);
    if (N <= 65535 && lds <= (size_t)kLdsBytes / 2) {
        auto kern = k_rand_perms_lds;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(B, kWave)), dim3(kWave), lds, as_stream(stream), perm, B, N, seed,
                           env_offset);
        return check_launch("k_rand_perms_lds");
    }
    hipLaunchKernelGGL(k_rand_perms, dim3((unsigned)ceil_div(B, 256)), dim3(256), 0, as_stream(stream), perm, B, N,
                       seed, env_offset);
    return check_launch("k_rand_perms");
}

}  // extern "C"
