// S2V / ECO / PECO spin-system env on ONE shared signed-weight graph:
//   batched f32   rlsolver/methods/ECO_S2V/src/envs/spinsystem_PECO.py:306-486 (and its shared-graph
//                 inference twin inference_network_env.py:295-444), visited-state memory util_envs_PECO.py:228-288
//   single f64    rlsolver/methods/ECO_S2V/src/envs/spinsystem.py:333-482 (numpy), HistoryBuffer util_envs.py:355-381
// Both are the same kernel, templated on the float type of the observable rows.
//
// The reference recomputes all single-flip gains with a dense [B,N,N] matmul every step and re-derives
// every per-env statistic from scratch.  Here everything that a flip changes in O(deg) is kept as resident
// per-env state and updated in O(deg):
//   delta[b,i] = s_i * sum_j W_ij s_j   ("immediate cuts available", int32) -- flipping a negates delta[a] and
//                                        moves delta[j] of each neighbour by 2 W_ja s_j s_a' (S2V_PPO/env.py:197-206)
//   num_nonpos[b] = #{i : delta_i <= 0}  (greedy-actions observable, basin test)
//   dist_best[b]  = Hamming distance to best_spins
//   packed[b,:], hash[b]                 bit-packed spins + their Zobrist hash (visited-state memory)
// and only the entries of the IMMEDIATE_REWARD_AVAILABLE row that changed are rewritten.  The rows the observation
// contract changes EVERYWHERE every step are not stored at all between observations:
//   time-since-flip / episode time   the reference adds fl(1/max_steps) to the row every step and zeroes the flipped entry:
//                                     after k such additions an entry holds time_table[k] (the same float additions, done
//                                     once on the host), k = step - last_flip[b, n]  -- last_flip int32 [B, N], one store per step
//   termination, greedy count, distance from best score / state      four per-env scalars (env.scalars)
// rls_spin_observation materialises them straight into the observation it has to write anyway; rls_spin_materialize
// writes them into `state` for whoever reads that tensor.  A step is O(deg): it was 6 * sizeof(T) * N bytes per env
// (250 us for 2^14 envs of a G22-sized graph, 38 us of which were the O(deg) chain).
//
// One wave per env.
#include "rls_tile.h"
#include <cmath>

namespace rls {

struct SpinRows {  // row index of each observable inside state[b], or -1 when absent
    int immediate, time_since_flip, episode_time, termination, greedy, dist_score, dist_state;
};

typedef double f64x2 __attribute__((ext_vector_type(2)));
template <typename T> struct RowVec;
template <> struct RowVec<float> { using type = f32x4; static constexpr int n = 4; };
template <> struct RowVec<double> { using type = f64x2; static constexpr int n = 2; };

// splitmix64 finaliser: the Zobrist key of a node
__device__ __forceinline__ uint64_t spin_zobrist(uint32_t node) {
    uint64_t z = ((uint64_t)node + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ uint64_t wave_xor_u64(uint64_t v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v ^= __shfl_xor(v, m, 64);
    return v;
}

struct SpinStepArgs {
    double max_local, termination_value, reward_div, stag_punishment, basin_reward;
    int reward_mode, use_stag, use_basin;
    int64_t hist_len;
};

// DENSE: per-env couplings matrix T [B, N, N] (the training envs: a fresh graph per env and reset) instead of the shared
// CSR graph -- the flipped node's row of its env's matrix is the neighbour list (symmetric, integer-valued:
// checked at reset), one lane per entry, no atomics; max_local is per env.
template <typename T, bool VEC, bool DENSE>
__global__ __launch_bounds__(256) void k_spin_step(rls_spin_env env, int64_t B, int64_t N, int R,
                                                   const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ col,
                                                   const int32_t* __restrict__ wgt,
                                                   const T* __restrict__ matrix, const T* __restrict__ max_local_env,
                                                   const int64_t* __restrict__ action, T* __restrict__ reward,
                                                   uint8_t* __restrict__ visited_new, SpinRows rows, SpinStepArgs p) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (b >= B) return;
    T* st = reinterpret_cast<T*>(env.state) + b * R * N;
    T* spins = st;  // row 0 = SPIN_STATE, signed {+1, -1}
    int32_t* dl = env.delta + b * N;
    T* score = reinterpret_cast<T*>(env.score);
    T* best_score = reinterpret_cast<T*>(env.best_score);
    T* bs = reinterpret_cast<T*>(env.best_spins) + b * N;
    const int64_t a = action[b];
    // ExtraAction.PASS (spinsystem.py:349-351): action N changes no spin and no score; everything else of a step happens
    const bool is_pass = env.allow_pass != 0 && a == N;
    if (!is_pass && (uint64_t)a >= (uint64_t)N) {   // the reference raises an IndexError; leave the env untouched, report NaN
        if (lane == 0) {
            reward[b] = (T)NAN;
            if (visited_new) visited_new[b] = 0;
        }
        return;
    }
    const T max_local = DENSE ? max_local_env[b] : (T)p.max_local;
    T* best_obs = reinterpret_cast<T*>(env.best_obs_score);

    // 1. flip + score change (spinsystem_PECO.py:336-348): gain = delta[a] before the flip.  Neighbour updates
    //    are returning L2 atomics (multi-edges may hit one node twice in a wave-instruction): every lane sees
    //    the value its own add replaced, so the <= 0 census telescopes correctly.
    T s_new = (T)0;
    int gain = 0, adj = 0;
    if (!is_pass) {
    const T s_old = spins[a];
    s_new = -s_old;
    const int sn = s_new > (T)0 ? 1 : -1;
    const int da = dl[a];
    gain = da;
    if constexpr (DENSE) {
        const T* wrow = matrix + (b * N + a) * N;
        // a diagonal entry (the reference's BA training graphs carry W_ii = +-1 on their seed clique, util_envs_PECO.py:93-95)
        // counts as the reference counts it: score change = -(s' * (W s'))_a = delta_a - 2 W_aa  (spinsystem_PECO.py:346-348)
        gain = da - 2 * (int)wrow[a];
        T* imm = rows.immediate >= 0 ? st + (int64_t)rows.immediate * N : nullptr;
        for (int64_t n = lane; n < N; n += kWave) {
            const T w = wrow[n];
            if (w != (T)0 && n != a) {
                const int sj = spins[n] > (T)0 ? 1 : -1;
                const int c = 2 * (int)w * sj * sn;
                const int old = dl[n];
                dl[n] = old + c;
                adj += (int)((old + c) <= 0) - (int)(old <= 0);
                if (imm) imm[n] = (T)(old + c) / max_local;
            }
        }
        if (lane == 0) {
            adj += (int)(-gain <= 0) - (int)(da <= 0);
            dl[a] = -gain;
            spins[a] = s_new;
            if (imm) imm[a] = (T)(-gain) / max_local;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    } else {
        const int r0 = rowptr[a], r1 = rowptr[a + 1];
        for (int j = r0 + lane; j < r1; j += kWave) {
            const int nb = col[j];
            const int w = wgt ? wgt[j] : 1;
            const int sj = spins[nb] > (T)0 ? 1 : -1;
            const int c = 2 * w * sj * sn;
            const int old = atomicAdd(&dl[nb], c);
            adj += (int)((old + c) <= 0) - (int)(old <= 0);
        }
        if (lane == 0) {
            adj += (int)(-gain <= 0) - (int)(gain <= 0);
            __hip_atomic_store(&dl[a], -gain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            spins[a] = s_new;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (rows.immediate >= 0) {   // only the entries whose gain changed (agent-scope loads see the atomics' results)
            T* imm = st + (int64_t)rows.immediate * N;
            for (int j = r0 + lane; j < r1; j += kWave) {
                const int nb = col[j];
                const int d = __hip_atomic_load(&dl[nb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                imm[nb] = (T)d / max_local;
            }
            if (lane == 0) imm[a] = (T)(-gain) / max_local;
        }
    }
    }   // !is_pass
    const int nonpos = env.num_nonpos[b] + wave_sum_i32(adj);

    // 2. reward w.r.t. the best observed score, best tracking (:366-401)
    const T sc = score[b] + (T)gain;
    const T best_before = best_score[b];
    const T improvement = sc - best_obs[b];            // w.r.t. the best OBSERVABLE score: the best of the finite memory, else the best ever
    T rew;
    if (p.reward_mode == 1) rew = improvement > (T)0 ? improvement : (T)0;                              // BLS
    else if (p.reward_mode == 2) rew = improvement > (T)0 ? improvement / (improvement + (T)0.1) : (T)0; // CUSTOM_BLS
    else if (p.reward_mode == 3) rew = improvement > (T)0 ? improvement / (improvement + (T)0.05) : (T)0; // CUSTOM_BLS, score in half units
    else rew = (T)gain;                                                                                  // DENSE
    rew = rew / (T)p.reward_div;                                                                        // norm_rewards
    const bool new_best = sc > best_before;
    const T best_now = new_best ? sc : best_before;

    // 3. visited-state memory (util_envs_PECO.py:228-288 / util_envs.py:355-381): exact compare of the bit-packed
    //    spins against every earlier state of this env, Zobrist hash as the pre-filter
    bool fresh = true;
    const int64_t W = (N + (env.allow_pass ? 1 : 0) + 63) >> 6;      // (bit N = parity of the PASS actions: util_envs.py:361-366 toggles it)
    uint64_t* pk = env.packed ? env.packed + b * W : nullptr;
    if (pk) {
        const int64_t wa = a >> 6;
        const uint64_t bit = 1ull << (a & 63);
        const uint64_t h = env.hash[b] ^ spin_zobrist((uint32_t)a);
        if (env.hist) {
            const uint64_t* hh = env.hist_hash + b * env.hist_cap;
            const uint64_t* hs = env.hist + b * env.hist_cap * W;
            bool found = false;
            for (int64_t t0 = 0; t0 < p.hist_len; t0 += kWave) {
                const int64_t t = t0 + lane;
                bool cand = t < p.hist_len && hh[t] == h;
                if (cand) {
                    const uint64_t* e = hs + t * W;
                    for (int64_t k = 0; k < W && cand; ++k) cand = e[k] == (pk[k] ^ (k == wa ? bit : 0ull));
                }
                found = found || cand;
            }
            fresh = ballot64(found) == 0;
            __builtin_amdgcn_wave_barrier();
            uint64_t* dst = env.hist + (b * env.hist_cap + p.hist_len) * W;
            for (int64_t k = lane; k < W; k += kWave) dst[k] = pk[k] ^ (k == wa ? bit : 0ull);
            if (lane == 0) env.hist_hash[b * env.hist_cap + p.hist_len] = h;
            if (p.use_stag && !fresh) rew = rew - (T)p.stag_punishment;
            if (p.use_basin && fresh && nonpos == (int)N) rew = rew + (T)p.basin_reward;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {                       // issued after every lane's read of the old word above
            pk[wa] ^= bit;
            env.hash[b] = h;
        }
    }

    // 4. best spins / Hamming distance to them, kept incrementally
    int ham;
    if (new_best) {
        ham = 0;
        if constexpr (VEC) {
            using V = typename RowVec<T>::type;
            constexpr int PER = RowVec<T>::n;
            const V* src = reinterpret_cast<const V*>(spins);
            V* dst = reinterpret_cast<V*>(bs);
            for (int64_t i = lane; i < N / PER; i += kWave) {
                V v = src[i];
#pragma unroll
                for (int q = 0; q < PER; ++q) if (i * PER + q == a) v[q] = s_new;
                dst[i] = v;
            }
        } else {
            for (int64_t n = lane; n < N; n += kWave) bs[n] = (n == a) ? s_new : spins[n];
        }
    } else {
        ham = env.dist_best[b] + (is_pass ? 0 : ((bs[a] != s_new) ? 1 : -1));
    }
    const int ham_best = ham;              // distance to the best-ever spins (kept incrementally across steps)

    // 4b. finite memory (spinsystem.py:398-404): the ring of the last mem_len scores / spin configurations; the best
    //     OBSERVABLE score and spins are the ring's maximum (first on ties, numpy argmax) -- what the reward of the next
    //     step and the two distance rows refer to
    T best_obs_now = best_now;
    if (env.mem_len > 0) {
        const int64_t M = env.mem_len, Wn = (N + 63) >> 6;
        T* ms = reinterpret_cast<T*>(env.mem_score) + b * M;
        uint64_t* mp = env.mem_spins + b * M * Wn;
        const int64_t pos = (p.hist_len + 1) % M;           // idx_memory starts at 1 and advances once per step
        const uint64_t passbit = env.allow_pass ? (1ull << (N & 63)) : 0ull;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the flipped bit of pk is in memory
        __builtin_amdgcn_wave_barrier();
        for (int64_t k = lane; k < Wn; k += kWave) mp[pos * Wn + k] = pk[k] & ~(k == (N >> 6) ? passbit : 0ull);
        if (lane == 0) ms[pos] = sc;
        T bv = (T)-INFINITY;
        int64_t bi = INT64_MAX;
        for (int64_t t = lane; t < M; t += kWave) {
            const T v = (t == pos) ? sc : ms[t];
            if (v > bv) { bv = v; bi = t; }
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            const T ov = __shfl_xor(bv, sft, kWave);
            const int64_t oi = __shfl_xor(bi, sft, kWave);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        best_obs_now = bv;
        int hm = 0;
        if (bi != pos)
            for (int64_t k = lane; k < Wn; k += kWave)
                hm += __builtin_popcountll((pk[k] & ~(k == (N >> 6) ? passbit : 0ull)) ^ mp[bi * Wn + k]);
        ham = wave_sum_i32(hm);
    }

    // 5. the rows that change everywhere every step (:412-451) are a table lookup away: remember when a flipped and the four
    //    broadcast values; rls_spin_observation / rls_spin_materialize write the rows
    if (lane == 0) {
        if (!is_pass) env.last_flip[b * N + a] = (int32_t)(p.hist_len + 1);
        T* sc4 = reinterpret_cast<T*>(env.scalars) + b * 4;
        sc4[0] = (T)p.termination_value;
        sc4[1] = (T)1 - (T)nonpos / (T)N;
        sc4[2] = (T)fabs((double)(sc - best_obs_now)) / max_local;
        sc4[3] = (T)ham;
        best_obs[b] = best_obs_now;
    }
    if (lane == 0) {
        score[b] = sc;
        best_score[b] = best_now;
        reward[b] = rew;
        env.num_nonpos[b] = nonpos;
        env.dist_best[b] = ham_best;
        if (visited_new) visited_new[b] = fresh ? 1 : 0;
    }
}

// reset: rows and per-env statistics from the signed spins in row 0 and the gain cache (spinsystem_PECO.py:150-195,
// spinsystem.py:176-252): immediate row, greedy row, every other row zero, score = cut, best := current, census,
// packed spins + hash, empty history.
template <typename T>
__global__ __launch_bounds__(256) void k_spin_reset(rls_spin_env env, int64_t B, int64_t N, int R, SpinRows rows,
                                                    double max_local_d, int64_t weight_sum_all,
                                                    const T* __restrict__ max_local_env, const T* __restrict__ weight_sum_env) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (b >= B) return;
    T* st = reinterpret_cast<T*>(env.state) + b * R * N;
    const int32_t* dl = env.delta + b * N;
    T* bs = reinterpret_cast<T*>(env.best_spins) + b * N;
    const T max_local = max_local_env ? max_local_env[b] : (T)max_local_d;
    const int64_t weight_sum = weight_sum_env ? (int64_t)weight_sum_env[b] : weight_sum_all;
    int nonpos = 0;
    int64_t dsum = 0;
    uint64_t h = 0;
    const int64_t W = (N + (env.allow_pass ? 1 : 0) + 63) >> 6, Wn = (N + 63) >> 6;
    if (env.packed && W > Wn && lane == 0) env.packed[b * W + Wn] = 0;      // (the PASS parity bit alone in the last word)
    for (int64_t n0 = 0; n0 < N; n0 += kWave) {
        const int64_t n = n0 + lane;
        const bool in = n < N;
        const T s = in ? st[n] : (T)-1;
        const int d = in ? dl[n] : 1;
        nonpos += (in && d <= 0);
        dsum += in ? d : 0;
        if (in) bs[n] = s;
        const uint64_t word = ballot64(in && s > (T)0);
        if (env.packed) {
            if (in && s > (T)0) h ^= spin_zobrist((uint32_t)n);
            if (lane == 0) env.packed[b * W + (n0 >> 6)] = word;
        }
        if (env.mem_len > 0 && lane < env.mem_len)      // the finite memory starts full of the initial configuration (spinsystem.py:206-209)
            for (int64_t t = lane; t < env.mem_len; t += kWave) env.mem_spins[(b * env.mem_len + t) * Wn + (n0 >> 6)] = word;
    }
    nonpos = wave_sum_i32(nonpos);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) dsum += __shfl_xor(dsum, m, 64);
    const T greedy = (T)1 - (T)nonpos / (T)N;
    for (int64_t n = lane; n < N; n += kWave) env.last_flip[b * N + n] = 0;
    if (lane == 0) {
        T* sc4 = reinterpret_cast<T*>(env.scalars) + b * 4;
        sc4[0] = (T)0; sc4[1] = greedy; sc4[2] = (T)0; sc4[3] = (T)0;
    }
    for (int r = 1; r < R; ++r) {
        T* row = st + (int64_t)r * N;
        for (int64_t n = lane; n < N; n += kWave) {
            T v = (T)0;
            if (r == rows.immediate) v = (T)dl[n] / max_local;
            else if (r == rows.greedy) v = greedy;
            row[n] = v;
        }
    }
    if (lane == 0) {
        // cut = 1/4 sum_ij W_ij (1 - s_i s_j) = (sum_ij W_ij - sum_i delta_i) / 4: an integer, exact in T
        const T sc = (T)(weight_sum - dsum) / (T)4;
        reinterpret_cast<T*>(env.score)[b] = sc;
        reinterpret_cast<T*>(env.best_score)[b] = sc;
        reinterpret_cast<T*>(env.best_obs_score)[b] = sc;
        env.num_nonpos[b] = nonpos;
        env.dist_best[b] = 0;
    }
    if (env.mem_len > 0) {
        const T sc = (T)(weight_sum - dsum) / (T)4;
        for (int64_t t = lane; t < env.mem_len; t += kWave) reinterpret_cast<T*>(env.mem_score)[b * env.mem_len + t] = sc;
    }
    if (env.packed) {
        h = wave_xor_u64(h);
        if (lane == 0) env.hash[b] = h;
    }
}

// Per-env couplings (spinsystem_PECO.py:150-170 with the generators of util_envs_PECO.py): from matrix T [B, N, N] and the
// signed spins in row 0 of state, the gain cache delta[b,i] = s_i sum_j W_ij s_j (_get_immeditate_cuts_avaialable, :660-661),
// max_local[b] = max_i sum_j W_ij (the same expression on all-ones spins, :162-168), weight_sum[b] = sum_ij W_ij, and
// flags[b]: bit 0 = the reference would draw the graph again (sum_i |sum_j W_ij| == 0 or max_local == 0), bit 1 = a matrix
// the integer gain cache cannot hold (non-integer entry, not symmetric).  A workgroup per env, a wave per row.
template <typename T>
__global__ __launch_bounds__(256) void k_spin_dense_prepare(const T* __restrict__ matrix, const T* __restrict__ state, int64_t B,
                                                             int64_t N, int R, int32_t* __restrict__ delta, T* __restrict__ max_local,
                                                             T* __restrict__ weight_sum, uint8_t* __restrict__ flags) {
    __shared__ double s_max[4], s_mnz[4], s_abs[4], s_tot[4];
    __shared__ int s_bad[4];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int64_t b = blockIdx.x;
    const T* m = matrix + b * N * N;
    const T* spins = state + b * R * N;
    double wmax = -INFINITY, wmnz = -INFINITY, wabs = 0.0, wtot = 0.0;   // wmnz: the maximum over the NONZERO row sums
    int bad = 0;
    for (int64_t i = wv; i < N; i += 4) {
        const T* row = m + i * N;
        double dot = 0.0, rs = 0.0;
        for (int64_t j = lane; j < N; j += kWave) {
            const T w = row[j];
            if (w != (T)0) {
                dot += (double)w * (spins[j] > (T)0 ? 1.0 : -1.0);
                rs += (double)w;
                bad |= (int)(w != (T)rint((double)w)) | (int)(m[j * N + i] != w);
            }
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            dot += __shfl_xor(dot, sft, 64);
            rs += __shfl_xor(rs, sft, 64);
        }
        if (lane == 0) delta[b * N + i] = (int32_t)((spins[i] > (T)0 ? 1.0 : -1.0) * dot);
        wmax = fmax(wmax, rs);
        if (rs != 0.0) wmnz = fmax(wmnz, rs);
        wabs += fabs(rs);
        wtot += rs;
    }
    bad = ballot64(bad != 0) != 0;
    if (lane == 0) { s_max[wv] = wmax; s_mnz[wv] = wmnz; s_abs[wv] = wabs; s_tot[wv] = wtot; s_bad[wv] = bad; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double mx = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
        const double ab = s_abs[0] + s_abs[1] + s_abs[2] + s_abs[3];
        // the numpy env takes the maximum over the NONZERO entries (spinsystem.py:190-196): it differs from the batched env's
        // rule only where that one draws again -- a zero maximum although rows with a (negative) sum exist -- and is delivered
        // for exactly that case, with flag bit 2
        const bool nz_case = mx == 0.0 && ab != 0.0;
        max_local[b] = (T)(nz_case ? fmax(fmax(s_mnz[0], s_mnz[1]), fmax(s_mnz[2], s_mnz[3])) : mx);
        weight_sum[b] = (T)(s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3]);
        flags[b] = (uint8_t)((ab == 0.0 || mx == 0.0 ? 1 : 0) | ((s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3]) ? 2 : 0) | (nz_case ? 4 : 0));
    }
}

// ---- batched random couplings for the training envs (util_envs_PECO.py:15-112).  The reference draws them with torch ops
// (ER: rand < p, triu, symmetrise; BA: a Python loop over the nodes with adj.sum + multinomial + two scatters per node:
// ~1000 launches and N full [B, N, N] reductions per reset at N = 200).  Build-defined counter-based draws (Philox keyed
// by seed and GLOBAL env id, so a shard draws what the whole batch would), the same distributions:
//   edge sign  EdgeType.UNIFORM (1): +1;  DISCRETE (2): one +-1 per node pair shared by all envs of the call (:23-26, 72-75);
//              RANDOM (3): one +-1 per node pair and env (:30-33, 79-82)
__device__ __forceinline__ float coupling_sign(const Philox& ph, int edge_type, uint64_t gb, uint32_t lo, uint32_t hi) {
    if (edge_type == 1) return 1.0f;
    uint32_t r[4];
    if (edge_type == 2) ph(lo, hi, 0u, 0x5349474Eu, r);                       // 'SIGN': no env in the counter
    else ph((uint32_t)gb, (uint32_t)(gb >> 32), lo * 65536u + hi, 0x5349474Fu, r);
    return (r[0] & 1u) ? 1.0f : -1.0f;
}

// Erdos-Renyi (RandomERGraphGenerator.generate_er_graph :42-52): pair (i < j) is an edge iff its 32-bit draw < p * 2^32.
// One thread per matrix element (both halves recompute the pair's draw): a streaming write of B * N^2 elements.
template <typename T>
__global__ __launch_bounds__(256) void k_rand_couplings_er(T* __restrict__ matrix, int64_t B, int64_t N, uint32_t threshold,
                                                            int all_edges, int edge_type, uint64_t seed, int64_t env_offset) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * N) return;
    const int64_t i = e / N, j = e - i * N;
    const uint32_t lo = (uint32_t)(i < j ? i : j), hi = (uint32_t)(i < j ? j : i);
    const Philox ph(seed);
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        const uint64_t gb = (uint64_t)(b + env_offset);
        float w = 0.0f;
        if (i != j) {
            uint32_t r[4];
            ph((uint32_t)gb, (uint32_t)(gb >> 32), lo * 65536u + hi, 0x45524750u, r);       // 'ERGP'
            if (all_edges || r[0] < threshold) w = coupling_sign(ph, edge_type, gb, lo, hi);
        }
        matrix[(b * N + i) * N + j] = (T)w;
    }
}

// Barabasi-Albert (RandomBAGraphGenerator.generate_barabasi_albert :84-107): a seed clique on nodes 0..m WITH its self-loops
// (adj[:, i, :i+1] = 1 sets the diagonal too, :93-95 -- kept: it is what the reference trains on), then every node v > m
// attaches to m distinct earlier nodes drawn in proportion to their degree (row sums of adj, self-loop counted once).
// Degree-proportional = uniform over the list of edge endpoints, duplicates drawn again: the list is never stored -- entry
// idx is a clique node for idx < (m+1)^2, else endpoint r of node vv's m edges: its target t[vv][r] (r < m) or vv itself.
// Lane = env (epw envs per wave: 64 when the batch fills the chip that way, fewer for smaller batches so that more waves share
// the writing), the targets of the wave's envs in LDS as uint16 [N * m][epw + 2]; then the wave writes each env's matrix: zero
// fill, wait for the stores, scatter the signed edges.
// q / d for q < 2^23 by a float reciprocal and one correction step (the divisors are run-time values: a generic 32-bit
// division is ~35 instructions, and the draw loop below is one dependent chain per env)
__device__ __forceinline__ uint32_t small_div(uint32_t q, uint32_t d, float inv_d) {
    uint32_t est = (uint32_t)((float)q * inv_d);
    const int32_t r = (int32_t)(q - est * d);
    est += r < 0 ? -1 : (r >= (int32_t)d ? 1 : 0);
    return est;
}

template <typename T>
__global__ __launch_bounds__(kWave) void k_rand_couplings_ba(T* __restrict__ matrix, int64_t B, int64_t N, int m, int edge_type,
                                                              uint64_t seed, int64_t env_offset, int epw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t* t = reinterpret_cast<uint16_t*>(smem);                         // [(v * m + r) * kStride + lane]
    const int kStride = epw + 2;
    const int lane = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * epw, b = b0 + lane;
    const Philox ph(seed);
    const uint64_t gb = (uint64_t)(b + env_offset);
    const uint32_t L0 = (uint32_t)(m + 1) * (uint32_t)(m + 1), m1 = (uint32_t)m + 1u, m2 = 2u * (uint32_t)m;
    const float inv_m1 = 1.0f / (float)m1, inv_m2 = 1.0f / (float)m2;
    for (uint32_t v = m1; v < (uint32_t)N && lane < epw; ++v) {
        const uint32_t L = L0 + m2 * (v - m1);
        int cnt = 0;
        uint32_t r[4];
        for (uint32_t a = 0; cnt < m; ++a) {
            if ((a & 3u) == 0) ph((uint32_t)gb, (uint32_t)(gb >> 32), v * 4096u + (a >> 2), 0x42414752u, r);   // 'BAGR'
            const uint32_t idx = __umulhi(r[a & 3u], L);
            uint32_t node;
            if (idx < L0) node = small_div(idx, m1, inv_m1);
            else {
                const uint32_t q = idx - L0, qd = small_div(q, m2, inv_m2), vv = m1 + qd, rr = q - qd * m2;
                node = rr < (uint32_t)m ? t[(vv * m + rr) * kStride + lane] : vv;
            }
            bool dup = false;
            for (int c = 0; c < cnt; ++c) dup |= t[(v * m + c) * kStride + lane] == node;
            if (!dup) { t[(v * m + cnt) * kStride + lane] = (uint16_t)node; ++cnt; }
        }
    }
    __syncthreads();
    const int64_t nb = (B - b0 < epw) ? B - b0 : epw;
    for (int64_t e = 0; e < nb; ++e) {
        T* me = matrix + (b0 + e) * N * N;
        for (int64_t k = lane; k < N * N; k += kWave) me[k] = (T)0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int64_t e = 0; e < nb; ++e) {
        T* me = matrix + (b0 + e) * N * N;
        const uint64_t ge = (uint64_t)(b0 + e + env_offset);
        for (int64_t k = lane; k < (int64_t)L0; k += kWave) {               // the seed clique, diagonal included
            const int64_t i = k / (m + 1), j = k - i * (m + 1);
            me[i * N + j] = (T)coupling_sign(ph, edge_type, ge, (uint32_t)(i < j ? i : j), (uint32_t)(i < j ? j : i));
        }
        for (int64_t k = (int64_t)(m + 1) * m + lane; k < N * m; k += kWave) {
            const int64_t v = k / m, u = t[k * kStride + e];
            const T w = (T)coupling_sign(ph, edge_type, ge, (uint32_t)u, (uint32_t)v);
            me[v * N + u] = w;
            me[u * N + v] = w;
        }
    }
}

// The value of observable row rr of env b at columns c .. c + PER - 1 as the reference's state tensor would hold it after
// `step` steps: resident rows (spins, immediate reward, anything this kernel family does not know) come from `state`, the
// others from last_flip / time_table / scalars (see the header of this file).
template <typename T, int PER>
__device__ __forceinline__ void spin_row_values(const rls_spin_env& env, const SpinRows& rows, int64_t b, int R, int64_t N, int64_t rr,
                                                int64_t c, int64_t step, T (&v)[PER]) {
    const T* table = reinterpret_cast<const T*>(env.time_table);
    const T* sc4 = reinterpret_cast<const T*>(env.scalars) + b * 4;
    if (rr == rows.time_since_flip) {
        const int32_t* lf = env.last_flip + b * N + c;
#pragma unroll
        for (int q = 0; q < PER; ++q) v[q] = table[step - lf[q]];
    } else if (rr == rows.episode_time) {
#pragma unroll
        for (int q = 0; q < PER; ++q) v[q] = table[step];
    } else if (rr == rows.termination || rr == rows.greedy || rr == rows.dist_score || rr == rows.dist_state) {
        const T x = sc4[rr == rows.termination ? 0 : (rr == rows.greedy ? 1 : (rr == rows.dist_score ? 2 : 3))];
#pragma unroll
        for (int q = 0; q < PER; ++q) v[q] = x;
    } else {
        const T* src = reinterpret_cast<const T*>(env.state) + (b * R + rr) * N + c;
#pragma unroll
        for (int q = 0; q < PER; ++q) v[q] = src[q];
    }
}

// get_observation (spinsystem_PECO.py:455 / spinsystem.py:484-495): out[b] = the R observable rows of env b (row 0 mapped
// from signed to {0, 1} spins under SpinBasis.BINARY: (1 - s) / 2) followed by the N rows of the matrix.  One streaming
// pass: the reference clones the state, rewrites row 0 and concatenates a [B, N, N] expansion of the matrix.
template <typename T, bool V4>
__global__ __launch_bounds__(256) void k_spin_observation(rls_spin_env env, SpinRows rows, int64_t step, const T* __restrict__ matrix,
                                                           int64_t matrix_env_stride, int64_t B, int R, int64_t N, int binary,
                                                           T* __restrict__ out) {
    constexpr int PER = V4 ? (int)(16 / sizeof(T)) : 1;
    const int64_t nrows = R + (matrix ? N : 0), per_row = N / PER, per_env = nrows * per_row;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_env) return;
    const int64_t rr = idx / per_row, c = (idx - rr * per_row) * PER;
    using V = typename RowVec<T>::type;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        T v[PER];
        if (rr < R) {
            spin_row_values<T, PER>(env, rows, b, R, N, rr, c, step, v);
            if (binary && rr == 0) {
#pragma unroll
                for (int q = 0; q < PER; ++q) v[q] = ((T)1 - v[q]) / (T)2;
            }
        } else {
            const T* src = matrix + b * matrix_env_stride + (rr - R) * N + c;
            if constexpr (V4) {
                const V m = *reinterpret_cast<const V*>(src);
#pragma unroll
                for (int q = 0; q < PER; ++q) v[q] = m[q];
            } else {
                v[0] = src[0];
            }
        }
        T* dst = out + (b * nrows + rr) * N + c;
        if constexpr (V4) {
            V o;
#pragma unroll
            for (int q = 0; q < PER; ++q) o[q] = v[q];
            *reinterpret_cast<V*>(dst) = o;
        } else {
            dst[0] = v[0];
        }
    }
}

// The R observable rows of the observation, a thread per (env, 16-byte column chunk): the chunk's spins, immediate rewards and
// last-flip steps are read ONCE and all R rows of the chunk leave from registers (the element-per-(row, chunk) form above
// re-derived row kinds and re-read the per-env scalars for every 16 bytes: 0.54 of HBM on the rows-only observation).
template <typename T, bool V4>
__global__ __launch_bounds__(256) void k_spin_observation_rows(rls_spin_env env, SpinRows rows, int64_t step, int64_t B, int R, int64_t N,
                                                                int64_t out_rows, int binary, T* __restrict__ out) {
    constexpr int PER = V4 ? (int)(16 / sizeof(T)) : 1;
    const int64_t per_row = N / PER;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_row) return;
    const int64_t c = idx * PER;
    using V = typename RowVec<T>::type;
    for (int64_t b = blockIdx.y; b < B; b += gridDim.y) {
        for (int rr = 0; rr < R; ++rr) {
            T v[PER];
            spin_row_values<T, PER>(env, rows, b, R, N, rr, c, step, v);
            if (binary && rr == 0) {
#pragma unroll
                for (int q = 0; q < PER; ++q) v[q] = ((T)1 - v[q]) / (T)2;
            }
            T* dst = out + (b * out_rows + rr) * N + c;
            if constexpr (V4) {
                V o;
#pragma unroll
                for (int q = 0; q < PER; ++q) o[q] = v[q];
                __builtin_nontemporal_store(o, reinterpret_cast<V*>(dst));
            } else {
                dst[0] = v[0];
            }
        }
    }
}

// The rows that a step does not store, written into `state` for whoever reads that tensor (rls_spin_materialize).
template <typename T, bool V4>
__global__ __launch_bounds__(256) void k_spin_materialize(rls_spin_env env, SpinRows rows, int64_t step, int64_t B, int R, int64_t N) {
    constexpr int PER = V4 ? (int)(16 / sizeof(T)) : 1;
    const int64_t per_row = N / PER;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_row) return;
    const int64_t c = idx * PER;
    const int64_t rr = blockIdx.y;   // one of the six lazy rows
    const int which[6] = {rows.time_since_flip, rows.episode_time, rows.termination, rows.greedy, rows.dist_score, rows.dist_state};
    const int r = which[rr];
    if (r < 0) return;
    using V = typename RowVec<T>::type;
    for (int64_t b = blockIdx.z; b < B; b += gridDim.z) {
        T v[PER];
        spin_row_values<T, PER>(env, rows, b, R, N, r, c, step, v);
        T* dst = reinterpret_cast<T*>(env.state) + (b * R + r) * N + c;
        if constexpr (V4) {
            V o;
#pragma unroll
            for (int q = 0; q < PER; ++q) o[q] = v[q];
            *reinterpret_cast<V*>(dst) = o;
        } else {
            dst[0] = v[0];
        }
    }
}

}  // namespace rls

using namespace rls;

static int check_spin_env(const rls_spin_env* env, int state_bytes, int32_t num_rows, const int32_t* row_index,
                          SpinRows* rows) {
    RLS_REQUIRE(env && row_index, RLS_EINVAL, "env / row_index is NULL");
    RLS_REQUIRE(state_bytes == 4 || state_bytes == 8, RLS_EINVAL, "state_bytes must be 4 (f32) or 8 (f64)");
    RLS_REQUIRE(num_rows >= 1, RLS_EINVAL, "num_rows < 1");
    RLS_REQUIRE(env->state && env->delta && env->score && env->best_score && env->best_spins && env->num_nonpos &&
                    env->dist_best && env->last_flip && env->scalars && env->time_table, RLS_EINVAL, "NULL pointer in rls_spin_env");
    RLS_REQUIRE(env->best_obs_score, RLS_EINVAL, "best_obs_score is NULL");
    RLS_REQUIRE(!env->packed || env->hash, RLS_EINVAL, "packed spins need their hash");
    RLS_REQUIRE(!env->hist || (env->packed && env->hist_hash && env->hist_cap > 0), RLS_EINVAL,
                "visited-state memory needs packed, hash, hist, hist_hash and hist_cap > 0");
    RLS_REQUIRE(env->mem_len >= 0 && (env->mem_len == 0 || (env->packed && env->mem_spins && env->mem_score)), RLS_EINVAL,
                "a finite memory needs packed, mem_spins and mem_score");
    *rows = SpinRows{row_index[0], row_index[1], row_index[2], row_index[3], row_index[4], row_index[5], row_index[6]};
    const int* ri = &rows->immediate;
    for (int k = 0; k < 7; ++k) RLS_REQUIRE(ri[k] < num_rows && ri[k] != 0, RLS_EINVAL, "row_index[%d]=%d out of range", k, ri[k]);
    return RLS_OK;
}

extern "C" {

int rls_spin_reset(const rls_graph* g, const rls_spin_env* env, int state_bytes, int64_t B, int32_t num_rows,
                   const int32_t* row_index, double max_local, int64_t weight_sum, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    SpinRows rows;
    if (int rc = check_spin_env(env, state_bytes, num_rows, row_index, &rows)) return rc;
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(max_local != 0.0, RLS_EINVAL, "max_local_reward_available is 0");
    const dim3 grid((unsigned)ceil_div(B, 4)), block(256);
    if (state_bytes == 4)
        hipLaunchKernelGGL(k_spin_reset<float>, grid, block, 0, as_stream(stream), *env, B, g->num_nodes, num_rows, rows,
                           max_local, weight_sum, (const float*)nullptr, (const float*)nullptr);
    else
        hipLaunchKernelGGL(k_spin_reset<double>, grid, block, 0, as_stream(stream), *env, B, g->num_nodes, num_rows, rows,
                           max_local, weight_sum, (const double*)nullptr, (const double*)nullptr);
    return check_launch("k_spin_reset");
}

int rls_spin_reset_dense(const void* matrix, const rls_spin_env* env, int state_bytes, int64_t B, int64_t N, int32_t num_rows,
                         const int32_t* row_index, void* max_local, void* weight_sum, uint8_t* flags, void* stream) {
    RLS_REQUIRE(B >= 0 && N > 0 && N < (1ll << 24), RLS_EINVAL, "bad sizes B=%lld N=%lld", (long long)B, (long long)N);
    SpinRows rows;
    if (int rc = check_spin_env(env, state_bytes, num_rows, row_index, &rows)) return rc;
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(matrix && max_local && weight_sum && flags, RLS_EINVAL, "NULL pointer");
    hipStream_t s = as_stream(stream);
    const dim3 grid((unsigned)ceil_div(B, 4)), block(256);
    if (state_bytes == 4) {
        hipLaunchKernelGGL(k_spin_dense_prepare<float>, dim3((unsigned)B), block, 0, s, (const float*)matrix, (const float*)env->state,
                           B, N, (int)num_rows, env->delta, (float*)max_local, (float*)weight_sum, flags);
        hipLaunchKernelGGL(k_spin_reset<float>, grid, block, 0, s, *env, B, N, num_rows, rows, 1.0, (int64_t)0,
                           (const float*)max_local, (const float*)weight_sum);
    } else {
        hipLaunchKernelGGL(k_spin_dense_prepare<double>, dim3((unsigned)B), block, 0, s, (const double*)matrix, (const double*)env->state,
                           B, N, (int)num_rows, env->delta, (double*)max_local, (double*)weight_sum, flags);
        hipLaunchKernelGGL(k_spin_reset<double>, grid, block, 0, s, *env, B, N, num_rows, rows, 1.0, (int64_t)0,
                           (const double*)max_local, (const double*)weight_sum);
    }
    return check_launch("k_spin_reset (dense)");
}

static int spin_step_common(const rls_graph* g, const void* matrix, const void* max_local_env, int64_t N, const rls_spin_env* env,
                            int state_bytes, int64_t B, int32_t num_rows, const int32_t* row_index, const int64_t* action,
                            void* reward, uint8_t* visited_new, double max_local, double termination_value,
                            int32_t reward_mode, double reward_div, int64_t hist_len, int32_t use_stag, double stag_punishment,
                            int32_t use_basin, double basin_reward, void* stream) {
    RLS_REQUIRE(B >= 0, RLS_EINVAL, "B < 0");
    SpinRows rows;
    if (int rc = check_spin_env(env, state_bytes, num_rows, row_index, &rows)) return rc;
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(action && reward, RLS_EINVAL, "action / reward is NULL");
    RLS_REQUIRE(reward_mode >= 0 && reward_mode <= 3, RLS_EINVAL,
                "reward_mode must be 0 (DENSE), 1 (BLS), 2 (CUSTOM_BLS), 3 (CUSTOM_BLS on a score kept in half units)");
    RLS_REQUIRE(max_local != 0.0 && reward_div != 0.0, RLS_EINVAL, "max_local_reward_available / reward_div is 0");
    RLS_REQUIRE(!(use_stag || use_basin) || env->hist, RLS_EINVAL, "stag_punishment / basin_reward need the visited-state memory");
    RLS_REQUIRE(!env->hist || (hist_len >= 0 && hist_len < env->hist_cap), RLS_EINVAL,
                "hist_len %lld outside [0, hist_cap = %lld)", (long long)hist_len, (long long)env->hist_cap);
    RLS_REQUIRE(hist_len >= 0 && hist_len + 1 < env->table_len, RLS_EINVAL, "step %lld outside the time table of %lld entries",
                (long long)hist_len + 1, (long long)env->table_len);
    const bool vec = ((((uintptr_t)env->state) | ((uintptr_t)env->best_spins)) & 15) == 0 && (N * state_bytes) % 16 == 0;
    SpinStepArgs p{max_local, termination_value, reward_div, stag_punishment, basin_reward,
                   reward_mode, use_stag, use_basin, hist_len};
    const dim3 grid((unsigned)ceil_div(B, 4)), block(256);
    hipStream_t s = as_stream(stream);
#define LAUNCH_SPIN(T, VEC)                                                                                                     \
    do {                                                                                                                        \
        if (matrix)                                                                                                             \
            hipLaunchKernelGGL((k_spin_step<T, VEC, true>), grid, block, 0, s, *env, B, N, num_rows, nullptr, nullptr, nullptr,   \
                               (const T*)matrix, (const T*)max_local_env, action, (T*)reward, visited_new, rows, p);           \
        else                                                                                                                    \
            hipLaunchKernelGGL((k_spin_step<T, VEC, false>), grid, block, 0, s, *env, B, N, num_rows, g->rowptr, g->col, g->wgt,  \
                               (const T*)nullptr, (const T*)nullptr, action, (T*)reward, visited_new, rows, p);                \
    } while (0)
    if (state_bytes == 4) { if (vec) LAUNCH_SPIN(float, true); else LAUNCH_SPIN(float, false); }
    else                  { if (vec) LAUNCH_SPIN(double, true); else LAUNCH_SPIN(double, false); }
#undef LAUNCH_SPIN
    return check_launch("k_spin_step");
}

int rls_spin_step(const rls_graph* g, const rls_spin_env* env, int state_bytes, int64_t B, int32_t num_rows,
                  const int32_t* row_index, const int64_t* action, void* reward, uint8_t* visited_new, double max_local,
                  double termination_value, int32_t reward_mode, double reward_div, int64_t hist_len,
                  int32_t use_stag, double stag_punishment, int32_t use_basin, double basin_reward, void* stream) {
    if (int rc = check_graph(g)) return rc;
    return spin_step_common(g, nullptr, nullptr, g->num_nodes, env, state_bytes, B, num_rows, row_index, action, reward, visited_new,
                            max_local, termination_value, reward_mode, reward_div, hist_len, use_stag, stag_punishment,
                            use_basin, basin_reward, stream);
}

int rls_spin_step_dense(const void* matrix, const void* max_local, const rls_spin_env* env, int state_bytes, int64_t B, int64_t N,
                        int32_t num_rows, const int32_t* row_index, const int64_t* action, void* reward, uint8_t* visited_new,
                        double termination_value, int32_t reward_mode, double reward_div, int64_t hist_len,
                        int32_t use_stag, double stag_punishment, int32_t use_basin, double basin_reward, void* stream) {
    RLS_REQUIRE(N > 0 && N < (1ll << 24), RLS_EINVAL, "bad N=%lld", (long long)N);
    RLS_REQUIRE(B == 0 || (matrix && max_local), RLS_EINVAL, "matrix / max_local is NULL");
    return spin_step_common(nullptr, matrix, max_local, N, env, state_bytes, B, num_rows, row_index, action, reward, visited_new, 1.0,
                            termination_value, reward_mode, reward_div, hist_len, use_stag, stag_punishment, use_basin,
                            basin_reward, stream);
}

int rls_rand_couplings(void* matrix, int state_bytes, int64_t B, int64_t N, int32_t kind, double p_connection, int32_t m_insertion_edges,
                       int32_t edge_type, uint64_t seed, int64_t env_offset, void* stream) {
    RLS_REQUIRE(B >= 0 && N > 0 && N < 65536, RLS_EINVAL, "bad sizes B=%lld N=%lld", (long long)B, (long long)N);
    RLS_REQUIRE(state_bytes == 4 || state_bytes == 8, RLS_EINVAL, "state_bytes must be 4 (f32) or 8 (f64)");
    RLS_REQUIRE(edge_type >= 1 && edge_type <= 3, RLS_EINVAL, "edge_type must be 1 (UNIFORM), 2 (DISCRETE), 3 (RANDOM)");
    RLS_REQUIRE(kind == 0 || kind == 1, RLS_EINVAL, "kind must be 0 (Erdos-Renyi) or 1 (Barabasi-Albert)");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(matrix, RLS_EINVAL, "matrix is NULL");
    hipStream_t s = as_stream(stream);
    if (kind == 0) {
        RLS_REQUIRE(p_connection >= 0.0 && p_connection <= 1.0, RLS_EINVAL, "p_connection outside [0, 1]");
        const double scaled = p_connection * 4294967296.0;
        const uint32_t threshold = scaled >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)scaled;
        const int all = p_connection >= 1.0;
        const dim3 grid((unsigned)ceil_div(N * N, 256), (unsigned)(B < 16384 ? B : 16384)), block(256);
        if (state_bytes == 4)
            hipLaunchKernelGGL(k_rand_couplings_er<float>, grid, block, 0, s, (float*)matrix, B, N, threshold, all, (int)edge_type, seed, env_offset);
        else
            hipLaunchKernelGGL(k_rand_couplings_er<double>, grid, block, 0, s, (double*)matrix, B, N, threshold, all, (int)edge_type, seed, env_offset);
        return check_launch("k_rand_couplings_er");
    }
    const int m = m_insertion_edges;
    RLS_REQUIRE(m >= 1 && m + 1 <= N && m < 4096, RLS_EINVAL, "m_insertion_edges=%d outside [1, N - 1]", m);
    int epw = kWave;                                                         // envs per wave: >= 2 waves per CU before lanes fill up
    while (epw > 8 && B < (int64_t)epw * 2 * num_cus()) epw >>= 1;
    while (epw > 1 && (size_t)N * m * (epw + 2) * sizeof(uint16_t) > (size_t)kLdsBytes) epw >>= 1;
    const size_t lds = (size_t)N * m * (epw + 2) * sizeof(uint16_t);
    RLS_REQUIRE(lds <= (size_t)kLdsBytes, RLS_EUNSUPPORTED, "N * m = %lld needs %zu B of LDS (max %d)", (long long)N * m, lds, kLdsBytes);
    const dim3 grid((unsigned)ceil_div(B, epw)), block(kWave);
    if (state_bytes == 4) {
        auto kern = k_rand_couplings_ba<float>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, s, (float*)matrix, B, N, m, (int)edge_type, seed, env_offset, epw);
    } else {
        auto kern = k_rand_couplings_ba<double>;
        if (lds > 64 * 1024) ensure_dyn_lds((const void*)kern, lds);
        hipLaunchKernelGGL(kern, grid, block, lds, s, (double*)matrix, B, N, m, (int)edge_type, seed, env_offset, epw);
    }
    return check_launch("k_rand_couplings_ba");
}

int rls_spin_observation(const rls_spin_env* env, const void* matrix, int32_t matrix_per_env, int state_bytes, int64_t B,
                         int32_t num_rows, int64_t N, const int32_t* row_index, int64_t step_index, int32_t binary_basis, void* out,
                         void* stream) {
    RLS_REQUIRE(B >= 0 && num_rows >= 1 && N > 0, RLS_EINVAL, "bad sizes B=%lld R=%d N=%lld", (long long)B, num_rows, (long long)N);
    SpinRows rows;
    if (int rc = check_spin_env(env, state_bytes, num_rows, row_index, &rows)) return rc;
    RLS_REQUIRE(step_index >= 0 && step_index < env->table_len, RLS_EINVAL, "step_index %lld outside the time table", (long long)step_index);
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(out, RLS_EINVAL, "out is NULL");
    const int per = 16 / state_bytes;
    const bool v4 = (N % per) == 0 && ((((uintptr_t)env->state) | ((uintptr_t)out) | ((uintptr_t)matrix)) & 15) == 0;
    const int64_t nrows = num_rows + (matrix ? N : 0);
    const int64_t per_env = nrows * (v4 ? N / per : N);
    RLS_REQUIRE(per_env < (1ll << 31) * 256, RLS_EUNSUPPORTED, "observation of %lld elements per env", (long long)(nrows * N));
    const dim3 grid((unsigned)ceil_div(per_env, 256), (unsigned)(B < 16384 ? B : 16384)), block(256);
    hipStream_t s = as_stream(stream);
    if (!matrix) {   // rows only: a thread per (env, column chunk) writes all the rows of its chunk
        const dim3 rgrid((unsigned)ceil_div(v4 ? N / per : N, 256), (unsigned)(B < 65535 ? B : 65535));
#define LAUNCH_ROWS(T, V4)                                                                                                    \
    hipLaunchKernelGGL((k_spin_observation_rows<T, V4>), rgrid, block, 0, s, *env, rows, step_index, B, (int)num_rows, N, nrows,  \
                       (int)binary_basis, (T*)out)
        if (state_bytes == 4) { if (v4) LAUNCH_ROWS(float, true); else LAUNCH_ROWS(float, false); }
        else                  { if (v4) LAUNCH_ROWS(double, true); else LAUNCH_ROWS(double, false); }
#undef LAUNCH_ROWS
        return check_launch("k_spin_observation_rows");
    }
#define LAUNCH_OBS(T, V4)                                                                                                     \
    hipLaunchKernelGGL((k_spin_observation<T, V4>), grid, block, 0, s, *env, rows, step_index, (const T*)matrix,                \
                       (int64_t)(matrix_per_env ? N * N : 0), B, (int)num_rows, N, (int)binary_basis, (T*)out)
    if (state_bytes == 4) { if (v4) LAUNCH_OBS(float, true); else LAUNCH_OBS(float, false); }
    else                  { if (v4) LAUNCH_OBS(double, true); else LAUNCH_OBS(double, false); }
#undef LAUNCH_OBS
    return check_launch("k_spin_observation");
}

int rls_spin_materialize(const rls_spin_env* env, int state_bytes, int64_t B, int64_t N, int32_t num_rows, const int32_t* row_index,
                         int64_t step_index, void* stream) {
    RLS_REQUIRE(B >= 0 && N > 0, RLS_EINVAL, "bad sizes B=%lld N=%lld", (long long)B, (long long)N);
    SpinRows rows;
    if (int rc = check_spin_env(env, state_bytes, num_rows, row_index, &rows)) return rc;
    RLS_REQUIRE(step_index >= 0 && step_index < env->table_len, RLS_EINVAL, "step_index %lld outside the time table", (long long)step_index);
    if (B == 0) return RLS_OK;
    const int per = 16 / state_bytes;
    const bool v4 = (N % per) == 0 && (((uintptr_t)env->state) & 15) == 0;
    const dim3 grid((unsigned)ceil_div(v4 ? N / per : N, 256), 6, (unsigned)(B < 8192 ? B : 8192)), block(256);
    hipStream_t s = as_stream(stream);
#define LAUNCH_MAT(T, V4) hipLaunchKernelGGL((k_spin_materialize<T, V4>), grid, block, 0, s, *env, rows, step_index, B, (int)num_rows, N)
    if (state_bytes == 4) { if (v4) LAUNCH_MAT(float, true); else LAUNCH_MAT(float, false); }
    else                  { if (v4) LAUNCH_MAT(double, true); else LAUNCH_MAT(double, false); }
#undef LAUNCH_MAT
    return check_launch("k_spin_materialize");
}

}  // extern "C"
