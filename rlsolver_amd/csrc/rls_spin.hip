// S2V / ECO / PECO spin-system step: the batched env of
// rlsolver/methods/ECO_S2V/src/envs/spinsystem_PECO.py:306-486 (and its shared-graph inference twin
// inference_network_env.py:295-444) on ONE shared signed-weight graph.
//
// The reference recomputes all single-flip gains with a dense [B,N,N] matmul every step.  Here the
// gains ("immediate cuts available", delta[b,i] = s_i * sum_j W_ij s_j) are a resident int32 [B,N]
// cache updated incrementally: flipping node a negates delta[a] and changes delta[j] of each
// neighbour j by 2 * W_ja * s_j * s_a'  -- O(deg) instead of O(N^2) -- the idea of
// rlsolver/methods/S2V_PPO/env.py:197-206.  The observable rows (state f32 [B,R,N]) are then written
// in the same kernel; they are O(N) per env by their definition (time-since-flip touches every
// node every step), which makes this kernel HBM-bound on (R + 2) * 4 * N bytes per env-step.
//
// One wave per env; lanes stride over nodes / over the action node's CSR row.
#include "rls_tile.h"

namespace rls {

struct SpinRows {  // row index of each observable inside state[b], or -1 when absent
    int immediate, time_since_flip, episode_time, termination, greedy, dist_score, dist_state;
};

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_spin_step(float* __restrict__ state, int64_t B, int64_t N, int R,
                                                   int32_t* __restrict__ delta,
                                                   const int32_t* __restrict__ rowptr,
                                                   const int32_t* __restrict__ col,
                                                   const int32_t* __restrict__ wgt,
                                                   const int64_t* __restrict__ action,
                                                   float* __restrict__ score, float* __restrict__ best_score,
                                                   float* __restrict__ best_spins, float* __restrict__ reward,
                                                   int32_t* __restrict__ num_nonpos, SpinRows rows,
                                                   float max_local, float time_inc, float termination_value,
                                                   int reward_mode, float reward_div) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
    if (b >= B) return;
    float* st = state + b * R * N;
    float* spins = st;  // row 0 = SPIN_STATE, signed {+1, -1}
    int32_t* dl = delta + b * N;
    const int64_t a = action[b];

    // 1. flip + score change (spinsystem_PECO.py:336-348): gain = delta[a] before the flip
    const float s_old = spins[a];
    const float s_new = -s_old;
    const int gain = dl[a];
    const int r0 = rowptr[a], r1 = rowptr[a + 1];
    // Neighbour updates are L2 atomics (multi-edges may hit one node twice in a wave-instruction);
    // everything that re-reads delta / spins below uses agent-scope (sc1) loads, which bypass this
    // CU's L1 and therefore see the atomics' results once vmcnt has drained.
    for (int j = r0 + lane; j < r1; j += kWave) {
        const int nb = col[j];
        const int w = wgt ? wgt[j] : 1;
        const int sj = spins[nb] > 0.0f ? 1 : -1;
        atomicAdd(&dl[nb], 2 * w * sj * (s_new > 0.0f ? 1 : -1));
    }
    if (lane == 0) {
        __hip_atomic_store(&dl[a], -gain, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&spins[a], s_new, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();

    // 2. reward w.r.t. the best observed score, best tracking (:366-401)
    const float sc = score[b] + (float)gain;
    const float best_before = best_score[b];
    const float improvement = sc - best_before;
    float rew;
    if (reward_mode == 1) rew = improvement > 0.0f ? improvement : 0.0f;                       // BLS
    else if (reward_mode == 2) rew = improvement > 0.0f ? improvement / (improvement + 0.1f) : 0.0f;  // CUSTOM_BLS
    else rew = (float)gain;                                                                     // DENSE
    rew = rew / reward_div;                                                                     // norm_rewards: rew /= n_spins
    const bool new_best = sc > best_before;
    const float best_now = new_best ? sc : best_before;

    // 3. observables (:412-451) + the O(N) pass: greedy count, distance to best state, best copy
    int nonpos = 0, hamming = 0;
    float* bs = best_spins + b * N;
    for (int64_t n = lane; n < N; n += kWave) {
        const int d = __hip_atomic_load(&dl[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nonpos += (d <= 0);
        const float s = __hip_atomic_load(&spins[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (new_best) bs[n] = s;
        else hamming += (bs[n] != s);
        if (rows.immediate >= 0) st[(int64_t)rows.immediate * N + n] = (float)d / max_local;
        if (rows.time_since_flip >= 0) {
            float* p = st + (int64_t)rows.time_since_flip * N + n;
            *p = (n == a) ? 0.0f : (*p + time_inc);
        }
        if (rows.episode_time >= 0) st[(int64_t)rows.episode_time * N + n] += time_inc;
        if (rows.termination >= 0) st[(int64_t)rows.termination * N + n] = termination_value;
    }
    nonpos = wave_sum_i32(nonpos);
    hamming = wave_sum_i32(hamming);
    const float greedy = 1.0f - (float)nonpos / (float)N;
    const float dscore = fabsf(sc - best_now) / max_local;
    for (int64_t n = lane; n < N; n += kWave) {
        if (rows.greedy >= 0) st[(int64_t)rows.greedy * N + n] = greedy;
        if (rows.dist_score >= 0) st[(int64_t)rows.dist_score * N + n] = dscore;
        if (rows.dist_state >= 0) st[(int64_t)rows.dist_state * N + n] = (float)hamming;
    }
    if (lane == 0) {
        score[b] = sc;
        best_score[b] = best_now;
        reward[b] = rew;
        num_nonpos[b] = nonpos;
    }
}

// reset helper: delta[b,i] = s_i * sum_j W_ij s_j from signed f32 spins (row 0 of state)
__global__ void k_spin_delta_init(const float* __restrict__ state, int64_t B, int64_t N, int R,
                                  const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                  const int32_t* __restrict__ wgt, int32_t* __restrict__ delta) {
    const int64_t total = B * N;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / N, i = t - b * N;
        const float* s = state + b * R * N;
        const int si = s[i] > 0.0f ? 1 : -1;
        int acc = 0;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) acc += (wgt ? wgt[j] : 1) * (s[col[j]] > 0.0f ? 1 : -1);
        delta[t] = si * acc;
    }
}

}  // namespace rls

using namespace rls;

extern "C" {

int rls_spin_delta_init(const rls_graph* g, const float* state, int64_t B, int32_t num_rows, int32_t* delta,
                        void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0 && num_rows >= 1, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(state && delta, RLS_EINVAL, "NULL pointer");
    hipLaunchKernelGGL(k_spin_delta_init, dim3(grid_for(B * g->num_nodes, 256)), dim3(256), 0, as_stream(stream), state,
                       B, g->num_nodes, num_rows, g->rowptr, g->col, g->wgt, delta);
    return check_launch("k_spin_delta_init");
}

int rls_spin_step(const rls_graph* g, float* state, int64_t B, int32_t num_rows, const int32_t* row_index,
                  int32_t* delta, const int64_t* action, float* score, float* best_score, float* best_spins,
                  float* reward, int32_t* num_nonpos, float max_local, float time_inc, float termination_value,
                  int32_t reward_mode, float reward_div, void* stream) {
    if (int rc = check_graph(g)) return rc;
    RLS_REQUIRE(B >= 0 && num_rows >= 1, RLS_EINVAL, "bad sizes");
    if (B == 0) return RLS_OK;
    RLS_REQUIRE(state && row_index && delta && action && score && best_score && best_spins && reward && num_nonpos,
                RLS_EINVAL, "NULL pointer");
    RLS_REQUIRE(reward_mode >= 0 && reward_mode <= 2, RLS_EINVAL, "reward_mode must be 0 (DENSE), 1 (BLS), 2 (CUSTOM_BLS)");
    RLS_REQUIRE(max_local != 0.0f && reward_div != 0.0f, RLS_EINVAL, "max_local_reward_available / reward_div is 0");
    SpinRows rows{row_index[0], row_index[1], row_index[2], row_index[3], row_index[4], row_index[5], row_index[6]};
    const int* ri = &rows.immediate;
    for (int k = 0; k < 7; ++k) RLS_REQUIRE(ri[k] < num_rows && ri[k] != 0, RLS_EINVAL, "row_index[%d]=%d out of range", k, ri[k]);
    hipLaunchKernelGGL(k_spin_step, dim3((unsigned)ceil_div(B, 4)), dim3(256), 0, as_stream(stream), state, B,
                       g->num_nodes, num_rows, delta, g->rowptr, g->col, g->wgt, action, score, best_score, best_spins,
                       reward, num_nonpos, rows, max_local, time_inc, termination_value, reward_mode, reward_div);
    return check_launch("k_spin_step");
}

}  // extern "C"
