/*
 * CPU ORACLE (plain C) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Restatement of the reference's MaxCut env path in the shape the reference computes it
 * (every candidate = a full objective evaluation over the stored edge list), used
 *   (1) by tests/ as a second, independent checker next to oracle/oracle_np.py, and
 *   (2) by bench.py's `cpu_baseline` leg ("port"), timed on the host cores with OpenMP.
 * Nothing under rlsolver_amd/ links, loads or calls this file.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks every function here against
 * golden vectors captured by importing the reference (tools/gen_golden.py).
 * Citations are relative to the reference root (Open-Finance-Lab/RLSolver).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* calculate_obj_values(xs, if_sum=True), rlsolver/envs/env_L2A.py:54-66:
 * values = xs[sim, n0_ids] ^ xs[sim, n1_ids]; sum(1); // 2 if bidirectional. */
static inline int64_t cut_of_row_u8(const uint8_t* row, const int32_t* eu, const int32_t* ev, int64_t E, int bidir) {
    int64_t s = 0;
    for (int64_t e = 0; e < E; ++e) s += (row[eu[e]] != 0) ^ (row[ev[e]] != 0);
    return bidir ? s / 2 : s;
}

static inline int64_t cut_of_row_f32(const float* row, const int32_t* eu, const int32_t* ev, int64_t E, int bidir) {
    int64_t s = 0;
    for (int64_t e = 0; e < E; ++e) s += (row[eu[e]] > 0.0f) ^ (row[ev[e]] > 0.0f); /* xs > 0, env_PPO.py:109 */
    return bidir ? s / 2 : s;
}

void orc_maxcut_obj(const uint8_t* xs, int64_t B, int64_t N, const int32_t* eu, const int32_t* ev, int64_t E,
                    int bidir, int64_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) out[b] = cut_of_row_u8(xs + b * N, eu, ev, E, bidir);
}

/* env_PPO.EnvMaxcut.step(action), rlsolver/envs/env_PPO.py:92-106: flip one node per env
 * (logical_not on the float32 state), recompute the whole cut, reward = cur - last. */
void orc_ppo_step(float* xs, int64_t B, int64_t N, const int64_t* action, const int32_t* eu, const int32_t* ev,
                  int64_t E, int bidir, float* last, float* reward, float* cur) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        float* row = xs + b * N;
        row[action[b]] = (row[action[b]] == 0.0f) ? 1.0f : 0.0f;
        const float c = (float)cut_of_row_f32(row, eu, ev, E, bidir);
        reward[b] = c - last[b];
        last[b] = c;
        cur[b] = c;
    }
}

/* same step on the 1-byte bool state (the L2A surface), for the byte-accounted headline */
void orc_step_u8(uint8_t* xs, int64_t B, int64_t N, const int64_t* action, const int32_t* eu, const int32_t* ev,
                 int64_t E, int bidir, int64_t* last, int64_t* reward) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        uint8_t* row = xs + b * N;
        row[action[b]] = !row[action[b]];
        const int64_t c = cut_of_row_u8(row, eu, ev, E, bidir);
        reward[b] = c - last[b];
        last[b] = c;
    }
}

/* the 'addition' loop of local_search_inplace, rlsolver/envs/env_L2A.py:109-116: for every node
 * flip it, re-evaluate the full objective, keep if not worse (update_xs_by_vs uses ge,
 * rlsolver/methods/util_read_data.py:199).  O(N * E') per env, as in the reference. */
void orc_greedy_sweep(uint8_t* xs, int64_t B, int64_t N, const int32_t* eu, const int32_t* ev, int64_t E,
                      int bidir, int64_t* vs) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        uint8_t* row = xs + b * N;
        for (int64_t i = 0; i < N; ++i) {
            row[i] = !row[i];
            const int64_t v1 = cut_of_row_u8(row, eu, ev, E, bidir);
            if (v1 >= vs[b]) vs[b] = v1;
            else row[i] = !row[i];
        }
    }
}

/* per-node cut degree over the env's stored adjacency, env_L2A.py:68-76 */
void orc_node_cutdeg(const uint8_t* xs, int64_t B, int64_t N, const int32_t* erowptr, const int32_t* ev,
                     int64_t* out) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const uint8_t* row = xs + b * N;
        for (int64_t i = 0; i < N; ++i) {
            int64_t c = 0;
            for (int32_t j = erowptr[i]; j < erowptr[i + 1]; ++j) c += (row[i] != 0) ^ (row[ev[j]] != 0);
            out[b * N + i] = c;
        }
    }
}

/* ISCO_TSP.calculate_distance, rlsolver/envs/env_ISCO.py:346-350 (float32 accumulation) */
void orc_tsp_tour_length(const float* dist, int64_t N, const int64_t* perm, int64_t B, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const int64_t* p = perm + b * N;
        float s = 0.0f;
        for (int64_t k = 0; k + 1 < N; ++k) s += dist[p[k] * N + p[k + 1]];
        s += dist[p[N - 1] * N + p[0]];
        out[b] = s;
    }
}
