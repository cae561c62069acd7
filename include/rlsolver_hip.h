/*
 * rlsolver_hip.h -- C ABI of the MI355X (gfx950) combinatorial-optimisation
 * environment engine.
 *
 * The reference (Open-Finance-Lab/RLSolver) has no FFI: its hot path is a set
 * of duck-typed Python classes that launch chains of ATen ops.  Each entry
 * point below replaces one such chain with ONE hand-written HIP kernel; the
 * comment on each names the reference interface it stands in for (file:line
 * relative to the reference root).  The Python classes in rlsolver_amd/ bind
 * these symbols with ctypes (see INTEGRATION.md for the stub a reference
 * maintainer would add).
 *
 * Conventions
 *   - every function returns RLS_OK (0) or a negative RLS_E* code and records a
 *     message retrievable with rls_last_error_string() (thread local);
 *   - no function allocates, frees or synchronises; all work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - every pointer is a DEVICE pointer owned by the caller unless marked
 *     [host]; sizes are int64_t; tensors are dense row-major;
 *   - "spins" x are one byte per node (torch.bool / uint8, values 0|1),
 *     env-major [B, N], unless the function takes `spin_bytes` (1 = uint8,
 *     4 = float32 holding 0.0f|1.0f, the env_PPO surface);
 *   - B = number of parallel environments ("sims"), N = nodes, E' = edges as
 *     stored by the env (E, or 2E when if_bidirectional).
 *   - graph size: the MaxCut kernels keep a bit tile of the spins in LDS -- 64 envs x N nodes (N <= 20 224) or, past
 *     that and wherever it measures faster, a half tile of 32 envs (N <= ~40 000; csrc/rls_tile32.h).  Larger graphs take
 *     one-env-per-wave forms (N <= 160 000): correct, an order of magnitude slower.  The choice is per launch and changes
 *     no result.
 */
#ifndef RLSOLVER_HIP_H
#define RLSOLVER_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLS_ABI_VERSION 12

enum {
    RLS_OK = 0,
    RLS_EINVAL = -1,      /* bad argument (null pointer, negative size, bad enum) */
    RLS_EUNSUPPORTED = -2,/* valid request outside what the kernels cover (e.g. N too large for LDS) */
    RLS_ELAUNCH = -3,     /* hipLaunchKernel / hipGetLastError reported a failure */
    RLS_ENODEVICE = -4    /* no HIP device visible */
};

/* One shared graph, resident in HBM/L2.  Host struct of device pointers; built
 * by the caller from rlsolver_amd.graph.GraphCSR.  Replaces the per-env index
 * tensors n0_ids/n1_ids/sim_ids [B,E'] int64 of envs/env_L2A.py:46-59 (24*B*E'
 * bytes there; (N+1+2E+2E')*4 bytes here, independent of B). */
typedef struct rls_graph {
    int64_t num_nodes;        /* N */
    int64_t num_stored_edges; /* E' */
    int64_t nnz;              /* 2E (symmetric CSR, self loops dropped) */
    int32_t if_bidirectional; /* result of objective kernels is count / 2 when set */
    int32_t max_degree;
    const int32_t* eu;        /* [E'] edge endpoints as stored, sorted by (eu, ev) */
    const int32_t* ev;        /* [E'] */
    const int32_t* erowptr;   /* [N+1] offsets of each node's run in eu/ev (the env's adjacency_indies) */
    const int32_t* rowptr;    /* [N+1] symmetric CSR */
    const int32_t* col;       /* [nnz] */
    const int32_t* wgt;       /* [nnz] integer edge weights or NULL (= all ones) */
    const int32_t* sweep_rowptr; /* [N+1] offsets into sweep_stream per schedule position, bit 31 = first position
                                  * of a batch (rls_graph_sweep_schedule), or NULL: single-wave sweep */
    const int32_t* sweep_stream; /* [nnz+N] per schedule position: node id, then its neighbours; or NULL */
    /* Lane-per-node slabs of the two adjacencies (rls_graph_ell), or NULL: for the 64-node group g, entries
     * ell_ptr[g] + 64 k + l = k-th neighbour of node 64 g + l, or that node itself past the end of its row. */
    const int32_t* ell_sym_ptr;  /* [ceil(N/64)+1]   symmetric CSR (rowptr/col): K3 */
    const int32_t* ell_sym;
    const int32_t* ell_st_ptr;   /* [ceil(N/64)+1]   adjacency as stored (erowptr/ev): K2, local-search weights */
    const int32_t* ell_st;
    /* Level-parallel form of the sweep schedule (rls_graph_sweep_levels), or NULL / 0 */
    const int32_t* sweep_lv_ptr; /* [num_sweep_groups+1] offset of each 64-lane group in sweep_lv_data, bit 31 = first group of a level, bit 30 = hub group */
    const int32_t* sweep_lv_data;
    int64_t num_sweep_groups;
} rls_graph;

int rls_version(void);
const char* rls_last_error_string(void);
/* Number of HIP devices visible (0 on a CPU-only host; never an error). */
int rls_device_count(void);

/* [host] Tuning table (ABI v11).  Which tile form / wave count / store policy a launch takes is decided per launch from
 * its shapes; every such choice can be FORCED by name ("RLS_K1_TILE32", "RLS_STEP_CHASE", ... -- rls_tuning_name
 * enumerates them; the "RLS_" prefix is optional) for A/B measurements and for the forced-form parity tests.  Forcing a
 * form never changes a result.  The production library reads no environment variable; only a -DRLS_DEV build seeds this
 * table from the environment once, when it is loaded.  The reference has no counterpart (it has no launch policies).
 * rls_tuning_unset(NULL) clears every entry.  Process-wide, not per stream. */
int rls_tuning_set(const char* name, int64_t value);
int rls_tuning_unset(const char* name);
int rls_tuning_get(const char* name, int64_t* value, int32_t* is_set);
int rls_tuning_name(int32_t index, const char** name);   /* index 0 .. until RLS_EINVAL */

/* [host] Level schedule of the sequential greedy sweep (envs/env_L2A.py:109-116 visits i = 0..N-1 and every
 * decision sees the flips of the earlier nodes).  level(i) = 1 + max level of i's lower-numbered neighbours:
 * nodes of one level are pairwise non-adjacent, all earlier neighbours of a node sit in lower levels and all
 * later ones in higher levels, so deciding level after level (any order inside a level) gives exactly the
 * sequential result.  Positions are nodes sorted by (level, id); a level is cut into batches of <= max_nodes
 * nodes and <= max_entries stream entries (what the kernels' LDS ring can hold).
 * rowptr/col: HOST symmetric CSR.  Outputs (host): rowptr_flagged [N+1] = offset of each position's run in
 * `stream`, bit 31 set on the first position of a batch; stream [nnz+N] = node id followed by its neighbours,
 * position after position.  Upload both and store the device pointers in rls_graph.sweep_rowptr /
 * sweep_stream. */
int rls_graph_sweep_schedule(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t max_nodes,
                             int32_t max_entries, int32_t* rowptr_flagged, int32_t* stream, int64_t* num_batches,
                             int64_t* num_levels);

/* [host] The level schedule of rls_graph_sweep_schedule in lane-per-node form: the nodes of a level, longest rows
 * first, in groups of 64 LANES.  A group costs its longest lane and the waves meet at every level boundary, so a row
 * too long for the level's cap (none / 32 / 16 / 8 entries per lane, chosen per level by an instruction-count
 * estimate; never more than 64) takes L = 2, 4 or 8 ADJACENT lanes, lane j of them holding neighbours j, j + L, ...;
 * the kernel adds the lanes' counters before the compare.  Group record in lv_data:
 *     64 words  node | (deg / 2) << 20 | log2 L << 28      (node = N on idle lanes)
 *     per block of 8 rounds, 2 slabs of [64 lanes][4 rounds] (ABI v11: a lane fetches a block in two 16-byte loads) --
 *     round r of lane l at 64 + 512 (r / 8) + 256 ((r / 4) % 2) + 4 l + r % 4:
 *               8 nb                                       (byte offset of the neighbour's word in the tile; the node's
 *                                                           own where the lane's share of the row has ended, 8 N on idle lanes)
 * with rounds = the group's longest lane rounded up to a multiple of 8; the table ends in sixteen spare rows (counted in
 * *total) so that a group's first two blocks can be read unguarded.  A wave decides a whole group at once on 64-env
 * words: bit-sliced count of differing neighbours, bit-sliced compare with deg/2, XOR of the flip mask into
 * the node's word -- bit-identical to the sequential pass (see rls_graph_sweep_schedule).
 * A row of 256 or more entries (a hub) is a group of its own, the first of its level, with lane = neighbour:
 *     64 words  [0] = node, [1] = degree, N elsewhere
 *     rounds (a multiple of 8) in the same block layout: 8 nb, neighbour e in round e / 64 of lane e % 64, the node's own
 *     offset past its end
 * lv_ptr [host, groups+1] (bit 31 = first group of a level, bit 30 = hub group); lv_data [host, capacity] or NULL to size
 * (*num_groups, *total).  Needs N < 2^20 and max degree < 4096. */
int rls_graph_sweep_levels(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t* lv_ptr, int64_t ptr_capacity,
                           int32_t* lv_data, int64_t data_capacity, int64_t* num_groups, int64_t* total);

/* [host] Lane-per-node ("ELL") slabs of a CSR adjacency for the bit-sliced per-node kernels: nodes in groups of
 * 64; group g holds max-degree-in-group rounds of 64 entries, every lane's column = the neighbours of its node in an
 * order chosen by the builder (the node itself where its row is shorter: x_i ^ x_i contributes nothing) -- each round's 32 words
 * of a half-wave are spread over the LDS bank classes n mod 32 as far as the rows allow; consumers must not rely on an order.  ell_ptr [host, G+1] with
 * G = ceil(N/64); ell [host, capacity] or NULL to only compute *total (= ell_ptr[G]) for sizing. */
int rls_graph_ell(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t* ell_ptr, int32_t* ell,
                  int64_t capacity, int64_t* total);

/* [host] Cut the node order 0..N-1 into maximal runs of pairwise NON-adjacent consecutive nodes
 * (<= max_nodes nodes and <= max_entries CSR entries each).  The greedy sweep may decide the nodes of
 * a run in parallel with results identical to the sequential pass of envs/env_L2A.py:109-116.
 * rowptr/col: HOST symmetric CSR; rowptr_flagged [host, N+1] = rowptr with bit 31 set on the first
 * node of each run.  (Used for visiting orders other than 0..N-1: the MCPG visit stream, methods/MCPG.py.) */
int rls_graph_sweep_batches(const int32_t* rowptr, const int32_t* col, int64_t N, int32_t max_nodes,
                            int32_t max_entries, int32_t* rowptr_flagged, int64_t* num_batches);


/* ------------------------------------------------------------------ MaxCut */

/* K1  EnvMaxcut.calculate_obj_values(xs, if_sum=True)  envs/env_L2A.py:54-66
 *     (= env_MCPG.py:54-66, env_PPO.py:108-121, env_k_spin.py:246-258).
 * obj[b] = #{(u,v) in stored edges : x[b,u] != x[b,v]}  (// 2 if bidirectional). */
int rls_maxcut_obj(const rls_graph* g, const void* x, int spin_bytes,
                   int64_t B, int64_t* obj, void* stream);

/* K1' calculate_obj_values(xs, if_sum=False): cutmask[b,e] = x[b,eu[e]] ^ x[b,ev[e]]
 *     as bool bytes [B,E'].  envs/env_L2A.py:61 */
int rls_maxcut_edge_cut_mask(const rls_graph* g, const uint8_t* x, int64_t B,
                             uint8_t* cutmask, void* stream);

/* K2  EnvMaxcut.calculate_obj_values_for_loop(xs, if_sum=False)  env_L2A.py:68-80
 * cutdeg[b,i] = #{j in adj(i) : x[b,j] != x[b,i]} where adj = the env's stored
 * adjacency (out-neighbours only when !if_bidirectional).  int64 [B,N]. */
int rls_maxcut_node_cutdeg(const rls_graph* g, const uint8_t* x, int64_t B,
                           int64_t* cutdeg, void* stream);

/* K3  all-node single-flip gain: delta[b,i] = obj(flip_i(x_b)) - obj(x_b)
 *     = sum_{j in N(i)} w_ij * (x_i == x_j ? +1 : -1), int32 [B,N].
 * Replaces autograd of the energy (envs/env_ISCO.py:51-63), dense
 * matmul(W,s)*s (ECO_S2V/src/envs/spinsystem_PECO.py:661) and the scatter_add
 * form (S2V_PPO/env.py:84-100). */
int rls_maxcut_delta_all(const rls_graph* g, const uint8_t* x, int64_t B,
                         int32_t* delta, void* stream);

/* K4  gym step: env_PPO.EnvMaxcut.step(action)  envs/env_PPO.py:92-106.
 * For every env b: a = action[b]; x_out[b,:] = x_in[b,:] with node a flipped;
 * d = cut gain of that flip; obj[b] += d (in/out, int32); reward[b] = (float)d;
 * cur[b] = (float)obj[b] if cur != NULL; done[b] = done_value if done != NULL.
 * x_out may equal x_in (in-place: only the flipped byte is written, O(deg)
 * traffic); otherwise the whole next state is emitted (2N + 20 B / env-step,
 * the headline byte accounting of SURVEY.md section 8d).  An action outside
 * [0, N) leaves its env untouched and yields reward = NaN (the reference raises). */
int rls_maxcut_step(const rls_graph* g, const void* x_in, void* x_out, int spin_bytes,
                    int64_t B, const int64_t* action, int32_t* obj,
                    float* reward, float* cur, float* done, float done_value,
                    void* stream);

/* K5  greedy single-flip sweep ("addition" loop)  envs/env_L2A.py:109-116,
 *     methods/LocalSearch.py:77-83:  for i in 0..N-1: flip i if the cut does
 *     not decrease (ties accept, update_xs_by_vs uses ge, util_read_data.py:199).
 * x [B,N] and obj [B] (int64) are updated in place.  PRECONDITION: obj[b] == cut(x[b]) on entry.  The accept rule itself works on
 * gains (flip i iff its gain >= 0) and does not read obj; what is written back differs by kernel form -- the level-parallel form
 * (the one every unweighted graph with a level schedule takes) counts the cut of the swept tile once and OVERWRITES obj with it,
 * the stream / generic forms ADD the accepted gains to the incoming obj.  The two agree exactly when the precondition holds, which
 * is how every caller on the path uses it (envs/env_L2A.py:91-92 computes good_vs first); a stale or offset obj gives a
 * form-dependent result and is a caller error.  One sequential O(E) pass per env instead of N full objective evaluations. */
int rls_maxcut_greedy_sweep(const rls_graph* g, uint8_t* x, int64_t B,
                            int64_t* obj, void* stream);

/* K6  one multi-flip proposal round of local_search_inplace  envs/env_L2A.py:102-107
 *     (= methods/LocalSearch.py:71-75):  xs = good_xs.clone(); xs[mask] = ~xs[mask];
 *     vs = calculate_obj_values(xs); update_xs_by_vs(good_xs, good_vs, xs, vs).
 * mask uint8 [B,N] is the caller's spin_rand.gt(thresh) (the noise and the
 * kthvalue threshold stay torch ops so they consume torch's generator exactly as
 * the reference does).  x/obj (int64) are updated in place: rows whose proposal
 * has cut >= obj[b] take the proposal.
 * mask_bits != 0: the mask is bit-packed, uint64 [ceil(B/64), N], bit e of word (t, n) = env 64 t + e at node n (bits of envs >= B
 * zero) -- the tile the kernel works on, N / 8 bytes per env instead of N (graphs within the 64-env tile, N <= 20 224, or the
 * half tile of 32 envs, N <= 40 448, which takes the matching dword of each word). */
int rls_maxcut_propose_accept(const rls_graph* g, uint8_t* x, int64_t B, const void* mask, int32_t mask_bits,
                              int64_t* obj, void* stream);

/* Pre-pass of the fused local search  envs/env_L2A.py:90-94 (methods/LocalSearch.py:64-65):
 *   ws[b,i] = n0_num_n1[i] - mult * cutdeg[b,i]   (stored adjacency; exact)
 *   ws_minmax[0][i] = min_b ws[b,i],  ws_minmax[1][i] = max_b ws[b,i]   (int32 [2][N], or NULL: not wanted)
 * mult = 1 for local_search_inplace in both env flavours (bidirectional: 2 * (cutdeg / 2)),
 * 2 for LocalSearch.random_search on a unidirectional env.  ws is ws_bytes = 1 | 2 | 4 bytes per entry (int8 /
 * int16 / int32 [B,N]): |ws| <= max(1, mult - 1) * max degree, and the fused kernel re-reads ws once per proposal
 * round, so the narrowest type that holds the graph's degrees is what keeps that kernel off the HBM roofline
 * (RLS_EINVAL when the range does not fit).  The reference's ws_std = max - min over the batch (:93) is
 * ws_minmax[1] - ws_minmax[0]: the tiles fold their extremes in with atomics that fire only where they improve the
 * table (the table is initialised here, also when B = 0). */
int rls_maxcut_ls_weights(const rls_graph* g, const uint8_t* x, int64_t B, int32_t mult, void* ws, int32_t ws_bytes,
                          int64_t ws_pitch, int32_t* ws_minmax, void* stream);

/* K2+K6+K5 fused: EnvMaxcut.local_search_inplace  envs/env_L2A.py:87-116 (first_draw_proposes = 0)
 * and the body of LocalSearch.random_search  methods/LocalSearch.py:53-83 (first_draw_proposes = 1)
 * in one kernel, the 64-env tile resident in LDS throughout:
 *   thresh[b] = kthvalue(ws[b,:] + noise[0,b,:] * rd_std, k = N - num_spin)
 *   for t in rounds: mask = (ws + noise[t] * rd_std) > thresh; proposal = x ^ mask;
 *                    rows whose proposal has cut >= obj take it                     (:98-107)
 *   greedy single-flip sweep (:109-116)
 * ws int8 / int16 [B,N] (ws_bytes = 1 | 2, as rls_maxcut_ls_weights wrote it) = the reference's
 * n0_num_n1 - k * cutdeg  (exact integer), rd_std f32 [N] = (max_b ws - min_b ws) * noise_std  -- a whole-batch
 * statistic, hence the pre-pass's ws_minmax.
 * noise f32 [num_iters + 1 - first_draw_proposes, B, N] = the randn_like draws in call order (test
 * mode: bit-exact against the reference) or NULL = in-kernel Philox + Box-Muller keyed by
 * (seed, env_offset + b, node, round).  obj int64 [B]: in/out, or out only when compute_obj != 0
 * (the reference's good_vs.shape == () case).  Unweighted graphs, max degree <= 512,
 * num_spin <= 15, rows of x / noise that start 4-byte aligned on 16-byte bases (N % 4 == 0).  ws is read in 16-byte pieces:
 * its rows sit ws_pitch ENTRIES apart (0 = N) and ws_pitch * ws_bytes must be a multiple of 16 on a 16-byte base -- the layout
 * rls_maxcut_ls_weights writes with a padded pitch -- so that no piece leaves its row; an exactly sized [B, N] array whose rows
 * are not 16-byte multiples is refused, never over-read.  RLS_EUNSUPPORTED
 * otherwise (callers take rls_maxcut_ls_threshold + rls_maxcut_ls_rounds, or K2 / K6 / K5). */
int rls_maxcut_local_search(const rls_graph* g, uint8_t* x, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                            const float* rd_std, const float* noise, uint64_t seed, int64_t env_offset, int32_t num_iters,
                            int32_t num_spin, int32_t first_draw_proposes, int64_t* obj, int32_t compute_obj,
                            void* stream);

/* 1 when rls_maxcut_local_search covers this graph / batch / num_spin (the same test its launcher applies: the
 * 8- or 4-wave LDS layout must fit 160 KB), else 0 -- callers then take the K2 + K6 + K5 path. */
int rls_maxcut_local_search_supported(const rls_graph* g, int64_t B, int32_t num_spin);

/* Which kernel family rls_maxcut_node_cutdeg / rls_maxcut_ls_weights (what = 0) or rls_maxcut_delta_all (what = 1) take for
 * a batch of B envs on this graph: 1 = bit-sliced, lane = node (unweighted graphs; a tile of 64 envs costs the same however
 * few it holds, so from the batch that pays for it), 2 = lane = env tile (weighted graphs, from a larger batch still),
 * 0 = element-parallel.  All three give the same result; tests use this to know which one they exercised. */
int rls_maxcut_node_stats_form(const rls_graph* g, int64_t B, int32_t what);

/* The same local search as separate launches, for graphs rls_maxcut_local_search does not cover (its LDS layout holds
 * two tiles and rd_std: N <= ~7100; these hold one tile: N <= ~15 000 -- and up to N = 39 936 on half tiles of 32 envs
 * (the bare 64-env tile for rows that are not 16-byte multiples, N <= 20 224), through the scratch buffer, which is
 * then REQUIRED; past ~24 900 nodes rd_std is read from global memory and N must be a multiple of 4).  Both use the fused kernel's
 * in-kernel draws -- normal(seed, env_offset + b, node, draw) -- so a caller that passes the same seed gets the result
 * the fused kernel would give.  They replace the torch ops of the decomposed path (randn_like, ws + noise * rd_std,
 * kthvalue, gt: envs/env_L2A.py:95-101, methods/LocalSearch.py:66-72) and K6's mask input.
 *   rls_maxcut_ls_threshold: thresh[b] (f32 [B]) = kthvalue(ws[b,:] + normal(draw) * rd_std, k = N - num_spin)
 *   rls_maxcut_ls_propose:   mask = ws + normal(draw) * rd_std > thresh[b]; rows of x whose x ^ mask has cut >= obj[b]
 *                            take it, obj[b] updated (update_xs_by_vs, util_read_data.py:199)
 * ws / ws_bytes / rd_std as for rls_maxcut_local_search; draw = 0 for the threshold, 1.. for the rounds
 * (0.. when the first draw proposes: LocalSearch.random_search).  ws_pitch = entries between the starts of two rows of ws
 * (0 = N; rls_maxcut_ls_weights writes that layout when given the same pitch): the rows must start 16-byte aligned, which a
 * pitch rounded up to 16 bytes gives for ANY N (the entries between N and the pitch are never read); x may have any N and
 * any alignment. */
int rls_maxcut_ls_threshold(const rls_graph* g, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch, const float* rd_std,
                            uint64_t seed, int64_t env_offset, int32_t draw, int32_t num_spin, float* thresh, void* scratch,
                            int64_t scratch_bytes, void* stream);
int rls_maxcut_ls_propose(const rls_graph* g, uint8_t* x, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                          const float* rd_std, const float* thresh, uint64_t seed, int64_t env_offset, int32_t draw, int64_t* obj,
                          void* scratch, int64_t scratch_bytes, void* stream);
/* scratch (16-byte aligned device memory, contents irrelevant, may be shared by all calls of one local search) lets a small
 * batch split each tile's noise pass -- the VALU-bound part -- over up to 8 workgroups: partial top-k lists / bit-packed mask
 * words go through it.  rls_maxcut_ls_scratch_bytes gives the size that enables this for (graph, B, ws_bytes) and num_draws
 * rounds' worth of mask words (1 for the per-round entry points; more than one: every round's, for any batch, while they stay under
 * 1 GiB); 0 = nothing to gain.  With scratch = NULL (or too small) both run one workgroup per tile; the results are the same either way. */
int64_t rls_maxcut_ls_scratch_bytes(const rls_graph* g, int64_t B, int32_t ws_bytes, int32_t num_draws);
/* num_draws rounds of rls_maxcut_ls_propose (draws first_draw, first_draw + 1, ...) in one call.  With scratch of
 * rls_maxcut_ls_scratch_bytes(g, B, ws_bytes, num_draws) bytes a small batch computes the mask words of all rounds first (they do
 * not depend on x) and applies them on ONE load of each tile; otherwise it is the loop over rls_maxcut_ls_propose. */
int rls_maxcut_ls_rounds(const rls_graph* g, uint8_t* x, int64_t B, const void* ws, int32_t ws_bytes, int64_t ws_pitch,
                         const float* rd_std, const float* thresh, uint64_t seed, int64_t env_offset, int32_t first_draw,
                         int32_t num_draws, int64_t* obj, void* scratch, int64_t scratch_bytes, void* stream);
/* The draws of the local-search kernels as a tensor: out f32 [B, N], out[b, n] = normal(seed, env_offset + b, n, draw) -- the
 * value rls_maxcut_local_search / _ls_threshold / _ls_propose use for that (env, node, draw).  Replaces torch.randn_like
 * (envs/env_L2A.py:95,99; methods/LocalSearch.py:66) on the decomposed path (weights wider than 16 bits), so that path too is
 * keyed by the global env id, and feeds the statistical tests of the generator. */
int rls_maxcut_ls_normals(float* out, int64_t B, int64_t N, uint64_t seed, int64_t env_offset, int32_t draw, void* stream);
/* Workgroups per tile the noise passes of a batch of B envs are split over when the scratch buffer is there (1 = the tiles fill the
 * chip by themselves). */
int rls_maxcut_ls_slices(const rls_graph* g, int64_t B, int32_t ws_bytes);
/* 1 when the two entry points above cover this graph / num_spin, else 0 (callers then keep the torch ops + K6). */
int rls_maxcut_ls_rounds_supported(const rls_graph* g, int32_t num_spin);

/* K10 update_xs_by_vs(xs0, vs0, xs1, vs1, if_maximize)  methods/util_read_data.py:190-202:
 *     rows of (xs1, vs1) that are >= (<= when !if_maximize) replace (xs0, vs0). */
int rls_select_better_rows(uint8_t* xs0, int64_t* vs0, const uint8_t* xs1, const int64_t* vs1,
                           int64_t B, int64_t N, int if_maximize, void* stream);

/* K10 pick_xs_by_vs(xs, vs, num_repeats, if_maximize)  methods/util_read_data.py:204-216:
 *     view xs as [R,S,N]; for each s take the r with the best vs (first on ties,
 *     torch.argmax semantics). */
int rls_pick_best_of_repeats(const uint8_t* xs, const int64_t* vs, int64_t R, int64_t S, int64_t N,
                             int if_maximize, uint8_t* good_xs, int64_t* good_vs, void* stream);

/* Evaluator.record2(i, vs, xs)  methods/util_evaluator.py:90-107 without the host read: good = first argmax
 * (argmin when !if_maximize) of vs [B] (vs_kind 0 = int64, 1 = float32, 2 = float64); if good is STRICTLY better
 * than best_v[0] (or force != 0: the constructor's first record), best_v[0] = good and best_x[:] = xs[good, :];
 * improved[0] = 1 | 0; log_v[log_index] = good when log_v != NULL.  best_v double[1], best_x uint8[N], improved
 * uint8[1], log_v double[*]: device memory the host reads only when it prints.  One workgroup. */
int rls_best_update(const uint8_t* xs, const void* vs, int vs_kind, int64_t B, int64_t N, int if_maximize,
                    uint8_t* best_x, double* best_v, uint8_t* improved, double* log_v, int64_t log_index, int force,
                    void* stream);

/* The key of the episode-boundary exchange between ranks ("one int64 key per rank,
 * (best_obj << 20) + (W - 1 - rank)", reduced with MAX = MAXLOC; the single-device analogue is best_vs.argmax(),
 * L2A/demo_instance.py:165) in one launch: key[0] = (max_b vs[b] << rank_bits) | low_code, index[0] (may be NULL) = the first
 * position of the maximum.  vs_kind 0 = int64, 3 = int32, 1 = float32, 2 = float64; float values must be integers or
 * half-integers (the bidirectional envs return count / 2) and are carried doubled.  flag[0] (int32, zeroed by the caller
 * once) gets bit 0 set when |value| >= limit or a float value is not a half-integer: the key is then built from 0. */
int rls_best_key(const void* vs, int vs_kind, int64_t B, int32_t rank_bits, int64_t low_code, int64_t limit, int64_t* key,
                 int64_t* index, int32_t* flag, void* stream);
/* ABI v12.  The reduced key back into what the caller wants, in one launch after the all-reduce (dist.py did it with three
 * [1]-sized torch ops): obj_out = key >> rank_bits as int64[1], or -- as_float -- double[1] = that / 2 (the doubled
 * half-integers of rls_best_key); owner_out (may be NULL) = world - 1 - (key & (2^rank_bits - 1)), the rank whose key won
 * (the `best_vs.argmax()` of L2A/demo_instance.py:165 across ranks).  flag (may be NULL): bit 1 set when key == empty_key,
 * the value a rank without envs contributes -- it survives the MAX only when no rank had any. */
int rls_key_unpack(const int64_t* key, int32_t rank_bits, int64_t world, int as_float, void* obj_out, int64_t* owner_out,
                   int64_t empty_key, int32_t* flag, void* stream);

/* ABI v12.  C2 of the episode-boundary exchange without a host read ("everyone restarts from the best", envs/env_MCPG.py:452-458:
 * best_xs[:] = best_xs[best_vs.argmax()], across ranks).  After the key's all-reduce every rank calls rls_winner_message: the rank
 * whose low code survived in the reduced key[0] writes msg = { (index[0] + env_offset) as 8 little-endian bytes | its row xs[index[0]]
 * bit-packed, bit k of byte j = x[8 j + k] }, every other rank (and a rank with B = 0) zeros -- msg uint8 [8 + ceil(N / 8)].  A
 * SUM all-reduce of msg is then the winner's message on every rank, and rls_winner_unpack turns it into x_out uint8 [N] (may be
 * NULL) and index_out int64[1] (may be NULL).  xs uint8 [B, N] 0|1; B == 1: the row itself (index only supplies the env id). */
int rls_winner_message(const uint8_t* xs, int64_t B, int64_t N, const int64_t* index, const int64_t* key, int32_t rank_bits,
                       int64_t my_low_code, int64_t env_offset, uint8_t* msg, void* stream);
int rls_winner_unpack(const uint8_t* msg, int64_t N, uint8_t* x_out, int64_t* index_out, void* stream);

/* K14 generate_xs_randomly(num_sims)  envs/env_L2A.py:82-85: i.i.d. Bernoulli(1/2)
 *     spins from a counter-based generator keyed by (seed, global env id), node 0
 *     forced to 0.  env_offset lets a rank generate its shard of a global batch. */
int rls_rand_spins(uint8_t* x, int64_t B, int64_t N, uint64_t seed, int64_t env_offset,
                   void* stream);
/* The same generator for R repeats of S envs in ONE launch (ABI v11): row r * S + s of x [R * S, N] is what
 * rls_rand_spins(seed = repeat_seeds[r], env_offset) writes for env s -- methods/LocalSearch.py:44-50 draws num_sims
 * batches of num_sims random rows in a Python loop.  repeat_seeds: DEVICE uint64 [R]. */
int rls_rand_spins_repeats(uint8_t* x, int64_t R, int64_t S, int64_t N, const uint64_t* repeat_seeds, int64_t env_offset,
                           void* stream);


/* uniform actions in [0, N) from the same generator (bench / MCMC proposals). */
int rls_rand_actions(int64_t* action, int64_t B, int64_t N, uint64_t seed, uint64_t step,
                     int64_t env_offset, void* stream);

/* ----------------------------------------------------- S2V / ECO / PECO spin system */

/* Resident per-env state of the spin-system env (all device pointers, owned by the caller).  T = float
 * (state_bytes = 4: the batched PECO env, spinsystem_PECO.py) or double (state_bytes = 8: the numpy env,
 * ECO_S2V/src/envs/spinsystem.py). */
typedef struct rls_spin_env {
    void* state;           /* T [B, R, N]: observable rows, row 0 = signed spins {+1,-1} */
    int32_t* delta;        /* [B, N] gain cache: delta[b,i] = s_i * sum_j W_ij s_j ("immediate cuts available",
                            * _get_immeditate_cuts_avaialable, spinsystem_PECO.py:660-661: a dense matmul there) */
    void* score;           /* T [B] */
    void* best_score;      /* T [B] */
    void* best_spins;      /* T [B, N] */
    int32_t* num_nonpos;   /* [B] #{i : delta[b,i] <= 0}, kept incrementally (greedy-actions row, basin test :392-397) */
    int32_t* dist_best;    /* [B] Hamming distance between the spins and best_spins, kept incrementally */
    /* bit-packed spins (needed by the two memories below), or NULL: W = ceil((N + allow_pass) / 64) words per state */
    uint64_t* packed;      /* [B, W] current spins, bit n % 64 of word n / 64 = (s_n > 0) */
    uint64_t* hash;        /* [B] Zobrist hash of packed (pre-filter of the exact compare) */
    /* visited-state memory behind stag_punishment / basin_reward (HistoryBuffer, util_envs_PECO.py:228-288,
     * util_envs.py:355-381), or NULL / 0 */
    uint64_t* hist;        /* [B, hist_cap, W] the state after each earlier step of the episode */
    uint64_t* hist_hash;   /* [B, hist_cap] */
    int64_t hist_cap;
    /* the rows a step changes EVERYWHERE are not stored between observations (a step is O(deg)): */
    int32_t* last_flip;    /* [B, N] the step (1-based) at which the node last flipped in this episode, 0 = not yet */
    void* scalars;         /* T [B, 4] current value of the rows TERMINATION_IMMANENCY, NUMBER_OF_GREEDY_ACTIONS_AVAILABLE,
                            * DISTANCE_FROM_BEST_SCORE, DISTANCE_FROM_BEST_STATE */
    const void* time_table;/* T [table_len]: time_table[k] = 0 + inc + inc + ... (k additions in T, inc = (T)(1 / max_steps)):
                            * what `state[TIME_SINCE_FLIP] += 1. / max_steps` (spinsystem_PECO.py:417) has accumulated after k steps */
    int64_t table_len;     /* >= max_steps + 1 */
    /* options of the single-instance env (spinsystem.py; the batched reference env cannot be constructed with them): */
    void* best_obs_score;  /* T [B] the best OBSERVABLE score: what rewards and the distance rows refer to (= best_score unless
                            * the memory is finite) */
    uint64_t* mem_spins;   /* [B, mem_len, ceil(N / 64)] finite memory (memory_length, spinsystem.py:398-404): the spins after each */
    void* mem_score;       /* T [B, mem_len]             of the last mem_len steps and their scores; NULL / 0 = infinite memory */
    int64_t mem_len;
    int64_t allow_pass;    /* ExtraAction.PASS (spinsystem.py:349-351): action N is legal and flips nothing; the packed spins then
                            * have ceil((N + 1) / 64) words per state, bit N = the parity of the passes (HistoryBuffer,
                            * util_envs.py:361-366, keys its states by the set of actions taken an odd number of times) */
} rls_spin_env;

/* State rows: row 0 (signed spins) and the IMMEDIATE_REWARD_AVAILABLE row of `state` are current after every call; the
 * other observable rows are written by rls_spin_reset, and after steps on demand: rls_spin_observation builds them
 * straight into the observation, rls_spin_materialize writes them into `state`.
 *
 * reset()  spinsystem_PECO.py:150-195 / spinsystem.py:176-252, after the caller has written the signed spins into
 * row 0 of state and their gains into delta (rls_maxcut_delta_all on the spins as bits: the two definitions
 * coincide).  Fills the IMMEDIATE_REWARD_AVAILABLE and NUMBER_OF_GREEDY_ACTIONS_AVAILABLE rows, zeroes the other
 * rows, score = best_score = cut = (weight_sum - sum_i delta_i) / 4 with weight_sum = sum of W over ORDERED pairs,
 * best_spins = spins, num_nonpos, dist_best = 0, last_flip = 0, scalars, packed / hash (the history starts empty: hist_len = 0).
 * row_index [host] int32[7]: as rls_spin_step. */
int rls_spin_reset(const rls_graph* g, const rls_spin_env* env, int state_bytes, int64_t B, int32_t num_rows,
                   const int32_t* row_index, double max_local, int64_t weight_sum, void* stream);

/* One env step of SpinSystemUnbiased  ECO_S2V/src/envs/spinsystem_PECO.py:306-486 (f32, batched) and
 * ECO_S2V/src/envs/spinsystem.py:333-482 (f64, B = 1) on a shared graph: flip action[b]; gain = delta[b,a];
 * incremental update of delta (O(deg), cf. S2V_PPO/env.py:197-206); score += gain; reward (mode 0 DENSE = gain,
 * 1 BLS = max(score - best_before, 0), 2 CUSTOM_BLS = impr / (impr + 0.1), 3 = impr / (impr + 0.05): CUSTOM_BLS for a caller
 * whose score is twice the one kept here -- OptimisationTarget.ENERGY, spinsystem.py:531-543, run as CUT on the negated
 * couplings: -E = 2 cut(-W) + const; with reward_div halved for modes 0 and 1 every reward is the reference's bit for bit), divided by
 * reward_div (n_spins under norm_rewards, else 1); visited-state test against the hist_len earlier states of the episode and append
 * (when env->packed): reward -= stag_punishment on a revisit (use_stag), reward += basin_reward on a first
 * visit of a state with no improving flip (use_basin); best_score / best_spins tracking; last_flip[b, a] = hist_len + 1
 * (hist_len = the number of steps already taken this episode, also without the visited-state memory) and the four
 * scalars.  row_index [host] int32[7] gives the row of IMMEDIATE_REWARD_AVAILABLE,
 * TIME_SINCE_FLIP, EPISODE_TIME, TERMINATION_IMMANENCY, NUMBER_OF_GREEDY_ACTIONS_AVAILABLE,
 * DISTANCE_FROM_BEST_SCORE, DISTANCE_FROM_BEST_STATE in state (-1 = not observed; row 0 is always the signed
 * spins).  reward T [B]; visited_new uint8 [B] or NULL (1 = state not seen before).  The scalar parameters are
 * rounded to T inside (pass the f32-rounded values for the f32 env).  An action outside [0, N) leaves its env
 * untouched and yields reward = NaN.  g->wgt = integer weights or NULL. */
int rls_spin_step(const rls_graph* g, const rls_spin_env* env, int state_bytes, int64_t B, int32_t num_rows,
                  const int32_t* row_index, const int64_t* action, void* reward, uint8_t* visited_new, double max_local,
                  double termination_value, int32_t reward_mode, double reward_div, int64_t hist_len,
                  int32_t use_stag, double stag_punishment, int32_t use_basin, double basin_reward, void* stream);

/* The same env with PER-ENV couplings: the training envs of spinsystem_PECO.py hold matrix T [B, N, N], redrawn by the graph
 * generator at every reset (:150-160; util_envs_PECO.py:40-56, 86-112), and recompute s * (W s) with a batched dense matmul
 * every step.  Here the flipped node's row of its env's matrix is the neighbour list of the same O(changed entries) update.
 *
 * rls_spin_reset_dense: after the caller has written the signed spins into row 0 of state: delta[b,i] = s_i sum_j W_ij s_j,
 * max_local T [B] = max_i sum_j W_ij (_get_immeditate_cuts_avaialable on all-ones spins, :162-168), weight_sum T [B] =
 * sum_ij W_ij, flags uint8 [B]: bit 0 = the reference draws the graph again (sum_i |sum_j W_ij| == 0 or max_local == 0, :164-169),
 * bit 2 = that maximum is 0 although rows with a nonzero (negative) sum exist: max_local then holds the maximum over the NONZERO
 * row sums, which is the rule of the single-instance numpy env (spinsystem.py:190-196: it goes on where the batched env draws again),
 * bit 1 = not a symmetric integer-valued matrix (the int32 gain cache cannot hold it; diagonal entries are allowed and count as
 * in the reference: flipping a changes the score by delta_a - 2 W_aa, :346-348); then everything
 * rls_spin_reset does, with the per-env max_local / weight_sum.  The caller reads flags before stepping. */
int rls_spin_reset_dense(const void* matrix, const rls_spin_env* env, int state_bytes, int64_t B, int64_t N, int32_t num_rows,
                         const int32_t* row_index, void* max_local, void* weight_sum, uint8_t* flags, void* stream);

/* rls_spin_step with matrix T [B, N, N] and max_local T [B] (as rls_spin_reset_dense left them) in place of the shared graph. */
int rls_spin_step_dense(const void* matrix, const void* max_local, const rls_spin_env* env, int state_bytes, int64_t B, int64_t N,
                        int32_t num_rows, const int32_t* row_index, const int64_t* action, void* reward, uint8_t* visited_new,
                        double termination_value, int32_t reward_mode, double reward_div, int64_t hist_len,
                        int32_t use_stag, double stag_punishment, int32_t use_basin, double basin_reward, void* stream);

/* The graph generators of the training envs  ECO_S2V/src/envs/util_envs_PECO.py:15-112 in one launch: matrix T [B, N, N]
 * (state_bytes 4 | 8), symmetric, entries 0 / +-1.
 *   kind 0  RandomERGraphGenerator (:42-57): every pair i < j is an edge with probability p_connection; zero diagonal.
 *   kind 1  RandomBAGraphGenerator (:84-113): seed clique on nodes 0..m INCLUDING its self-loops (as :93-95 sets them), each later
 *           node attached to m_insertion_edges distinct earlier nodes in proportion to their degree (torch.multinomial without
 *           replacement there; uniform draws over the edge-endpoint list with duplicates redrawn here: the same distribution).
 *   edge_type 1 UNIFORM (+1), 2 DISCRETE (one +-1 per node pair, shared by all envs of the call), 3 RANDOM (+-1 per pair and env).
 * Build-defined counter-based draws keyed by (seed, env_offset + b): distributionally equivalent to the reference's torch
 * streams, independent of how the batch is sharded.  N < 65536; kind 1 needs N * m * 6 B of LDS. */
int rls_rand_couplings(void* matrix, int state_bytes, int64_t B, int64_t N, int32_t kind, double p_connection, int32_t m_insertion_edges,
                       int32_t edge_type, uint64_t seed, int64_t env_offset, void* stream);

/* get_observation()  ECO_S2V/src/envs/spinsystem_PECO.py:455,497 (cat(state, matrix_obs)) and spinsystem.py:484-495
 * (vstack(state, matrix)): out T [B, num_rows + N, N] = the num_rows observable rows as the reference's state tensor holds
 * them after step_index steps of the episode (row 0 mapped from signed spins to (1 - s) / 2 when binary_basis,
 * SpinBasis.BINARY; the rows a step does not store built from last_flip / time_table / scalars), followed by the N
 * rows of the matrix: T [N, N] shared by all envs (matrix_per_env = 0) or T [B, N, N] (matrix_per_env = 1); matrix NULL:
 * out = [B, num_rows, N] (the rows only).  One streaming pass, nothing else touched. */
int rls_spin_observation(const rls_spin_env* env, const void* matrix, int32_t matrix_per_env, int state_bytes, int64_t B,
                         int32_t num_rows, int64_t N, const int32_t* row_index, int64_t step_index, int32_t binary_basis, void* out,
                         void* stream);

/* Writes the rows a step does not store (TIME_SINCE_FLIP, EPISODE_TIME, TERMINATION_IMMANENCY, NUMBER_OF_GREEDY_ACTIONS_
 * AVAILABLE, DISTANCE_FROM_BEST_SCORE, DISTANCE_FROM_BEST_STATE) into env->state as the reference holds them after
 * step_index steps: for callers that read the state tensor itself (spinsystem.py's `env.state`). */
int rls_spin_materialize(const rls_spin_env* env, int state_bytes, int64_t B, int64_t N, int32_t num_rows, const int32_t* row_index,
                         int64_t step_index, void* stream);

/* -------------------------------------------------------------------- MCPG */
/* Layouts of a batch of C chains:
 *   spin_bytes = 4 | 1   node-major x[N, C] as in the reference (chains are the fast axis): float32 0.0|1.0 (what
 *                        metro_sampling returns) or uint8;
 *   spin_bytes = 0       bit-packed, tile-major uint64 [ceil(C / 64), N]: word (t, n) holds node n of the chains
 *                        64 t .. 64 t + 63 (bit e = chain 64 t + e; bits of chains >= C are 0).  1/32 of the f32
 *                        surface's bytes, a 64-chain tile is N consecutive words; the form in which a sampling round
 *                        stays on the device (rls_mcpg_pack_chains / rls_mcpg_unpack_chains convert). */

/* [host struct] Global ids of a shard's chains ("per-env RNG keyed by global env id, not rank").  The in-kernel
 * generators of the three entry points below are keyed by (seed, GLOBAL chain id, round / pass / position); the global id of
 * local chain c is
 *     offset + c + (period > 0 ? (c / period) * skip : 0)
 * A rank that owns kept chains [m0, m0 + M_local) of a batch of M kept chains x R repeats (chain = repeat * M + kept, the
 * column order of the reference's  xs_bool.repeat(1, repeat_times), MCPG.py:393-394) passes {m0, M_local, M - M_local}: its
 * chains then draw exactly what they draw in the one-process run, whatever the rank count.  NULL = {0, 0, 0}, the
 * single-process numbering.  period > 0 (with skip > 0) must be a multiple of 64 that divides the launch's chain count C -- the
 * kernels then run one grid row per repeat instead of dividing on the device.  rls_mcpg_local_search_levels draws one coin WORD
 * per 64-chain tile: there offset and skip must be multiples of 64 as well (RLS_EINVAL otherwise).  Recorded draws (index / u /
 * uniforms / coins: test hooks) stay indexed by the LOCAL chain. */
typedef struct rls_chain_ids {
    int64_t offset;
    int64_t period;
    int64_t skip;
} rls_chain_ids;

/* K9  metro_sampling(probs, start_status, max_transfer_time)  methods/MCPG.py:88-117.
 * Runs rounds t = t_offset .. t_offset + min(T, *t_limit_dev) - 1 for every chain c:
 *   i = index[t,c]; p = x[i,c] ? probs[i] : 1 - probs[i];
 *   accept iff u[t,c] < (1 - p) / p  -> flip x[i,c];  accepts[r][t - t_offset] += #accepted chains (r = tile % accept_rows).
 * index int64 [*,C] and u f32 [*,C] are the reference's randint / rand draws in call order (rows
 * indexed by the absolute round t; test mode) or both NULL for the in-kernel counter-based generator
 * (murmur3 finaliser) keyed by (seed, global chain id, t) -- chain_ids [host] maps local chains to global ids (NULL: identity).
 * The reference stops after the first round whose cumulative accept count reaches C*T_transfer
 * samples_in (same dtype, may be NULL = samples) is where the chains are READ; they are written to samples:
 * the first chunk of a call turns the caller's start state into the result buffer without a copy.
 * (one host sync per round there).  Here the caller walks the 5*T_transfer rounds in chunks: a dry
 * call (write_back = 0) fills accepts [accept_rows][T] (int64, zeroed by the caller; a round's count is its COLUMN SUM --
 * workgroups spread their adds over the rows, one row would serialise thousands of atomics per round), the stop round is derived
 * on the device, a second call with t_limit_dev pointing at it (device int64) and write_back = 1
 * applies the chunk; *t_limit_dev <= 0 makes a call return immediately, so chunks after the stop
 * round cost one empty launch.  t_limit_dev NULL = all T rounds.  samples is updated in place only
 * when write_back != 0.
 * C_in (bit-packed layout only; 0 or C otherwise): samples_in holds C_in < C chains, C_in a multiple of 64, and chain c
 * starts from chain c % C_in -- the reference's  xs_bool = temp_max_info.repeat(1, repeat_times)  (MCPG.py:393-394)
 * without materialising the repeat. */
int rls_mcpg_metro_rounds(void* samples, const void* samples_in, int64_t C_in, int spin_bytes, int64_t N, int64_t C,
                          const float* probs, int64_t T, int64_t t_offset, const int64_t* index, const float* u,
                          uint64_t seed, const int64_t* t_limit_dev, int write_back, int64_t* accepts, int64_t accept_rows,
                          const rls_chain_ids* chain_ids, void* scratch, int64_t scratch_bytes, void* stream);
/* ABI v11: `scratch` (device, may be NULL) of at least rls_mcpg_metro_scratch_bytes(N, C) bytes lets the bit-packed walk keep its
 * draw windows (32 KB per 64-chain tile) in L2-resident global memory instead of LDS where that makes room for a second workgroup
 * per CU (N = 10^4: the tile alone is 80 KB); 0 from the query = the windows stay in LDS and scratch is ignored.  Contents are
 * undefined afterwards; results do not depend on it. */
int64_t rls_mcpg_metro_scratch_bytes(int64_t N, int64_t C);

/* The stop rule between two chunks of rls_mcpg_metro_rounds (MCPG.py:103,115: the reference compares `count` with
 * total_mcmc_num * max_transfer_time on the host after every round), one small launch instead of a chain of [T]-sized torch ops:
 * accepts [accept_rows][T] as the chunk's dry (or, for the first chunk, direct) pass filled it; ctl int64 [3] on the device =
 * {accepts of the earlier chunks, walk still live (0 / 1), limit of the next dry pass} -- written, and read unless first != 0;
 * *apply_limit (device int64, may be NULL) = min(this chunk's limit, the first round at which the running count reaches
 * target) for the apply pass; ctl[2] = next_T while the walk is live, else 0 (rls_mcpg_metro_rounds returns at once on 0).
 * first: 1 = the call's first chunk (applied directly: a round accepts at most C proposals, so the count cannot reach C * T
 * before round T), 2 = a later chunk inside the first T rounds, applied directly as well (ctl is read), 0 = a dry pass. */
int rls_mcpg_metro_stop(const int64_t* accepts, int64_t accept_rows, int64_t T, int64_t target, int32_t first, int64_t next_T,
                        int64_t* ctl, int64_t* apply_limit, void* stream);
/* Rounds one rls_mcpg_metro_rounds launch can take with accept counts (the node-major kernels keep them in LDS beside the
 * 64-chain tile: a G81-sized graph, N = 20 000, leaves room for 956); 0 = this layout does not fit at all for N. */
int64_t rls_mcpg_metro_max_rounds(int64_t N, int32_t spin_bytes);

/* K7 + K8 first half  sampler_func  methods/MCPG.py:128-152.
 * xs_in [N,C] holds 0|1 (the sampler's input before the reference maps it to -0.5|1.5).
 * For each of num_ls passes, for pos in 0..N-1, node = order[pos] (int32 [N], degree-descending):
 *     s = sum_{j in nbr(node)} value(j)      value = -0.5|1.5 for nodes not yet visited in
 *                                            pass 0, else 0|1 (MCPG.py:131-133,142)
 *     x[node,c] = (s + u * 0.25f) < (deg(node) + 0.25f) / 2
 * with u = uniforms[pass, pos, c] (f32 [num_ls,N,C], the torch.rand draws in visiting order; NULL =
 * in-kernel Philox / counter hash keyed by (seed, global chain id [chain_ids, host; NULL = identity], pass, position)).  Then expected[c] = sum_e (2x_u - 1)(2x_v - 1) over the stored edge list
 * (= E - 2*cut, exact in f32).  Outputs xs_out f32 [N,C] (0|1) and expected f32 [C].
 * visit_stream (optional, int32) is the visiting order flattened by the caller: visiting positions level-
 * scheduled (a position's level = 1 + max level of its earlier-visited neighbours; sorted by (level, position))
 * and cut into batches of pairwise non-adjacent nodes (<= 32 nodes and <= 400 entries per batch) whose
 * earlier-visited neighbours all sit in earlier batches:
 *     per batch: m, next_batch_offset, 0, off_0 .. off_{m-1}
 *     per node (at stream offset off_k): node, deg, nfresh, visiting position, then deg entries  nb | (fresh << 31)
 * where fresh marks a neighbour visited LATER than node (it still holds -0.5|1.5 in pass 0) and
 * nfresh counts them (rlsolver_amd.methods.MCPG.build_visit_stream builds it).  With it the kernel
 * streams the graph through an LDS ring, the waves of a workgroup share each batch, and nothing waits on
 * global memory per node; NULL selects the generic kernel (one CSR row fetch per node).
 * Weighted graphs (upstream MCPG's weighted MaxCut sampler, methods/MCPG/sampling.py:89-127): edge_weights int32 [E'] in
 * the order of g->eu / g->ev, and a visit stream whose node records are
 *     node, deg, Wfresh, visiting position, Wdeg, then deg PAIRS (nb | fresh << 31, weight)
 * (Wfresh = sum of the weights of the not-yet-visited neighbours, Wdeg = the weighted degree).  The test becomes
 * sum_j w_j v_j + u / 4 < Wdeg / 2 + 0.125 (:114-116) and expected = sum_e w_e (2x_u - 1)(2x_v - 1).  gauge_node >= 0
 * applies that sampler's gauge fix first: every chain is XORed with its own value at gauge_node (:101-104); -1: none.
 * Both need the visit stream (RLS_EUNSUPPORTED without it). */
int rls_mcpg_local_search(const rls_graph* g, const void* xs_in, int spin_bytes, float* xs_out, int64_t C,
                          const int32_t* order, const int32_t* visit_stream, int64_t visit_len, int64_t num_ls,
                          const float* uniforms, uint64_t seed, const int32_t* edge_weights, int64_t gauge_node,
                          float* expected, const rls_chain_ids* chain_ids, void* stream);

/* [host] Level-parallel form of the K7 visiting order (lane = node; same dependency-level argument as
 * rls_graph_sweep_levels, levels taken over visiting POSITIONS: order[pos] = node).  The nodes of a level with degree
 * <= 128 are packed into groups of 64 LANES, sorted by degree; a row too long for the level's cap (chosen per level by
 * an instruction-count estimate: none / 32 / 16 / 8 entries per lane, always <= 64) takes L = 2, 4 or 8 ADJACENT lanes,
 * lane j of them holding neighbours j, j + L, ... -- the kernel adds the lanes' counters before the compare, so a
 * level's longest row costs deg / L rounds and pads 63 lanes that much less.  A node of higher degree forms a group of
 * its own and is decided with lane = neighbour.  Group record in lv_data (ABI v11: LANE-major, so that a lane fetches its
 * header in one 8-byte load and a block of 8 rounds in two 16-byte loads):
 *     64 x 2 words  per lane l, words 2 l and 2 l + 1:
 *                   node | K << 20 | log2 L << 28 | tie << 31   K = ceil(deg / 2), tie = deg even               (passes >= 1)
 *                   pos  | K0 << 20 | tie0 << 31      K0 = ceil((deg + nfresh) / 2), tie0 = (deg + nfresh) even  (pass 0)
 *     per block of 8 rounds, 2 slabs of [64 lanes][4 rounds]: round r of lane l at 128 + 512 (r / 8) + 256 ((r / 4) % 2) + 4 l + r % 4:
 *                   8 nb | fresh << 31      (8 nb = byte offset of the neighbour's word in the tile); padding: 8 N
 * (rounds = the longest lane's, rounded up to a multiple of 8; idle lanes carry node = N; a hub group uses lane 0's two header
 * words -- K / K0 in 11 bits, no L --, word 4 = deg, and lists neighbour r in round r / 64 of lane r % 64).  Behind the last
 * record come sixteen spare rows (a group's header and first two blocks are requested unguarded), and behind those THE SAME TABLE
 * AGAIN with the fresh flags cleared, at offset (lv_ptr[groups] & 0x3fffffff) + 1024: passes >= 1 read it and use an entry as
 * the LDS address it is.  *total counts both copies.
 * The accept rule of MCPG.py:139-141, (s + u/4) < (deg + 1/4)/2 with s in half-integers, is  2s < deg, or
 * 2s == deg and u < 1/2 (in float32: see methods/MCPG.py tie_coins_from_uniforms):
 * new bit = [count < K] | ([count == K] & tie & coin), count = #ones among the neighbours
 * (pass 0: #ones among visited + 2 #ones among not-yet-visited ones, against K0).
 * lv_ptr [host, groups+1]: offset | bit 31 = first group of a level | bit 30 = hub group.  NULL outputs: sizing.
 * Needs N < 2^20 and max degree < 1024. */
int rls_mcpg_visit_levels(const int32_t* rowptr, const int32_t* col, int64_t N, const int32_t* order, int32_t* lv_ptr,
                          int64_t ptr_capacity, int32_t* lv_data, int64_t data_capacity, int64_t* num_groups,
                          int64_t* total);

/* K7 + K8 first half on that schedule (production path: tie coins from a counter hash keyed by (seed, GLOBAL 64-chain
 * block [chain_ids, host; NULL = identity], pass, position); coins uint64 [num_ls * N, ceil(C / 64)] -- bit c % 64 of word [pass * N + pos, c / 64] =
 * "u < 1/2" for chain c -- replaces them for tests).  Same outputs as rls_mcpg_local_search; xs_out is float32
 * node-major (out_spin_bytes = 4) or bit-packed (0; then xs_out may alias a bit-packed xs_in).  C_in: as in
 * rls_mcpg_metro_rounds. */
int rls_mcpg_local_search_levels(const rls_graph* g, const void* xs_in, int spin_bytes, int64_t C_in, void* xs_out,
                                 int out_spin_bytes, int64_t C, const int32_t* lv_ptr, const int32_t* lv_data,
                                 int64_t num_groups, int64_t num_ls, const uint64_t* coins, uint64_t seed, float* expected,
                                 const rls_chain_ids* chain_ids, void* stream);

/* 1 when rls_mcpg_local_search_levels covers this graph with a schedule of num_groups groups (bit tile + group
 * offsets + scratch within 160 KB of LDS, unweighted, N < 2^20, degrees < 1024), else 0. */
int rls_mcpg_local_search_levels_supported(const rls_graph* g, int64_t num_groups);

/* K8 second half  methods/MCPG.py:154-161: best_index[m] = m + M * argmin_r expected[r*M + m]
 * (first minimum), vs_good[m] = (num_edges - expected[best]) / 2, xs_good[:, m] = xs[:, best].
 * M = total_mcmc_num, R = repeat_times, xs f32 [N, M*R], xs_good f32 [N, M]. */
int rls_mcpg_pick_best(const float* expected, const void* xs, int spin_bytes, int64_t N, int64_t total_mcmc_num,
                       int64_t repeat_times, int64_t num_edges, int64_t* best_index, float* vs_good,
                       void* xs_good, void* stream);

/* The best-merge of the MCPG outer loop  methods/MCPG.py:376-391 on bit-packed kept chains (M = total_mcmc_num,
 * temp_info / now_info uint64 [ceil(M/64), N]), all on the device:
 *   for m: if temp_max[m] > now_max_res[m]: now_max_res[m] = temp_max[m]; now_info[:, m] = temp_info[:, m]   (:377-380)
 *   hi = first argmax, lo = first argmin of now_max_res; now_max_res[lo] = now_max_res[hi];
 *   now_info[:, lo] = now_info[:, hi]; temp_info[:, lo] = now_info[:, hi]                                     (:383-391)
 * temp_info afterwards is the start state of the next round (rls_mcpg_metro_rounds with C_in = M).  mask_scratch
 * uint64 [ceil(M/64)]; best_value f32 [1] / best_index int64 [1] (may be NULL) receive max(now_max_res) and its chain.
 * replace_worst = 0 (a rank's shard of the kept chains: the worst and the best incumbent are properties of the WHOLE batch):
 * only the per-chain merge (:377-380) is applied, and best_value / best_index are [2] = {max, min} of the shard's now_max_res
 * and their first chains -- what the caller needs to replace the global worst by the global best itself. */
int rls_mcpg_merge_best(const float* temp_max, uint64_t* temp_info, float* now_max_res, uint64_t* now_info, int64_t N,
                        int64_t total_mcmc_num, uint64_t* mask_scratch, float* best_value, int64_t* best_index,
                        int32_t replace_worst, void* stream);

/* What get_return  methods/MCPG.py:292-302 needs from the samples: A[n] += sum_c value[c] * s[n, c] over the C
 * bit-packed chains (A f32 [N], zeroed by the caller).  With s in {0,1},  log(s p + (1-s)(1-p)) summed over nodes is
 * sum_n log(1-p_n) + sum_n s_n (log p_n - log(1-p_n)), so the objective mean_c(log_prob_sum_c * value_c) and its
 * gradient with respect to p follow from A and sum(value) -- the [C, N] float products are never formed. */
int rls_mcpg_value_bit_sums(const uint64_t* samples, int64_t N, int64_t C, const float* value, float* A, void* stream);

/* node-major uint8 / float32 [N, C] -> bit-packed tiles, and back to float32 (the shims of the reference-shaped API) */
int rls_mcpg_pack_chains(const void* xs, int spin_bytes, int64_t N, int64_t C, uint64_t* packed, void* stream);
int rls_mcpg_unpack_chains(const uint64_t* packed, int64_t N, int64_t C, float* xs, void* stream);

/* K11 mcpg_sampling_qubo / mcpg_sampling_qubo_bin  methods/MCPG/sampling.py:323-370 (after the
 * metro step): num_ls Gauss-Seidel sweeps  x_i <- [Q[i,:] . x (x_i := 0) > thr_i]  over variables in
 * index order, then value[c] = x^T Q x.  binary = 0: spins +-1, thr = 0 (:332-340); binary = 1:
 * spins 0|1, thr = -Q_ii / 2 (:357-365).  Q f32 [n,n] dense; xs_in/xs_out f32 [n,C] node-major holding
 * 0|1 (the +-1 variant maps 0 -> -1 internally and returns (s+1)/2 as the reference does).
 * Exact for integer-valued Q (f32 sums order-independent below 2^24). */
int rls_qubo_local_search_value(const float* Q, int64_t n, const float* xs_in, float* xs_out, int64_t C,
                                int64_t num_ls, int binary, float* value, void* stream);

/* K11, sparse form: the same coordinate search and value with Q in CSR form -- rowptr int32
 * [n+1], col int32 [nnz], val f32 [nnz], the diagonal stored as ordinary entries; step i costs O(nnz_i), not O(n).
 * Same arguments and outputs otherwise; same exactness condition (integer-valued Q).
 * ABI v12: lv_ptr int32 [num_levels + 1] / lv_rows int32 [n] (both NULL = none): the rows grouped by level(i) = 1 + max level of
 * the neighbours j < i (rows in ascending order within a level) -- rows of one level share no entry, so the waves of a
 * workgroup sweep them side by side and the result is the sequential sweep's; without them one wave walks the rows in order. */
int rls_qubo_sparse_local_search_value(const int32_t* rowptr, const int32_t* col, const float* val, int64_t n,
                                       const int32_t* lv_ptr, const int32_t* lv_rows, int32_t num_levels,
                                       const float* xs_in, float* xs_out, int64_t C, int64_t num_ls, int binary,
                                       float* value, void* stream);

/* --------------------------------------------------------------------- TSP */

/* K12 ISCO_TSP.calculate_distance(sample)  envs/env_ISCO.py:346-350.
 * len[b] = sum_k D[p[k], p[k+1]] + D[p[N-1], p[0]], f32, sequential k order. */
int rls_tsp_tour_length(const float* dist, int64_t N, const int64_t* perm, int64_t B,
                        float* length, void* stream);

/* K13 ISCO_TSP.opt_2(sample, temperature)  envs/env_ISCO.py:238-335 (ABI v12: the whole of it).
 * For every position i of every tour: the partner CITY (:245-262: rand < K / (K + 1) ? nearest[city_i, randint(K)] :
 * random[city_i, randint(N - K - 1)]), j = its position in perm[b] (:265-272), ban = the partner is a neighbour of position
 * i + 1 (:291-293), delta of swapping the cities at positions i + 1 and j (:318-333).  Outputs logratio = -delta /
 * temperature f32 [B,N], indices int64 [B,N], ban uint8 [B,N].
 *   selected == NULL (production): the partners are drawn IN the kernel from (seed, env_offset + b, i) -- the generator and the
 *     counters of rls_isco_tsp_step's iteration 0 -- through nearest int32 [N, K] / random int32 [N, random_stride]
 *     (random_stride >= N - K - 1; or their byte form tables8, below), near_threshold = K / (K + 1); selected_out (may be
 *     NULL) records them, int64 [B,N].
 *     Algorithmic bytes per tour: 8N in, 13N out.
 *   selected != NULL (the recorded-draw hook of the golden tests): int64 [B,N] partner cities, tables / seed ignored. */
int rls_tsp_swap_delta_all(const float* dist, int64_t N, const int64_t* perm, int64_t B, const int64_t* selected,
                           const int32_t* nearest, int32_t K, const int32_t* random, int32_t random_stride, const uint8_t* tables8,
                           float near_threshold, uint64_t seed, int64_t env_offset, int64_t* selected_out, float temperature,
                           float* logratio, int64_t* indices, uint8_t* ban, void* stream);
/* tables8 (may be NULL; N <= 256): the same two tables as BYTES for the kernel to keep in LDS beside the distance matrix --
 * rls_tsp_tables8_bytes(N, K) bytes, 4-byte aligned: uint8 [N, K] nearest, zero-padded to a multiple of 16 bytes, then uint8
 * [N, N - K - 1] = the first N - K - 1 columns of random, zero-padded likewise (0 = no byte form for this N / K). */
int64_t rls_tsp_tables8_bytes(int64_t N, int32_t K);

/* ISCO_TSP.switch  envs/env_ISCO.py:337-344 for one chosen position per env:
 * if pos[b] >= 0 swap perm[b, (pos[b]+1) % N] and perm[b, indices[b, pos[b]]]. */
int rls_tsp_apply_swap(int64_t* perm, int64_t B, int64_t N, const int64_t* pos,
                       const int64_t* indices, void* stream);

/* true 2-opt (segment reversal) delta for a closed tour, the move of
 * methods_problem_specific/TSP/opt_2.py:27-57: reversing perm[i..j] changes
 * the length by D[p[i-1],p[j]] + D[p[i],p[j+1]] - D[p[i-1],p[i]] - D[p[j],p[j+1]].
 * i, j int64 [B] with 0 <= i <= j < N. */
int rls_tsp_2opt_delta(const float* dist, int64_t N, const int64_t* perm, int64_t B,
                       const int64_t* i, const int64_t* j, float* delta, void* stream);

/* One pass of local_search_2_opt  methods_problem_specific/TSP/opt_2.py:27-57: over all reversals [i..j], 0 <= i < j <= N-1,
 * of the closed tour perm[b] (evaluated against that same tour, as the reference restores its seed after every candidate)
 * the best one, the first in (i, j) order among equals.  dist float64 [N, N].
 *   cur_length != NULL (float64 [B], the length the reference holds for the seed): candidates are compared by their WHOLE
 *     length summed the way distance_calc does (float64, edge after edge) -- the reference's own comparison values, so
 *     reversals that tie in exact arithmetic rank as they do there.  best_value[b] = that length (< cur_length[b]) and
 *     (best_i, best_j)[b] the pair, or cur_length[b] and (-1, -1) when no candidate is shorter.  O(N) per candidate.
 *   cur_length == NULL: candidates are compared by delta(i, j) = D[a,c] + D[b,e] - D[a,b] - D[c,e] (rls_tsp_2opt_delta's
 *     formula in float64; SYMMETRIC dist), O(1) per candidate: best_value[b] = the most negative delta, or 0 / (-1, -1).
 * perm rows must be permutations of 0..N-1 (not checked on the device).
 * slices >= 1 workgroups share the candidates of one tour (a lone tour still fills the chip): best_i, best_j and best_value
 * must then hold slices * B entries; the results are the first B of each (the rest is scratch). */
int rls_tsp_2opt_best(const double* dist, int64_t N, const int64_t* perm, int64_t B, const double* cur_length, int32_t slices,
                      int64_t* best_i, int64_t* best_j, double* best_value, void* stream);

/* ------------------------------------------------------------------- ISCO sampler steps */

/* ISCO_maxcut.step(x, path_length, temperature)  envs/env_ISCO.py:26-49 in ONE kernel (one wave per env):
 *   ll_x = cut(x) / T;  log_prob = log_softmax(gain / 2T)   (get_local_dist :51-63: the closed form of the autograd)
 *   Gumbel top-k without replacement of path_length[b] nodes  (multinomial, methods/util.py:514-555) -> mask, ll_x2y
 *   y = x with the masked nodes flipped; ll_y, log_prob(y); ll_y2x = probability of undoing the selection in
 *   reverse order (ll_y2x :65-77);  log_acc = min(ll_y + ll_y2x - ll_x - ll_x2y, 0);
 *   y_out[b] = accepted ? y : x  with  log(u_accept + 1e-24) < log_acc   (mh_step, methods/util.py:556-570)
 * x, y_out f32 [B, N] holding 0|1 (y_out may alias x); path_length int64 [B], clamped to [1, N] as the caller's
 * torch.clamp does (methods/ISCO/main_ISCO_maxcut.py:26).  u_gumbel f32 [B, N] and u_accept f32 [B]: the
 * reference's two torch.rand draws in call order (tests), or both NULL: counter-based generator keyed by
 * (seed, env_offset + b, node).  Outputs (each may be NULL): energy_out f32 [B] = ll_y * T (of the PROPOSAL, as
 * the reference returns it), acc_out f32 [B] = exp(log_acc), terms_out f32 [B, 5] = ll_x, ll_x2y, ll_y, ll_y2x,
 * log_acc, mask_out uint8 [B, N] = the selected nodes.  Unweighted (the reference's energy ignores weights).
 * scratch: device memory of rls_isco_maxcut_scratch_bytes(g, B) bytes -- 0 (scratch may be NULL) while a sample's rows fit LDS
 * (N <= ~15 900); past that the two f32 rows of every sample (log-probabilities, perturbed values: 8 N bytes) live there and a
 * workgroup takes each sample (N <= ~81 000: the two byte rows still sit in LDS; RLS_EUNSUPPORTED beyond).  Contents need not
 * survive between calls. */
int64_t rls_isco_maxcut_scratch_bytes(const rls_graph* g, int64_t B);
int rls_isco_maxcut_step(const rls_graph* g, const float* x, float* y_out, int64_t B, const int64_t* path_length,
                         float temperature, const float* u_gumbel, const float* u_accept, uint64_t seed,
                         int64_t env_offset, float* energy_out, float* acc_out, float* terms_out, uint8_t* mask_out,
                         void* scratch, int64_t scratch_bytes, void* stream);

/* ISCO_TSP.step(x, path_length, temperature)  envs/env_ISCO.py:188-236 in ONE kernel (one wave per env, the tour,
 * its inverse and -- when it fits -- the distance matrix in LDS): path_length times { opt_2 (:238-335: partner
 * city per position from the nearest / random tables, swap delta, ban mask) -> logits = -delta / 2T (banned:
 * -5e5) -> log_softmax -> one Gumbel draw -> selected position q; ll_x2y = log_prob[q]; ll_y2x = log_softmax of
 * the logits with entry q negated, at q (y2x :214-226); swap positions q + 1 and j(q) unless banned (switch
 * :337-344) }, then log_acc = min(sum of the delta_yx, -ll_x2y, ll_y2x terms, 0) and the Metropolis accept
 * between perm_in and the walked tour.  nearest int32 [N, K], random int32 [N, random_stride] of which columns 0 .. N-K-2 are drawn from (the
 * reference's table has N - 1 columns, ISCO/util_TSP.py:9-16),
 * near_threshold = (float)(K / (K + 1)) as the reference's comparison sees it.  Test draws, all or none:
 * u_partner f32, r_near int64, r_rand int64, u_gumbel f32, each [path_length, B, N] in call order, u_accept f32
 * [B].  perm_out int64 [B, N] (must not alias perm_in); log_acc_out / acc_out f32 [B] and cur_out int64 [B, N]
 * (the walked tour before the accept) may be NULL. */
int rls_isco_tsp_step(const float* dist, int64_t N, const int32_t* nearest, int32_t K, float near_threshold,
                      const int32_t* random, int32_t random_stride, const int64_t* perm_in, int64_t* perm_out, int64_t B, int32_t path_length,
                      float temperature, const float* u_partner, const int64_t* r_near, const int64_t* r_rand,
                      const float* u_gumbel, const float* u_accept, uint64_t seed, int64_t env_offset,
                      float* log_acc_out, float* acc_out, int64_t* cur_out, void* stream);

/* The row moves of evolutionary_replacement  methods/util.py:87-94:  xs[dst[k]] = xs[src[k]], vs[dst[k]] =
 * vs[src[k]] for k < K (dst and src disjoint: top_ids vs low_ids there).  xs uint8 [*, N], vs int64 or NULL. */
int rls_copy_rows(uint8_t* xs, int64_t* vs, int64_t N, const int64_t* dst, const int64_t* src, int64_t K, void* stream);

/* B random permutations (random_gen_init_sample, env_ISCO.py:352-354). */
int rls_rand_perms(int64_t* perm, int64_t B, int64_t N, uint64_t seed, int64_t env_offset,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RLSOLVER_HIP_H */
