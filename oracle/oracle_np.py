"""CPU ORACLE (numpy) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement of the reference's algorithms for the hot path, written the way the
reference computes them (full objective re-evaluation per candidate, Python loops over
nodes), so that it is an independent check of the restructured HIP kernels.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; nothing under ``rlsolver_amd/`` does.

Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
against golden vectors produced by importing the reference itself in the build
container (tools/gen_golden.py -> tests/golden/*.npz).  The reference has no tests or
golden vectors of its own.

All paths in the citations are relative to the reference root (Open-Finance-Lab/RLSolver).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------- graph forms


def stored_edges(graph: np.ndarray, if_bidirectional: bool):
    """Edge endpoint lists exactly as EnvMaxcut.__init__ builds them
    (rlsolver/envs/env_L2A.py:40-49 via util_read_data.py:144-187): per-n0 neighbour lists,
    each sorted by n1, concatenated in n0 order; both directions when bidirectional."""
    g = np.asarray(graph, dtype=np.int64).reshape(-1, 3)
    u, v = g[:, 0], g[:, 1]
    if if_bidirectional:
        u, v = np.concatenate([u, v]), np.concatenate([v, u])
    order = np.lexsort((v, u))
    return u[order], v[order]


def num_nodes_distinct(graph: np.ndarray) -> int:
    """calc_num_nodes_in_mygraph, rlsolver/methods/util.py:35-40."""
    g = np.asarray(graph, dtype=np.int64).reshape(-1, 3)
    return int(np.unique(g[:, :2]).shape[0])


def adjacency_lists(graph, num_nodes, if_bidirectional):
    """adjacency_indies: List[sorted neighbour array], env_L2A.py:40-44."""
    u, v = stored_edges(graph, if_bidirectional)
    return [v[u == i] for i in range(num_nodes)]


# --------------------------------------------------------------------------- MaxCut objective


def maxcut_obj(xs, graph, if_bidirectional):
    """calculate_obj_values(xs, if_sum=True), env_L2A.py:54-66:
    values = xs[:, n0_ids] ^ xs[:, n1_ids]; sum(1); // 2 if bidirectional.  int64 [B]."""
    xs = np.asarray(xs).astype(bool)
    u, v = stored_edges(graph, if_bidirectional)
    vals = (xs[:, u] ^ xs[:, v]).sum(axis=1).astype(np.int64)
    return vals // 2 if if_bidirectional else vals


def maxcut_edge_mask(xs, graph, if_bidirectional):
    """calculate_obj_values(xs, if_sum=False), env_L2A.py:61: bool [B, E'].  (The reference
    then applies ``// 2`` to the bool tensor when bidirectional, which zeroes it; callers only
    use if_sum=False with the unidirectional env, and so do we.)"""
    xs = np.asarray(xs).astype(bool)
    u, v = stored_edges(graph, if_bidirectional)
    return xs[:, u] ^ xs[:, v]


def maxcut_node_cutdeg(xs, graph, num_nodes, if_bidirectional):
    """calculate_obj_values_for_loop(xs, if_sum=False) before the bidirectional ``/ 2``,
    env_L2A.py:68-76: values[:, node0] = (xs[:, node0, None] ^ xs[:, node1s]).sum(1).  int64 [B,N]."""
    xs = np.asarray(xs).astype(bool)
    adj = adjacency_lists(graph, num_nodes, if_bidirectional)
    out = np.zeros((xs.shape[0], num_nodes), dtype=np.int64)
    for n0 in range(num_nodes):
        n1s = adj[n0]
        if n1s.shape[0] > 0:
            out[:, n0] = (xs[:, n0, None] ^ xs[:, n1s]).sum(axis=1)
    return out


def maxcut_obj_for_loop(xs, graph, num_nodes, if_bidirectional, if_sum=True):
    """calculate_obj_values_for_loop, env_L2A.py:68-80 including its dtype quirk: int64 when
    unidirectional, float32 (values / 2) when bidirectional."""
    raw = maxcut_node_cutdeg(xs, graph, num_nodes, if_bidirectional)
    vals = raw.sum(axis=1) if if_sum else raw
    if if_bidirectional:
        return vals.astype(np.float32) / np.float32(2)
    return vals


def maxcut_delta_all(xs, graph, num_nodes, weights=None):
    """Definition used for K3: delta[b,i] = obj(flip_i(x_b)) - obj(x_b), computed literally
    (flip, re-evaluate) with the (optionally weighted) cut; int64 [B,N]."""
    xs = np.asarray(xs).astype(bool)
    g = np.asarray(graph, dtype=np.int64).reshape(-1, 3)
    u, v = g[:, 0], g[:, 1]
    w = np.ones(len(u), np.int64) if weights is None else np.asarray(weights, np.int64)

    def cut(x):
        return ((x[:, u] ^ x[:, v]) * w).sum(axis=1)

    base = cut(xs)
    out = np.zeros((xs.shape[0], num_nodes), dtype=np.int64)
    for i in range(num_nodes):
        x1 = xs.copy()
        x1[:, i] = ~x1[:, i]
        out[:, i] = cut(x1) - base
    return out


# --------------------------------------------------------------------------- select ops


def update_xs_by_vs(xs0, vs0, xs1, vs1, if_maximize=True):
    """rlsolver/methods/util_read_data.py:190-202 (in place; returns B, sic)."""
    good = vs1 >= vs0 if if_maximize else vs1 <= vs0
    xs0[good] = xs1[good]
    vs0[good] = vs1[good]
    return good.shape[0]


def pick_xs_by_vs(xs, vs, num_repeats, if_maximize=True):
    """rlsolver/methods/util_read_data.py:204-216."""
    n = xs.shape[1]
    s = xs.shape[0] // num_repeats
    xv = xs.reshape(num_repeats, s, n)
    vv = vs.reshape(num_repeats, s)
    ids = vv.argmax(axis=0) if if_maximize else vv.argmin(axis=0)
    sid = np.arange(s)
    return xv[ids, sid], vv[ids, sid]


def evolutionary_replacement(xs, vs, low_k, perm, if_maximize=True):
    """rlsolver/methods/util.py:87-94 with the randperm supplied (in place)."""
    ids = np.argsort(vs, kind="stable")
    top_ids, low_ids = (ids[:-low_k], ids[-low_k:]) if if_maximize else (ids[:low_k], ids[low_k:])
    replace_ids = top_ids[perm[:low_k]]
    xs[replace_ids] = xs[low_ids]
    vs[replace_ids] = vs[low_ids]


# --------------------------------------------------------------------------- local search


def greedy_sweep(xs, vs, graph, if_bidirectional):
    """The 'addition' loop of local_search_inplace, env_L2A.py:109-116, as written there:
    for every node clone, flip the column, re-evaluate the full objective, keep rows that are
    not worse.  O(N * E * B).  In place; returns (xs, vs)."""
    n = xs.shape[1]
    for i in range(n):
        xs1 = xs.copy()
        xs1[:, i] = ~xs1[:, i]
        vs1 = maxcut_obj(xs1, graph, if_bidirectional)
        update_xs_by_vs(xs, vs, xs1, vs1, True)
    return xs, vs


def local_search_inplace(xs, graph, num_nodes, if_bidirectional, noise, num_iters=8, num_spin=8,
                         noise_std=0.3, good_vs=None):
    """EnvMaxcut.local_search_inplace, env_L2A.py:87-116, with the randn_like draws supplied
    as ``noise`` f32 [num_iters + 1, B, N] in call order (the first one only sets ``thresh``).
    Arithmetic follows torch's type promotion: int64 ws + f32 tensor -> f32."""
    xs = np.asarray(xs).astype(bool)
    vs_raw = maxcut_node_cutdeg(xs, graph, num_nodes, if_bidirectional)
    if if_bidirectional:  # calculate_obj_values_for_loop returns float / 2 (env_L2A.py:78-79)
        vs_raw_f = vs_raw.astype(np.float32) / np.float32(2)
    else:
        vs_raw_f = vs_raw
    good_vs = vs_raw_f.sum(axis=1).astype(np.int64) if good_vs is None else np.asarray(good_vs, np.int64)
    u, v = stored_edges(graph, if_bidirectional)
    n0_num_n1 = np.bincount(u, minlength=num_nodes).astype(np.int64)[None, :]
    ws = n0_num_n1 - (2 if if_bidirectional else 1) * vs_raw_f        # int64, or f32 when bidirectional
    ws_std = ws.max(axis=0, keepdims=True) - ws.min(axis=0, keepdims=True)
    rd_std = ws_std.astype(np.float32) * np.float32(noise_std)
    ws_f = ws.astype(np.float32)
    spin_rand = ws_f + noise[0].astype(np.float32) * rd_std
    k = num_nodes - num_spin
    thresh = np.partition(spin_rand, k - 1, axis=1)[:, k - 1][:, None]  # kthvalue = k-th smallest
    for it in range(num_iters):
        spin_rand = ws_f + noise[1 + it].astype(np.float32) * rd_std
        mask = spin_rand > thresh
        x1 = xs.copy()
        x1[mask] = ~x1[mask]
        v1 = maxcut_obj(x1, graph, if_bidirectional)
        update_xs_by_vs(xs, good_vs, x1, v1, True)
    greedy_sweep(xs, good_vs, graph, if_bidirectional)
    return xs, good_vs


def local_search_class_random_search(good_xs, good_vs, graph, num_nodes, noise, num_iters, num_spin,
                                     noise_std=0.3):
    """LocalSearch.random_search, rlsolver/methods/LocalSearch.py:53-86 (unidirectional env only;
    the reference raises for if_bidirectional=True).  noise f32 [num_iters, B, N]."""
    if_bidirectional = False
    kth = num_nodes - num_spin
    prev_xs = good_xs.copy()
    prev_vs_raw = maxcut_node_cutdeg(prev_xs, graph, num_nodes, if_bidirectional)
    prev_vs = prev_vs_raw.sum(axis=1)
    u, _ = stored_edges(graph, if_bidirectional)
    n0_num_n1 = np.bincount(u, minlength=num_nodes).astype(np.int64)[None, :]
    thresh = None
    for it in range(num_iters):
        ws = n0_num_n1 - 2 * prev_vs_raw
        ws_std = ws.max(axis=0, keepdims=True) - ws.min(axis=0, keepdims=True)
        spin_rand = ws.astype(np.float32) + noise[it].astype(np.float32) * (ws_std.astype(np.float32) * np.float32(noise_std))
        if thresh is None:
            thresh = np.partition(spin_rand, kth - 1, axis=1)[:, kth - 1][:, None]
        mask = spin_rand > thresh
        xs = prev_xs.copy()
        xs[mask] = ~xs[mask]
        vs = maxcut_obj(xs, graph, if_bidirectional)
        update_xs_by_vs(prev_xs, prev_vs, xs, vs, True)
    greedy_sweep(prev_xs, prev_vs, graph, if_bidirectional)
    num_update = update_xs_by_vs(good_xs, good_vs, prev_xs, prev_vs, True)
    return good_xs, good_vs, num_update


# --------------------------------------------------------------------------- gym step


class PPOEnvOracle:
    """env_PPO.EnvMaxcut, rlsolver/envs/env_PPO.py:63-126: xs kept as float32 0/1, step() flips
    one node per env, recomputes the whole cut, reward = cur - last, done every num_steps."""

    def __init__(self, graph, num_nodes, num_steps, if_bidirectional):
        self.graph, self.n, self.num_steps, self.bidir = graph, num_nodes, num_steps, if_bidirectional
        self.action_count = 0
        self.xs = None
        self.last = None

    def reset_to(self, xs_bool):
        self.xs = np.asarray(xs_bool).astype(np.float32)
        self.last = maxcut_obj(self.xs > 0, self.graph, self.bidir).astype(np.float32)
        return self.xs

    def step(self, action):
        self.action_count += 1
        b = np.arange(self.xs.shape[0])
        self.xs[b, action] = np.logical_not(self.xs[b, action]).astype(np.float32)
        cur = maxcut_obj(self.xs > 0, self.graph, self.bidir).astype(np.float32)
        reward = cur - self.last
        self.last = cur
        if self.action_count == self.num_steps:
            self.action_count = 0
            done = np.ones(self.xs.shape[0], np.float32)
        else:
            done = np.zeros(self.xs.shape[0], np.float32)
        return self.xs, reward, done, cur


# --------------------------------------------------------------------------- counter-based RNG (build-defined)


def philox4x32_10(key0, key1, c0, c1, c2, c3):
    """Philox-4x32-10 (Salmon et al. 2011), the generator the HIP kernels use for K14.  Not a
    reference algorithm (the reference calls torch.randint); restated so the kernels' output is
    checkable bit for bit.  Vectorised over numpy uint64 arrays holding 32-bit values."""
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    W0, W1 = np.uint64(0x9E3779B9), np.uint64(0xBB67AE85)
    mask = np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & mask for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(key0) & mask, np.uint64(key1) & mask
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        h0, l0 = p0 >> np.uint64(32), p0 & mask
        h1, l1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (h1 ^ c1 ^ k0) & mask, l1, (h0 ^ c3 ^ k1) & mask, l0
        k0 = (k0 + W0) & mask
        k1 = (k1 + W1) & mask
    return c0, c1, c2, c3


def rand_spins(B, N, seed, env_offset=0):
    """K14 definition: spin(b, n) = bit (n & 127) of Philox(seed; ctr = (gb_lo, gb_hi, n >> 7, 'SPIN')),
    gb = env_offset + b; node 0 forced to 0 (generate_xs_randomly, env_L2A.py:82-85)."""
    gb = (np.arange(B, dtype=np.uint64) + np.uint64(env_offset))[:, None]
    blk = np.arange((N + 127) // 128, dtype=np.uint64)[None, :]
    r = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, gb & np.uint64(0xFFFFFFFF), gb >> np.uint64(32),
                      blk + np.zeros_like(gb), np.uint64(0x5350494E))
    words = np.stack(r, axis=-1).astype(np.uint32)                      # [B, blocks, 4]
    bits = ((words[..., None] >> np.arange(32, dtype=np.uint32)) & 1).astype(np.uint8)  # [B, blocks, 4, 32]
    xs = bits.reshape(B, -1)[:, :N].copy()
    xs[:, 0] = 0
    return xs


def rand_actions(B, N, seed, step, env_offset=0):
    gb = np.arange(B, dtype=np.uint64) + np.uint64(env_offset)
    r0 = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, gb & np.uint64(0xFFFFFFFF), gb >> np.uint64(32),
                       np.uint64(step & 0xFFFFFFFF) + np.zeros_like(gb),
                       np.uint64(((step >> 32) & 0xFFFFFFFF) ^ 0x41435431))[0]
    return ((r0 * np.uint64(N)) >> np.uint64(32)).astype(np.int64)


def _coupling_signs(edge_type, seed, gb, lo, hi):
    """+-1 per (env gb, pair lo <= hi) as rls_rand_couplings draws it: UNIFORM (1) +1; DISCRETE (2) Philox(seed; lo, hi, 0,
    'SIGN') shared by the envs; RANDOM (3) Philox(seed; gb_lo, gb_hi, lo * 65536 + hi, 'SIGO'); bit 0 set -> +1."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    gb, lo, hi = np.broadcast_arrays(np.asarray(gb, np.uint64), np.asarray(lo, np.uint64), np.asarray(hi, np.uint64))
    if edge_type == 1:
        return np.ones(gb.shape)
    if edge_type == 2:
        r0 = philox4x32_10(k0, k1, lo, hi, np.zeros_like(lo), np.uint64(0x5349474E))[0]
    else:
        r0 = philox4x32_10(k0, k1, gb & np.uint64(0xFFFFFFFF), gb >> np.uint64(32), lo * np.uint64(65536) + hi, np.uint64(0x5349474F))[0]
    return np.where(r0 & np.uint64(1), 1.0, -1.0)


def rand_couplings_er(B, N, p_connection, edge_type, seed, env_offset=0):
    """rls_rand_couplings kind 0 (the distribution of RandomERGraphGenerator, util_envs_PECO.py:42-57): pair i < j of env gb is
    an edge iff Philox(seed; gb_lo, gb_hi, i * 65536 + j, 'ERGP')[0] < floor(p * 2^32)."""
    gb = (np.arange(B, dtype=np.uint64) + np.uint64(env_offset))[:, None, None]
    i, j = np.meshgrid(np.arange(N, dtype=np.uint64), np.arange(N, dtype=np.uint64), indexing="ij")
    lo, hi = np.minimum(i, j)[None], np.maximum(i, j)[None]
    r0 = philox4x32_10(seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, gb & np.uint64(0xFFFFFFFF) + 0 * lo, (gb >> np.uint64(32)) + 0 * lo,
                       lo * np.uint64(65536) + hi + 0 * gb, np.uint64(0x45524750))[0]
    thr = min(int(p_connection * 4294967296.0), 0xFFFFFFFF)
    edge = ((r0 < np.uint64(thr)) | (p_connection >= 1.0)) & (lo != hi)
    return np.where(edge, _coupling_signs(edge_type, seed, gb, lo, hi), 0.0)


def rand_couplings_ba(B, N, m, edge_type, seed, env_offset=0):
    """rls_rand_couplings kind 1 (the distribution of RandomBAGraphGenerator, util_envs_PECO.py:84-113): clique on 0..m with
    self-loops; node v > m draws list positions idx = (r * L) >> 32, L = (m+1)^2 + 2m(v-m-1), with the 32-bit draws
    r = Philox(seed; gb_lo, gb_hi, v * 4096 + a // 4, 'BAGR')[a % 4], a = 0, 1, ...; position idx is clique node idx // (m+1)
    below (m+1)^2, else endpoint rr = q % 2m of node vv = m+1 + q // 2m (q = idx - (m+1)^2): target[vv][rr] if rr < m else vv;
    duplicates are drawn again until m distinct targets are found."""
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    out = np.zeros((B, N, N))
    L0 = (m + 1) * (m + 1)
    for b in range(B):
        gb = b + env_offset
        tgt = np.zeros((N, m), dtype=np.int64)
        for v in range(m + 1, N):
            L = L0 + 2 * m * (v - m - 1)
            chosen, a = [], 0
            while len(chosen) < m:
                if a % 4 == 0:
                    r = [int(x) for x in philox4x32_10(k0, k1, gb & 0xFFFFFFFF, gb >> 32, v * 4096 + a // 4, 0x42414752)]
                idx = (r[a % 4] * L) >> 32
                a += 1
                if idx < L0:
                    node = idx // (m + 1)
                else:
                    q = idx - L0
                    vv, rr = m + 1 + q // (2 * m), q % (2 * m)
                    node = int(tgt[vv, rr]) if rr < m else vv
                if node not in chosen:
                    chosen.append(node)
            tgt[v] = chosen
        pairs = [(i, j) for i in range(m + 1) for j in range(m + 1)] + [(int(tgt[v, r]), v) for v in range(m + 1, N) for r in range(m)]
        for i, j in pairs:
            w = float(_coupling_signs(edge_type, seed, gb, min(i, j), max(i, j)))
            out[b, i, j] = out[b, j, i] = w
    return out


# --------------------------------------------------------------------------- MCPG


def metro_sampling(probs, start_status, max_transfer_time, index, u):
    """metro_sampling, rlsolver/methods/MCPG.py:88-117, with the randint/rand draws supplied
    (index int64 [T', C], u f32 [T', C] in call order).  Returns float32 0/1 [N, C]."""
    probs = np.asarray(probs, np.float32)
    samples = np.asarray(start_status).astype(bool).copy()
    num_chain = samples.shape[1]
    col = np.arange(num_chain)
    count = 0
    t_used = 0
    for t in range(max_transfer_time * 5):
        if count >= num_chain * max_transfer_time:
            break
        row = index[t]
        base = probs[row]
        val = samples[row, col]
        chosen = np.where(val, base, np.float32(1) - base).astype(np.float32)
        accept_rate = (np.float32(1) - chosen) / chosen
        is_accept = u[t].astype(np.float32) < accept_rate
        samples[row, col] = np.where(is_accept, ~val, val)
        count += int(is_accept.sum())
        t_used += 1
    return samples.astype(np.float32), t_used


def mcpg_neighbors(edge_index, num_nodes):
    """append_neighbors, MCPG.py:235-252: neighbour order = order of appearance in the edge list
    (both endpoints appended per edge)."""
    nb = [[] for _ in range(num_nodes)]
    for a, b in zip(edge_index[0], edge_index[1]):
        nb[int(a)].append(int(b))
        nb[int(b)].append(int(a))
    return [np.asarray(x, dtype=np.int64) for x in nb]


def sampler_func(edge_index, num_nodes, sorted_degree_nodes, xs_sample, num_ls, total_mcmc_num,
                 repeat_times, uniforms):
    """sampler_func, rlsolver/methods/MCPG.py:120-166 with torch.rand draws supplied as
    uniforms f32 [num_ls, N(visit order), C].  float32 arithmetic throughout."""
    k = np.float32(1 / 4)
    nb = mcpg_neighbors(edge_index, num_nodes)
    wdeg = [np.float32(len(x)) for x in nb]
    num_edges = edge_index.shape[1]
    x = xs_sample.astype(np.float32).copy()
    x *= np.float32(2)
    x -= np.float32(0.5)
    for cnt in range(num_ls):
        for pos, node in enumerate(sorted_degree_nodes):
            node = int(node)
            s = x[nb[node]].sum(axis=0, dtype=np.float32) if len(nb[node]) else np.zeros(x.shape[1], np.float32)
            rv = s + uniforms[cnt, pos].astype(np.float32) * k
            x[node] = (rv < (wdeg[node] + k) / np.float32(2)).astype(np.float32)
    C = total_mcmc_num * repeat_times
    expected = np.empty(C, np.float32)
    n0, n1 = edge_index[0], edge_index[1]
    for j in range(repeat_times):
        j0, j1 = total_mcmc_num * j, total_mcmc_num * (j + 1)
        a = np.float32(2) * x[n0, j0:j1] - np.float32(1)
        b = np.float32(2) * x[n1, j0:j1] - np.float32(1)
        expected[j0:j1] = (a * b).sum(axis=0, dtype=np.float32)
    er = expected.reshape(-1, total_mcmc_num)
    index = er.argmin(axis=0)
    index = np.arange(total_mcmc_num) + index * total_mcmc_num
    max_cut = expected[index]
    vs_good = (np.float32(num_edges) - max_cut) / np.float32(2)
    xs_good = x[:, index]
    value = expected.astype(np.float32).copy()
    value -= value.mean(dtype=np.float32)
    return vs_good, xs_good, value, x, expected


# --------------------------------------------------------------------------- TSP


def tsp_tour_length(distance, perms):
    """ISCO_TSP.calculate_distance, rlsolver/envs/env_ISCO.py:346-350 (float32)."""
    d = np.asarray(distance, np.float32)
    p = np.asarray(perms, np.int64)
    tot = d[p[:, :-1], p[:, 1:]].sum(axis=1, dtype=np.float32)
    tot = tot + d[p[:, -1], p[:, 0]]
    return tot.astype(np.float32)


def tsp_tour_length_f64(distance, perms):
    """distance_calc of methods_problem_specific/TSP/util.py:13-18 / opt_2.py:17-22 on the closed
    tour (f64 accumulation, sequential)."""
    d = np.asarray(distance, np.float64)
    out = []
    for p in np.asarray(perms, np.int64):
        t = list(p) + [p[0]]
        s = 0.0
        for k in range(len(t) - 1):
            s = s + d[t[k], t[k + 1]]
        out.append(s)
    return np.asarray(out)


def tsp_selected_partner(perms, nearest_indices, random_indices, rand, randint_nearest, randint_random, K):
    """First half of ISCO_TSP.opt_2, env_ISCO.py:246-266: the partner CITY drawn for each position."""
    p = np.asarray(perms, np.int64)
    cond = rand < np.float32(K / (K + 1))
    near = np.take_along_axis(nearest_indices[p], randint_nearest[..., None], axis=2)[..., 0]
    rnd = np.take_along_axis(random_indices[p], randint_random[..., None], axis=2)[..., 0]
    return np.where(cond, near, rnd)


def tsp_swap_delta_all(distance, perms, selected, temperature):
    """Second half of ISCO_TSP.opt_2, env_ISCO.py:268-335: position of the partner, ban mask,
    3-case delta; returns (-delta / T, indices, ban)."""
    d = np.asarray(distance, np.float32)
    p = np.asarray(perms, np.int64)
    B, N = p.shape
    inv = np.empty_like(p)
    inv[np.arange(B)[:, None], p] = np.arange(N)[None, :]        # sort + searchsorted == inverse perm
    indices = np.take_along_axis(inv, selected, axis=1)
    mask = np.broadcast_to(np.arange(N)[None, :], (B, N))
    mask0, mask1, mask2 = (mask - 1) % N, (mask + 1) % N, (mask + 2) % N
    ind0, ind1 = (indices - 1) % N, (indices + 1) % N
    g = lambda idx: np.take_along_axis(p, idx, axis=1)
    s_m1, s_m0 = g(mask1), g(mask0)
    c1, c2 = s_m1 == selected, s_m0 == selected
    ban = c1 | c2
    s_i0, s_i1, s_i = g(ind0), g(ind1), g(indices)
    c3 = s_m1 == s_i0
    nm, nm1, nm2 = g(mask), s_m1, g(mask2)
    D = lambda a, b: d[a, b]
    case3 = -(D(nm, nm1) + D(s_i, s_i1)) + (D(nm, s_i) + D(s_i0, s_i1))
    case4 = -(D(nm, nm1) + D(nm1, nm2) + D(s_i0, s_i) + D(s_i, s_i1)) + \
        (D(nm, s_i) + D(s_i, nm2) + D(s_i0, nm1) + D(nm1, s_i1))
    delta = np.where(ban, np.float32(0), np.where(c3, case3, case4)).astype(np.float32)
    return (-delta / np.float32(temperature)).astype(np.float32), indices, ban


def tsp_switch(perms, pos, indices):
    """ISCO_TSP.switch, env_ISCO.py:337-344 for one position per env (pos < 0 = no swap)."""
    x = np.asarray(perms, np.int64).copy()
    N = x.shape[1]
    for b in range(x.shape[0]):
        if pos[b] < 0:
            continue
        j = indices[b, pos[b]]
        a = (pos[b] + 1) % N
        x[b, a], x[b, j] = x[b, j], x[b, a]
    return x


def tsp_2opt_delta(distance, perms, env, i, j):
    """True 2-opt move of methods_problem_specific/TSP/opt_2.py:40-45: reverse tour[i..j] of the
    closed tour and re-evaluate with distance_calc; returns new - old (f64)."""
    d = np.asarray(distance, np.float64)
    out = []
    for b, a, c in zip(env, i, j):
        p = list(np.asarray(perms[b], np.int64))
        t = p + [p[0]]
        base = sum(d[t[k], t[k + 1]] for k in range(len(t) - 1))
        t2 = list(t)
        t2[a:c + 1] = list(reversed(t2[a:c + 1]))
        t2[-1] = t2[0]
        new = sum(d[t2[k], t2[k + 1]] for k in range(len(t2) - 1))
        out.append(new - base)
    return np.asarray(out)


# --------------------------------------------------------------------------- base-64 solution strings

BASE_DIGITS = "0123456789ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz_$"


def b64_str_to_bool(x_str: str, encode_len: int) -> np.ndarray:
    """EncoderBase64.str_to_bool, rlsolver/methods/util_evaluator.py:51-65."""
    s = x_str.replace("\n", "").replace(" ", "")
    x_int = 0
    for ch in s:
        x_int = x_int * 64 + BASE_DIGITS.index(ch)
    x_bin = bin(x_int)[2:]
    out = np.zeros(encode_len, dtype=bool)
    out[-len(x_bin):] = [c == "1" for c in x_bin]
    return out


def b64_bool_to_str(x_bool, encode_len: int) -> str:
    """EncoderBase64.bool_to_str, rlsolver/methods/util_evaluator.py:32-49."""
    string_len = -int(-(encode_len / 6) // 1)
    x_int = int("".join("1" if i else "0" for i in list(x_bool)), 2)
    x_str = ""
    while True:
        x_str = BASE_DIGITS[x_int % 64] + x_str
        x_int //= 64
        if x_int == 0:
            break
    if len(x_str) > 120:
        x_str = "\n".join(x_str[i:i + 120] for i in range(0, len(x_str), 120))
    if len(x_str) > 64:
        x_str = f"\n{x_str}"
    return x_str.zfill(string_len)


def _murmur_mix32(h):
    """murmur3's 32-bit finaliser on uint64 arrays holding 32-bit values (csrc/rls_draw.h: isco_mix)."""
    M = np.uint64(0xFFFFFFFF)
    h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x85EBCA6B)) & M
    h = h ^ (h >> np.uint64(13)); h = (h * np.uint64(0xC2B2AE35)) & M
    return h ^ (h >> np.uint64(16))


def rand_perms(B, N, seed, env_offset=0):
    """K14 (TSP) definition (round 6): Fisher-Yates with counter-based murmur draws -- the generator of the ISCO kernels
    (csrc/rls_draw.h), stream 7: env key = three finaliser rounds over (seed, global env id), swap k draws
    r = mix(mix(env_key ^ env_hi ^ k * 0x9E3779B1) ^ 7 * 0xC2B2AE3D):
    p = identity; for k = N-1..1: j = (r * (k+1)) >> 32; swap(p[k], p[j]).  Build-defined (the reference stacks
    torch.randperm calls, env_ISCO.py:352-354).  (Until round 6 a Philox call per swap: ten rounds per draw were the
    kernel's whole time.)"""
    M = np.uint64(0xFFFFFFFF)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    gb = np.arange(B, dtype=np.uint64) + np.uint64(env_offset)
    h = _murmur_mix32(np.uint64(seed & 0xFFFFFFFF) ^ np.uint64(0x9E3779B9))
    h = _murmur_mix32(h ^ np.uint64(seed >> 32))
    ekey = _murmur_mix32(h ^ (gb & M)) ^ (gb >> np.uint64(32))
    p = np.tile(np.arange(N, dtype=np.int64), (B, 1))
    rows = np.arange(B)
    sconst = np.uint64((7 * 0xC2B2AE3D) & 0xFFFFFFFF)
    for k in range(N - 1, 0, -1):
        pk = _murmur_mix32(ekey ^ np.uint64((k * 0x9E3779B1) & 0xFFFFFFFF))
        r0 = _murmur_mix32(pk ^ sconst)
        j = ((r0 * np.uint64(k + 1)) >> np.uint64(32)).astype(np.int64)
        tk = p[rows, k].copy()
        p[rows, k] = p[rows, j]
        p[rows, j] = tk
    return p


# --------------------------------------------------------------------------- MCPG outer loop (methods/MCPG.py:292-302, 376-394)


def mcpg_merge_best(temp_max, temp_max_info, now_max_res, now_max_info):
    """The per-chain best-merge loop and the min/max replacement of mcpg(), rlsolver/methods/MCPG.py:376-391, on
    copies.  temp_max / now_max_res f32 [M]; *_info [N, M].  Returns (now_max_res, now_max_info, temp_max_info,
    now_max, now_max_index)."""
    now_max_res, now_max_info, temp_max_info = now_max_res.copy(), now_max_info.copy(), temp_max_info.copy()
    for i0 in range(temp_max.shape[0]):
        if temp_max[i0] > now_max_res[i0]:
            now_max_res[i0] = temp_max[i0]
            now_max_info[:, i0] = temp_max_info[:, i0]
    now_max = now_max_res.max()
    now_max_index = int(np.argmax(now_max_res))
    now_min_index = int(np.argmin(now_max_res))
    now_max_res[now_min_index] = now_max
    now_max_info[:, now_min_index] = now_max_info[:, now_max_index]
    temp_max_info[:, now_min_index] = now_max_info[:, now_max_index]
    return now_max_res, now_max_info, temp_max_info, now_max, now_max_index


def mcpg_get_return(probs, samples, value, total_mcmc_num, repeat_times):
    """get_return, rlsolver/methods/MCPG.py:292-302 (samples [C, N] 0/1, probs [N], value [C]) in float64, and its
    gradient with respect to probs (what autograd returns): objective = mean_c(log_prob_sum_c * value_c)."""
    s = samples.astype(np.float64)
    p = probs.astype(np.float64)
    v = value.astype(np.float64)
    log_prob_sum = np.empty(s.shape[0])
    for j in range(repeat_times):
        j0, j1 = total_mcmc_num * j, total_mcmc_num * (j + 1)
        log_prob_sum[j0:j1] = np.log(s[j0:j1] * p + (1 - s[j0:j1]) * (1 - p)).sum(axis=1)
    objective = (log_prob_sum * v).mean()
    grad = ((s / p - (1 - s) / (1 - p)) * v[:, None]).mean(axis=0)
    return objective, grad


# --------------------------------------------------------------------------- upstream MCPG weighted MaxCut sampler


def mcpg_metro_sampling_upstream(probs, start_status, max_transfer_time, index, u):
    """metro_sampling of the MCPG package, rlsolver/methods/MCPG/sampling.py:67-86 (the same walk as MCPG.py:88-117;
    restated separately because it is a separate function there).  -> (float32 0/1 [N, C], rounds used)"""
    return metro_sampling(probs, start_status, max_transfer_time, index, u)


def mcpg_sampling_maxcut(graph_w, num_nodes, sorted_degree_nodes, start_status, probs, num_ls, change_times, total_mcmc_num,
                         metro_index, metro_u, uniforms):
    """mcpg_sampling_maxcut, rlsolver/methods/MCPG/sampling.py:89-127, float32, with every torch draw supplied.
    graph_w int [E, 3] (n0, n1, weight, 0-based, file order).  Neighbour lists follow append_neighbors
    (dataloader.py:106-125: both endpoints appended per edge, in file order).
    -> (vs, xs_good, start, value, expected)"""
    n0, n1, w = graph_w[:, 0], graph_w[:, 1], graph_w[:, 2].astype(np.float32)
    nbr = [[] for _ in range(num_nodes)]
    nbw = [[] for _ in range(num_nodes)]
    for a, b, ww in zip(n0, n1, w):
        nbr[int(a)].append(int(b)); nbw[int(a)].append(ww)
        nbr[int(b)].append(int(a)); nbw[int(b)].append(ww)
    wdeg = [np.float32(np.sum(x, dtype=np.float32)) for x in nbw]                       # dataloader.py:82
    edge_weight_sum = np.float32(w.sum(dtype=np.float32))
    start, _ = mcpg_metro_sampling_upstream(probs, start_status, change_times, metro_index, metro_u)
    x = start.copy()
    hub = int(sorted_degree_nodes[0])
    x = (x + x[hub]) % np.float32(2)                                                   # :101-103
    x = (x - np.float32(0.5)) * np.float32(2) + np.float32(0.5)                         # :104  -> -0.5 | 1.5
    for cnt in range(num_ls):
        for pos in range(num_nodes):
            node = int(sorted_degree_nodes[pos])
            if nbr[node]:
                tv = (np.asarray(nbw[node], np.float32)[None, :] @ x[nbr[node]]).astype(np.float32)[0]
            else:
                tv = np.zeros(x.shape[1], np.float32)
            tv = tv + uniforms[cnt, pos].astype(np.float32) / np.float32(4)
            x[node] = (tv < wdeg[node] / np.float32(2) + np.float32(0.125)).astype(np.float32)
    expected = ((np.float32(2) * x[n0] - 1) * (np.float32(2) * x[n1] - 1) * w[:, None]).sum(axis=0, dtype=np.float32)
    index = expected.reshape(-1, total_mcmc_num).argmin(axis=0)
    index = np.arange(total_mcmc_num) + index * total_mcmc_num
    vs = (edge_weight_sum - expected[index]) / np.float32(2)
    return vs, x[:, index], start, expected - expected.mean(dtype=np.float32), expected


# --------------------------------------------------------------------------- TSP true 2-opt local search


def tsp_distance_calc(distance_matrix, tour_closed_1based):
    """distance_calc, methods_problem_specific/TSP/opt_2.py:17-22: sequential float64 sum over the closed 1-based tour."""
    d = 0
    t = tour_closed_1based
    for k in range(len(t) - 1):
        d = d + distance_matrix[t[k] - 1, t[k + 1] - 1]
    return d


def tsp_local_search_2_opt(distance_matrix, tour_closed_1based, start_distance, recursive_seeding=-1):
    """local_search_2_opt, methods_problem_specific/TSP/opt_2.py:27-57, in the reference's shape: every pass tries EVERY
    reversal [i..j] of the pass's seed tour, recomputes the whole length of each candidate and keeps the shortest seen
    (strictly shorter only); passes repeat until one brings nothing (recursive_seeding < 0) or `recursive_seeding` times."""
    count = -2 if recursive_seeding < 0 else 0
    best_tour, best_d = list(tour_closed_1based), start_distance
    tracker = best_d * 2
    while count < recursive_seeding:
        seed = list(best_tour)
        n = len(seed)
        for i in range(0, n - 2):
            for j in range(i + 1, n - 1):
                cand = list(seed)
                cand[i:j + 1] = cand[i:j + 1][::-1]
                cand[-1] = cand[0]
                dc = tsp_distance_calc(distance_matrix, cand)
                if best_d > dc:
                    best_tour, best_d = cand, dc
        count += 1
        if tracker > best_d and recursive_seeding < 0:
            tracker, count, recursive_seeding = best_d, -2, -1
        elif best_d >= tracker and recursive_seeding < 0:
            count, recursive_seeding = -1, -2
    return best_tour, best_d


# --------------------------------------------------------------------------- Evaluator (best-so-far tracker)


class EvaluatorOracle:
    """Evaluator.record1 / record2, rlsolver/methods/util_evaluator.py:66-107, on numpy arrays: the constructor records
    (0, v) and (0, v, .) without changing the incumbent (v > v is false); record2 takes the FIRST argmax (argmin when
    minimising) of a batch, or the single solution as it is, logs its value, and replaces the incumbent only on STRICT
    improvement.  Returns if_update like the reference."""

    def __init__(self, x, v, if_maximize):
        self.best_x, self.best_v, self.if_maximize = np.asarray(x).copy(), v, if_maximize
        self.recorder1, self.recorder2 = [], []
        self.record1(0, self.best_v)
        self.record2(0, self.best_v, self.best_x)

    def record1(self, i, v):
        self.recorder1.append((i, v))

    def record2(self, i, vs, xs):
        xs = np.asarray(xs)
        if xs.ndim == 2:
            good_i = int(np.argmax(vs) if self.if_maximize else np.argmin(vs))
            good_x, good_v = xs[good_i], vs[good_i]
        else:
            good_x, good_v = xs, vs
        good_v = float(good_v)
        self.recorder2.append((i, good_v))
        if_update = (good_v > self.best_v) if self.if_maximize else (good_v < self.best_v)
        if if_update:
            self.best_x, self.best_v = good_x.copy(), good_v
        return if_update

    @property
    def first_v(self):
        return self.recorder2[0][1]
