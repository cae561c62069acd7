#!/usr/bin/env python3
"""Per-kernel throughput on the BASELINE.json configs (1 GPU): one JSON line per measurement.
Development / documentation aid; bench.py is the judged harness (headline K4 only).

    python tools/bench_configs.py [--quick] > gpurun_out/configs.jsonl
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rlsolver_amd import ops, ops_mcpg_tsp as mops
from rlsolver_amd.graph import build_csr, generate_ba, generate_gnm, generate_tsp_coords, tsp_tables
from rlsolver_amd.methods import MCPG as amcpg

ap = argparse.ArgumentParser()
ap.add_argument("--quick", action="store_true")
ap.add_argument("--profile", action="store_true", help="few launches per kernel: for rocprofv3 passes (PMC serialises kernels)")
ap.add_argument("--only", default="", help="comma list of suites: maxcut,synthetic,lsba,ls,g70,g14,narrow,tsp,isco,spin,qubo,mcpg")
a = ap.parse_args()
dev = torch.device("cuda:0")
HBM = 8e12


def timeit(fn, iters, warm=2):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def emit(config, kernel, unit, units_per_launch, t, alg_bytes_per_unit=None, note=""):
    rec = {"config": config, "kernel": kernel, "us_per_launch": round(t * 1e6, 2), "unit": unit,
           "units_per_s": units_per_launch / t}
    if alg_bytes_per_unit is not None:
        rec["algorithmic_GBps"] = units_per_launch * alg_bytes_per_unit / t / 1e9
        rec["frac_of_8TBps"] = units_per_launch * alg_bytes_per_unit / t / HBM
    if note:
        rec["note"] = note
    print(json.dumps(rec), flush=True)


def maxcut_suite(tag, n, m, B, seed, iters, mygraph=None):
    g = ops.DeviceGraph(build_csr(mygraph if mygraph is not None else generate_gnm(n, m, seed), num_nodes=n), dev)
    S = 4
    slots = [ops.rand_spins(B, n, s, dev) for s in range(S)]
    if B >= 4096:
        t = timeit(lambda i: ops.rand_spins(B, n, i, dev, out=slots[i % S]), max(3, iters // 4))
        emit(tag, "K14 rand_spins (a4)", "envs", B, t, n, "bytes = the uint8 [B, N] result")
        for sidx in range(S):
            ops.rand_spins(B, n, sidx, dev, out=slots[sidx])
    obj = ops.maxcut_obj(g, slots[0]).to(torch.int32)
    rew = torch.empty(B, dtype=torch.float32, device=dev)
    acts = [ops.rand_actions(B, n, 7, s, dev) for s in range(8)]
    t = timeit(lambda i: ops.maxcut_step(g, slots[i % S], slots[(i + 1) % S], acts[i % 8], obj, rew), iters)
    emit(tag, "K4 maxcut_step (emit)", "env-steps", B, t, 2 * n + 20)
    if B * n * 4 * 6 < 40e9:   # gym surface (env_PPO keeps f32 spins): its own label, 2*4N + 20 bytes per env-step
        fs = [sl.float() for sl in slots] + [slots[0].float(), slots[1].float()]
        tf = timeit(lambda i: ops.maxcut_step(g, fs[i % 6], fs[(i + 1) % 6], acts[i % 8], obj, rew), max(5, iters // 2))
        emit(tag, "K4 maxcut_step (emit, f32 gym surface)", "env-steps", B, tf, 8 * n + 20)
        del fs
    if B <= 4096:   # launch-bound regime: the class surface through the native torch op, eager
        import types
        from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
        for dt, lab in ((torch.float32, "f32 reference surface"), (torch.bool, "1-byte spins")):
            env = Gym(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=10 ** 9), mygraph=mygraph if mygraph is not None else generate_gnm(n, m, seed), device=dev,
                      spin_dtype=dt, reuse_buffers=True)
            env.reset()
            te = timeit(lambda i: env.step(acts[i % 8]), max(200, iters * 20))
            emit(tag, f"EnvMaxcutGym.step in place, eager, torch.ops path ({lab})", "env-steps", B, te, None,
                 "host-bound: one dispatcher call + one HIP launch per step")
            bufs = [torch.empty((B, n), dtype=dt, device=dev) for _ in range(4)]
            te = timeit(lambda i: env.step(acts[i % 8], out=bufs[i % 4]), max(200, iters * 20))
            emit(tag, f"EnvMaxcutGym.step(out=slot), eager, torch.ops path ({lab})", "env-steps", B, te, (8 if dt == torch.float32 else 2) * n + 20)
        tc = timeit(lambda i: ops.maxcut_step(g, slots[i % S], slots[(i + 1) % S], acts[i % 8], obj, rew), max(200, iters * 20))
        emit(tag, "K4 maxcut_step (emit), eager, ctypes path with per-call validation", "env-steps", B, tc, 2 * n + 20)
    if B <= 4096:   # launch-bound regime: the same steps captured as one hipGraph (rlsolver_amd.hipgraph)
        from rlsolver_amd.hipgraph import CapturedLaunches
        T = 64

        def rollout():
            for k in range(T):
                ops.maxcut_step(g, slots[k % S], slots[(k + 1) % S], acts[k % 8], obj, rew)
        cap = CapturedLaunches(rollout, dev)
        tg = timeit(lambda i: cap.replay(), max(3, iters // 16)) / T
        emit(tag, "K4 maxcut_step (emit), 64 steps per hipGraph replay", "env-steps", B, tg, 2 * n + 20,
             "per-step time inside the graph")
    x = slots[0].clone()
    t = timeit(lambda i: ops.maxcut_step(g, x, x, acts[i % 8], obj, rew), iters)
    emit(tag, "K4 maxcut_step (in place)", "env-steps", B, t, None, "O(deg) bytes per step")
    out = torch.empty(B, dtype=torch.int64, device=dev)
    t = timeit(lambda i: ops.maxcut_obj(g, slots[0], out), iters)
    emit(tag, "K1 maxcut_obj", "evals", B, t, n + 8)
    mask = torch.rand((B, n), device=dev) < 8.0 / n
    vs = ops.maxcut_obj(g, x)
    # accepted rows are WRITTEN (N bytes each): the same mask applied again and again flips a row back and forth, and at least one of
    # the two directions is accepted (ties are) -- the accept rate of this loop is measured and its rows counted
    def accept_rate(m):
        acc = 0.0
        for _ in range(4):
            v0 = vs.clone()
            x0 = x.clone()
            ops.maxcut_propose_accept(g, x, m, vs)
            acc += float((x != x0).any(dim=1).float().mean()) / 4
            del v0, x0
        return acc
    t = timeit(lambda i: ops.maxcut_propose_accept(g, x, mask, vs), max(3, iters // 4))
    ar = accept_rate(mask)
    emit(tag, "K6 propose_accept", "proposals", B, t, 2 * n + 16 + ar * n, f"x in, byte mask in, accepted rows out (accept rate of this loop {ar:.3f})")
    if n * 8 + 4096 <= 160 * 1024:
        from rlsolver_amd.ops_mcpg_tsp import PackedChains
        mwords = PackedChains.pack(mask.t().contiguous()).words
        t = timeit(lambda i: ops.maxcut_propose_accept(g, x, mwords, vs), max(3, iters // 4))
        ar = accept_rate(mwords)
        emit(tag, "K6 propose_accept, bit-packed mask", "proposals", B, t, 2 * n + n // 8 + 16 + ar * n,
             f"x in, mask words in, accepted rows out (accept rate of this loop {ar:.3f})")
    t = timeit(lambda i: ops.maxcut_greedy_sweep(g, x, vs), max(2, iters // 10))
    # LDS-op rate: a sweep reads one 64-env word per (node, neighbour) and per node, and writes one per node;
    # LDS peak = 128 B / clk / CU x 256 CUs x 2.4 GHz = 78.6 TB/s
    tiles = (B + 63) // 64
    lds_lane_ops = (int(g.struct.nnz) + 2 * n) * tiles
    emit(tag, "K5 greedy_sweep", "candidate flips", B * n, t, (2 * n + 16) / n,
         "on-chip bound by design: %d dependency levels (a workgroup barrier + one LDS round trip each), %.1f us per "
         "level; LDS word ops %.3g /s = %.2f TB/s of 8-byte lane accesses = %.1f %% of the LDS peak" % (
             g.num_sweep_levels, t * 1e6 / max(g.num_sweep_levels, 1), lds_lane_ops / t, lds_lane_ops * 8 / t / 1e12,
             100.0 * lds_lane_ops * 8 / t / 78.6e12))
    d = None
    t = timeit(lambda i: ops.maxcut_delta_all(g, x), max(3, iters // 4))
    emit(tag, "K3 delta_all", "envs", B, t, 5 * n)
    return g


def local_search_suite(tag, n, m, seed, B, iters, mygraph=None):
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    env = EnvMaxcut(mygraph=mygraph if mygraph is not None else generate_gnm(n, m, seed), device=dev, num_nodes=n)
    torch.manual_seed(0)
    xs = env.generate_xs_randomly(B)
    vs = env.calculate_obj_values(xs)
    t = timeit(lambda i: env.local_search_inplace(xs, vs, num_iters=8, num_spin=8, noise_std=0.3), iters, warm=1)
    few = ops.ls_rounds_supported(env.graph, 8) and ops.ls_scratch_bytes(env.graph, B, ops.ls_weight_dtype(env.graph, 1), 8) > 0
    form = ("ls_weights + threshold kernel + 8 mask kernels (a tile's rows over several workgroups) + all rounds on one tile load + K5: "
            "a batch of few tiles" if few
            else "ls_weights pre-pass + fused kernel: threshold, 8 proposal rounds, greedy sweep" if ops.local_search_fusable(env.graph, 8, B)
            else "ls_weights + threshold kernel + 8 proposal-round kernels + K5: a graph beyond the fused kernel's LDS layout"
            if ops.ls_rounds_supported(env.graph, 8) else "ls_weights + torch noise / kthvalue + 8 x K6 + K5")
    emit(tag, f"local_search_inplace ({form})", "candidate evaluations",
         B * (n + 8), t, None, f"B={B}; the reference performs N+8 full objective evaluations per env per call")


def mcpg_suite(tag, n, m_ba, C, num_ls, iters, mygraph=None):
    mg = mygraph if mygraph is not None else generate_ba(n, m_ba, seed=5)
    arr = np.asarray(mg, dtype=np.int64)
    data = amcpg.make_data(n, arr[:, 0], arr[:, 1], dev)
    torch.manual_seed(0)
    xs = (torch.rand((n, C), device=dev) < 0.5).float()
    probs = torch.full((n,), 0.5, device=dev)
    t = timeit(lambda i: amcpg.sampler_func(data, xs, num_ls, C // 128, 128, dev), iters, warm=1)
    emit(tag, f"K7+K8 sampler_func (num_ls={num_ls})", "node updates", C * n * num_ls, t, None,
         f"C={C} chains, E={len(mg)}; f32 [N,C] in/out = {2 * 4 * n} B per chain")
    emit(tag, f"K7+K8 sampler_func (num_ls={num_ls})", "chain-sweeps", C * num_ls, t, 2 * 4 * n / num_ls)
    T = n // 10
    t = timeit(lambda i: amcpg.metro_sampling(probs, xs, T, dev), iters, warm=1)
    emit(tag, "K9 metro_sampling (f32 [N,C] surface)", "proposals", C * T, t, None, f"T={T}")
    # the on-device round: bit-packed chains end to end
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    M, R = C // 128, 128
    kept = PackedChains.pack((torch.rand((n, M), device=dev) < 0.5).float())
    out = PackedChains.empty(n, C, dev)
    t = timeit(lambda i: amcpg.metro_sampling_packed(probs, kept, T, num_chains=C, out=out), iters, warm=1)
    emit(tag, "K9 metro_sampling_packed (broadcast start, T rounds)", "proposals", C * T, t, (n // 8 + n // 8) / T,
         f"T={T}; bytes = packed tile out + broadcast tile in")
    t = timeit(lambda i: amcpg.sampler_func_packed(data, out, num_ls, M, R), iters, warm=1)
    emit(tag, f"K7+K8 sampler_func_packed (num_ls={num_ls})", "chain-sweeps", C * num_ls, t, 2 * (n // 8) / num_ls,
         "bytes = packed tile in + out")
    emit(tag, f"K7+K8 sampler_func_packed (num_ls={num_ls})", "node updates", C * n * num_ls, t, None, f"C={C} chains")
    vs0 = torch.zeros(M, device=dev)
    rnd = amcpg.MCPGRound(data, kept.clone(), vs0, M, R, num_ls)
    t = timeit(lambda i: rnd.step(probs), iters, warm=1)
    emit(tag, "MCPG round on device (metro + sampler + best-merge)", "kept chains (reference's num_samples)", M, t, None,
         f"M={M}, R={R}: the reference's num_samples_per_second = M / round time")
    val = torch.randn(C, device=dev)
    pr = torch.full((n,), 0.4, device=dev, requires_grad=True)

    def ret(i):
        o = amcpg.get_return(pr, out, val)
        o.backward()
    t = timeit(ret, max(iters, 20), warm=3)     # a handful of small torch ops around one kernel: host-paced, needs the repeats
    emit(tag, "get_return forward + backward from bit sums", "chains", C, t, n // 8, "bytes = the packed samples")


def tsp_suite(tag, N, B, iters):
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, 100), K=20)
    d = torch.from_numpy(dist).to(dev)
    perms = mops.rand_perms(B, N, 3, dev)
    t = timeit(lambda i: mops.rand_perms(B, N, 3 + i, dev), max(3, iters // 4))
    emit(tag, "K14 rand_perms", "tours", B, t, 8 * N, "bytes = the int64 [B, N] result")
    t = timeit(lambda i: mops.tsp_tour_length(d, perms), iters)
    emit(tag, "K12 tsp_tour_length", "tours", B, t, 8 * N + 4)
    K = near.shape[1]
    near32, rnd32 = torch.from_numpy(near.astype("int32")).to(dev), torch.from_numpy(rnd.astype("int32")).to(dev)
    tab8 = mops.tsp_tables8(near32, rnd32)
    t = timeit(lambda i: mops.tsp_swap_delta_all(d, perms, None, 0.5, nearest=near32, random=rnd32, near_threshold=K / (K + 1), seed=i, tables8=tab8), iters)
    emit(tag, "K13 tsp_swap_delta_all", "envs (N candidate moves each)", B, t, 8 * N + 13 * N, "partners drawn in the kernel (SURVEY 8d: 21N)")
    sel = torch.roll(perms, 7, 1).contiguous()
    t = timeit(lambda i: mops.tsp_swap_delta_all(d, perms, sel, 0.5), max(3, iters // 2))
    emit(tag, "K13 tsp_swap_delta_all, selected given", "envs (N candidate moves each)", B, t, 8 * N + 8 * N + 13 * N, "the recorded-draw hook: + 8N in")


def isco_suite(iters):
    """I1 / I2: one ISCO sampler step = one kernel (the reference: ~25-40 torch ops incl. two full sorts and an autograd
    pass for MaxCut, a sort + searchsorted + [B, N, N-1] gather per round for TSP)."""
    from rlsolver_amd.envs.env_ISCO import ISCO_TSP
    from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
    N, B, L = 100, 1 << 16, 8
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, 100), K=20)
    params = {"num_nodes": N, "distance": torch.from_numpy(dist).to(dev), "nearest_indices": torch.from_numpy(near).to(dev),
              "random_indices": torch.from_numpy(rnd).to(dev)}
    env = ISCO_TSP(params, batch_size=B, K=20, device=dev)
    x = env.random_gen_init_sample()
    t = timeit(lambda i: env.step(x, L, 1.0), iters)
    emit("TSP-100 uniform, B=2^16", f"I2 ISCO_TSP.step (path_length={L}: {L} rounds of opt_2 + softmax + Gumbel draw + swap, MH accept)",
         "proposal rounds", B * L, t, None, "one kernel per step; tour, inverse and D in LDS")
    n, m, Bm, Lm = 2000, 19990, 4096, 16
    g = np.asarray(generate_gnm(n, m, 22), dtype=np.int64)
    pm = {"num_nodes": n, "num_edges": m, "edge_from": torch.from_numpy(g[:, 0].copy()).to(dev), "edge_to": torch.from_numpy(g[:, 1].copy()).to(dev)}
    envm = ISCO_maxcut(pm, batch_size=Bm, device=dev)
    xm = envm.random_gen_init_sample()
    t = timeit(lambda i: envm.step(xm, Lm, 1.0), iters)
    emit("G22-sized G(2000,19990), 4096 samples", f"I1 ISCO_maxcut.step (path_length={Lm}: local distribution, Gumbel top-k, both path log-probs, MH accept)",
         "sampler steps", Bm, t, 8 * n, "one kernel per step, wave per sample; bytes = the f32 sample in and out")


def spin_suite(tag, n, m, B, T, iters):
    """S1: the S2V / ECO / PECO env on a shared graph: the O(deg) step kernel (nothing streamed: the rows that change
    everywhere are per-env scalars + a last-flip step), the rows-only observation that materialises them (7 rows written,
    spins + immediate reward + last-flip read) and the step through the class surface."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
    rng = np.random.RandomState(1)
    mg = [(u, v, int(rng.choice([-1, 1]))) for u, v, _ in generate_gnm(n, m, 22)]
    for label, kw in (("ECO observables, BLS reward", dict()),
                      ("ECO observables, BLS + basin reward (visited-state memory)", dict(basin_reward=1.0 / n))):
        env = SpinSystem(mg, n, B, max_steps=T, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS,
                         norm_rewards=True, spin_basis=SpinBasis.BINARY, device=dev, include_adjacency=False, **kw)
        acts = [ops.rand_actions(B, n, 11, s, dev) for s in range(8)]
        rew = torch.empty(B, device=dev)
        R = torch.ops.rlsolver_hip

        def one(i):   # the bare kernel, as EnvMaxcutGym's launcher: no observation copy
            if env.current_step >= T:
                env.current_step = 0
            env.current_step += 1
            R.spin_step(env.graph.handle, env._env_handle, env._state, env._rows, acts[i % 8], rew, None, env._max_local, 1.0, 1,
                        float(n), env.current_step - 1, False, 0.0, "basin" in label, float(np.float32(1.0 / n)))
        t = timeit(one, iters)
        emit(tag, f"S1 spin_step ({label})", "env-steps", B, t, None,
             "O(deg) per env: latency-bound, no row is streamed (round 2: 6 rows x 4N bytes per env-step, 250 us at 2^14 envs)")
        out = torch.empty((B, 7, n), device=dev)
        t = timeit(lambda i: env.get_observation(out=out), iters)
        emit(tag, f"S1 get_observation, rows only ({label})", "envs", B, t, 4 * (7 + 2) * n + 4 * n,
             "bytes = 7 rows written, spins + immediate reward read (f32), last-flip steps read (int32)")

        def full(i):
            if env.current_step >= T:
                env.current_step = 0
            env.step(acts[i % 8])
        t = timeit(full, iters)
        emit(tag, f"SpinSystem.step() incl. the rows-only observation ({label})", "env-steps", B, t, 4 * (7 + 2) * n + 4 * n,
             "bytes = the observation kernel's")


def spin_train_suite(tag, n, m_ins, B, T, iters):
    """S1 on per-env couplings (the training envs): one draw of B Barabasi-Albert graphs, the dense reset, the step kernel and
    the step through the class surface (observation [B, 7 + N, N] with each env's own matrix rows included); beside the draw,
    the reference's way of drawing the same graphs with torch ops (a loop over the nodes: row sums of [B, N, N], multinomial,
    two scatters)."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
    from rlsolver_amd.envs.util_envs_PECO import EdgeType, RandomBAGraphGenerator
    gg = RandomBAGraphGenerator(n, m_ins, EdgeType.DISCRETE, B, dev)
    t = timeit(lambda i: gg.get(seed=i), iters, warm=2)
    emit(tag, "rand_couplings (BA, one kernel)", "graphs", B, t, 4 * n * n, "bytes = the f32 [B, N, N] result")

    def torch_ba():
        adj = torch.zeros((B, n, n), device=dev)
        for i in range(m_ins + 1):
            adj[:, i, :i + 1] = 1
            adj[:, :i + 1, i] = 1
        rows = torch.arange(B, device=dev).repeat_interleave(m_ins)
        for v in range(m_ins + 1, n):
            deg = adj.sum(dim=-1)
            pick = torch.multinomial(deg / deg.sum(dim=-1, keepdim=True), m_ins, replacement=False).view(-1)
            adj[rows, v, pick] = 1
            adj[rows, pick, v] = 1
        return adj
    if not a.profile:   # thousands of small torch kernels: kept out of the rocprofv3 passes
        t0 = timeit(lambda i: torch_ba(), 2, warm=1)
        emit(tag, "the same draw as a torch op chain (per-node loop, as the reference generator)", "graphs", B, t0, 4 * n * n, "baseline beside rand_couplings")
    env = SpinSystem(None, None, B, max_steps=T, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS, norm_rewards=True,
                     spin_basis=SpinBasis.BINARY, device=dev, graph_generator=gg)
    t = timeit(lambda i: env.reset(), max(3, iters // 3), warm=1)
    emit(tag, "SpinSystem.reset() on fresh graphs (draw + gain cache + rows + observation)", "envs", B, t, 4 * n * n * 2 + 4 * (7 + n) * n,
         "bytes = matrix written + read, observation written")
    acts = [ops.rand_actions(B, n, 11, s, dev) for s in range(8)]
    rew = torch.empty(B, device=dev)

    def one(i):
        if env.current_step >= T:
            env.current_step = 0
        env.current_step += 1
        torch.ops.rlsolver_hip.spin_step_dense(env._matrix, env.max_local_reward_available_, env._env_handle, env._state, env._rows,
                                               acts[i % 8], rew, None, 1.0, 1, float(n), env.current_step - 1, False, 0.0, False, 0.0)
    t = timeit(one, iters)
    emit(tag, "S1 spin_step_dense (ECO observables, BLS reward)", "env-steps", B, t, 4 * n * 4,
         "algorithmic bytes = the flipped node's matrix row read + the gains / immediate-reward entries it changes")
    out = torch.empty((B, 7 + n, n), device=dev)

    def full(i):
        if env.current_step >= T:
            env.current_step = 0
        env.step(acts[i % 8])
    t = timeit(full, iters)
    emit(tag, "SpinSystem.step() incl. observation [B, 7 + N, N] with per-env matrix rows", "env-steps", B, t, 4 * n + 4 * (7 + n) * n + 4 * n * n + 4 * 3 * n,
         "bytes = step + observation written + matrix and resident rows read")


def qubo_suite(tag, n, C, num_ls, iters, sparse=True):
    from rlsolver_amd.methods import MCPG_qubo as mq
    rng = np.random.RandomState(3)
    Q = np.triu(rng.randint(10, 101, size=(n, n)) * rng.choice([-1, 1], size=(n, n)) * (rng.rand(n, n) < 0.8), 0)
    Q = (Q + np.triu(Q, 1).T).astype(np.float32)
    Qd = torch.from_numpy(Q).to(dev)
    xs = (torch.rand((n, C), device=dev) < 0.5).float()
    for binary in (False, True):
        t = timeit(lambda i: mq.qubo_local_search_value(Qd, xs, num_ls, binary), iters, warm=1)
        rec_flops = 2 * n * n * (num_ls + 1) * C / t
        print(json.dumps({"config": tag, "kernel": f"K11 qubo_local_search_value ({'0/1' if binary else '+-1'}, num_ls={num_ls})",
                          "us_per_launch": round(t * 1e6, 2), "unit": "variable updates", "units_per_s": C * n * (num_ls + 1) / t,
                          "achieved_TFLOPs_f32": rec_flops / 1e12, "frac_of_157TFLOPs_f32_matrix_peak": rec_flops / 157.3e12,
                          "note": f"n={n} dense, C={C}; compute-bound contraction on v_mfma_f32_32x32x2_f32: roofline = the f32 "
                                  "matrix peak; flops = 2 n^2 per chain per sweep plus one pass for the value"}), flush=True)
    if not sparse:
        return
    Qs = Q * (rng.rand(n, n) < 0.02)
    Qs = (np.triu(Qs) + np.triu(Qs, 1).T).astype(np.float32)
    csr = mq.qubo_to_csr(torch.from_numpy(Qs).to(dev))
    t = timeit(lambda i: mq.qubo_sparse_local_search_value(csr, xs, num_ls, False), iters, warm=1)
    Qsd = torch.from_numpy(Qs).to(dev)
    td = timeit(lambda i: mq.qubo_local_search_value(Qsd, xs, num_ls, False), iters, warm=1)
    tq = timeit(lambda i: mq.qubo_sparse_local_search_value(csr[:3], xs, num_ls, False), max(1, iters // 2), warm=1)
    emit(tag, f"K11s qubo_sparse_local_search_value by levels (+-1, 2 % fill, num_ls={num_ls})", "variable updates", C * n * (num_ls + 1), t, None,
         f"nnz={int(csr[0][-1])}, {csr[3].numel() - 1} levels; the dense MFMA kernel on the SAME matrix: {td * 1e6:.0f} us "
         f"({t / td:.2f} x); one wave per 64 chains walking the rows in order (the form until round 5): {tq * 1e6:.0f} us")


it = 5 if (a.quick or a.profile) else 30
only = set(w for w in a.only.split(",") if w)
want = lambda k: not only or k in only
if want("maxcut"):
    maxcut_suite("G22-sized G(2000,19990), B=2^16", 2000, 19990, 1 << 16, 22, it)
if want("synthetic") and not a.profile:   # (grids coincide with the BASELINE rows of tools/kernel_table.py) north_star: "throughput on Gset and synthetic BA/ER graphs"
    from rlsolver_amd.graph import generate_ba
    maxcut_suite("BA n=2000 m=4 (hubs: max degree ~150), B=2^16", 2000, 0, 1 << 16, 0, it, mygraph=generate_ba(2000, 4, 3))
    maxcut_suite("BA n=10000 m=5, B=2^16", 10000, 0, 1 << 16, 0, max(3, it // 3), mygraph=generate_ba(10000, 5, 5))
    maxcut_suite("ER G(n=2000, p=0.005 -> m=9995), B=2^16", 2000, 9995, 1 << 16, 31, it)
    mcpg_suite("MCPG on a G22-sized G(2000, 19990), 2^16 chains", 2000, 0, 1 << 16, 8, max(3, it // 3), mygraph=generate_gnm(2000, 19990, 22))
if want("synthetic") or want("lsba"):   # (under --profile too: the round kernels have rows of their own in tools/kernel_table.py)
    from rlsolver_amd.graph import generate_ba
    if not a.profile:
        local_search_suite("BA n=2000 m=4, dREINFORCE batch", 2000, 0, 0, 4096, max(2, it // 5), mygraph=generate_ba(2000, 4, 3))
    local_search_suite("BA n=10000 m=5 (hubs of degree >= 256), dREINFORCE batch", 10000, 0, 0, 4096, max(2, it // 5), mygraph=generate_ba(10000, 5, 5))
    if a.profile:   # (2^15 envs: at 2^16 the sweep kernel's grid would coincide with the G22 row's)
        local_search_suite("BA n=10000 m=5 (hubs of degree >= 256), dREINFORCE batch x8", 10000, 0, 0, 32768, max(2, it // 5), mygraph=generate_ba(10000, 5, 5))
    else:
        local_search_suite("BA n=10000 m=5 (hubs of degree >= 256), dREINFORCE batch x16", 10000, 0, 0, 65536, max(2, it // 5), mygraph=generate_ba(10000, 5, 5))
if want("ls"):
    local_search_suite("G22-sized, dREINFORCE batch", 2000, 19990, 22, 4096, max(2, it // 5))
    local_search_suite("G22-sized, dREINFORCE batch x16", 2000, 19990, 22, 65536, max(2, it // 5))
if want("g70"):
    maxcut_suite("G70-sized G(10000,9999), B=2^17 (one GPU's shard of 2^20)", 10000, 9999, 1 << 17, 70, max(3, it // 3))
if want("g14"):
    maxcut_suite("G14-sized G(800,4694), B=256", 800, 4694, 256, 14, it)
if want("narrow") and not a.profile:   # round 5: small batches and graphs past the half tile run on narrow tiles (16 / 8 envs per workgroup)
    maxcut_suite("G70-sized G(10000,9999), B=4096 (the reference's batch: narrow tiles)", 10000, 9999, 4096, 70, it)
    maxcut_suite("G(44000, 88000), B=4096 (past the half tile: narrow tiles, uint16 words)", 44000, 88000, 4096, 44, max(3, it // 3))
    local_search_suite("G(44000, 88000), past the half tile", 44000, 88000, 44, 4096, 2)
    maxcut_suite("G(100000, 200000), B=4096 (narrow tiles, byte words)", 100000, 200000, 4096, 100, max(3, it // 3))
if want("tsp"):
    tsp_suite("TSP-100 uniform, B=2^16", 100, 1 << 16, it)
if want("isco"):
    isco_suite(it)
if want("spin"):
    spin_suite("G22-sized +-1 weighted, B=2^14", 2000, 19990, 1 << 14, 64, it)
    spin_suite("BA-200-sized (ECO), B=4096", 200, 784, 4096, 400, it)
    spin_train_suite("PECO training envs: BA-200 (m=4) per env, B=1024", 200, 4, 1024, 400, it)
    spin_train_suite("PECO training envs: BA-20 (m=4) per env, B=4096", 20, 4, 4096, 40, it)
if want("qubo"):
    qubo_suite("nbiq-style dense QUBO n=1000, 2^13 chains", 1000, 1 << 13, 2, 2)
    qubo_suite("nbiq-style dense QUBO n=1000, 2^15 chains", 1000, 1 << 15, 2, 2)
if want("mcpg"):
    mcpg_suite("BA n=10^4 m=5, 2^18 chains", 10000, 5, 1 << 18 if not a.quick else 1 << 14, 8, 2)
