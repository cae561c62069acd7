"""elegantrl vec-env contract + torch.ops registration + a dREINFORCE-shaped outer loop."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from tests.gpu_util import DEV, gnm_arr

pytestmark = pytest.mark.gpu


def test_vec_env_contract():
    from rlsolver_amd.envs.vec_env import MaxcutVecEnv
    n, m, B = 300, 1500, 70
    garr = gnm_arr(n, m, 3)
    env = MaxcutVecEnv([tuple(int(v) for v in r) for r in garr], n, B, max_step=5, gpu_id=0)
    assert (env.num_envs, env.state_dim, env.action_dim, env.if_discrete, env.max_step) == (B, n, n, True, 5)
    torch.manual_seed(0)
    state, info = env.reset()
    assert state.shape == (B, n) and state.dtype == torch.float32 and isinstance(info, dict)
    ref = onp.PPOEnvOracle(garr, n, 5, False)
    ref.reset_to(state.cpu().numpy() > 0)
    rng = np.random.RandomState(1)
    for t in range(5):
        a = torch.from_numpy(rng.randint(0, n, size=B).astype(np.int32)).to(DEV)     # int32 like AgentBase.py:149
        state, reward, terminal, truncate, info = env.step(a)
        _, rr, dd, cc = ref.step(a.cpu().numpy().astype(np.int64))
        assert reward.dtype == torch.float32 and terminal.dtype == torch.bool and truncate.dtype == torch.bool
        assert np.array_equal(reward.cpu().numpy(), rr) and np.array_equal(info["obj"].cpu().numpy(), cc)
        assert not terminal.any() and bool(truncate.all()) == (t == 4)
    assert np.array_equal(state.cpu().numpy(), ref.xs)


def test_torch_ops_equal_raw_c_abi_calls_and_check_shapes():
    """torch.ops.rlsolver_hip.* (C++ ops over the C ABI: the package's one host path) give what RAW ctypes calls of the
    same C-ABI functions give, honour torch's current stream, and turn wrongly shaped / placed arguments into errors
    instead of device out-of-bounds accesses."""
    import ctypes as C
    from rlsolver_amd import _abi, ops, ops_mcpg_tsp as mops, torch_ops
    from rlsolver_amd.graph import build_csr, generate_gnm, generate_tsp_coords, tsp_tables
    R = torch.ops.rlsolver_hip
    P = lambda t: C.c_void_p(t.data_ptr())
    S = lambda: C.c_void_p(torch.cuda.current_stream(DEV).cuda_stream)
    n, B = 256, 130
    g = ops.DeviceGraph(build_csr(generate_gnm(n, 1200, 5), num_nodes=n), DEV)
    h = torch_ops.graph_handle(g)
    assert h == g.handle
    xs = ops.rand_spins(B, n, 3, DEV)
    x_raw = torch.empty_like(xs)
    _abi.call("rls_rand_spins", P(x_raw), B, n, C.c_uint64(3), 0, S())
    assert torch.equal(xs, x_raw)
    obj, obj_raw = torch.empty(B, dtype=torch.int64, device=DEV), torch.empty(B, dtype=torch.int64, device=DEV)
    R.maxcut_obj(h, xs, obj)
    _abi.call("rls_maxcut_obj", g.ref, P(xs), 1, B, P(obj_raw), S())
    assert torch.equal(obj, obj_raw) and torch.equal(obj, ops.maxcut_obj(g, xs))
    d, d_raw = torch.empty((B, n), dtype=torch.int32, device=DEV), torch.empty((B, n), dtype=torch.int32, device=DEV)
    R.maxcut_delta_all(h, xs, d)
    _abi.call("rls_maxcut_delta_all", g.ref, P(xs), B, P(d_raw), S())
    assert torch.equal(d, d_raw)
    x2, v2 = xs.clone(), obj.clone()
    x3, v3 = xs.clone(), obj.clone()
    R.maxcut_greedy_sweep(h, x2, v2)
    _abi.call("rls_maxcut_greedy_sweep", g.ref, P(x3), B, P(v3), S())
    assert torch.equal(x2, x3) and torch.equal(v2, v3)
    # K4 through the op on a side stream == the raw call on the current one
    act = ops.rand_actions(B, n, 1, 0, DEV)
    o1, o2 = obj.to(torch.int32), obj.to(torch.int32)
    r1, r2 = torch.empty(B, device=DEV), torch.empty(B, device=DEV)
    y1, y2 = torch.empty_like(xs), torch.empty_like(xs)
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        R.maxcut_step(h, xs, y1, act, o1, r1, None, None, 0.0)
    side.synchronize()
    _abi.call("rls_maxcut_step", g.ref, P(xs), P(y2), 1, B, P(act), P(o2), P(r2), None, None, 0.0, S())
    assert torch.equal(y1, y2) and torch.equal(o1, o2) and torch.equal(r1, r2)
    # a TSP op and a random op; a seed with the top bit set survives the int64 schema
    dist, near, rnd = tsp_tables(generate_tsp_coords(40, 1), K=5)
    big = (1 << 63) + 12345
    perms = mops.rand_perms(64, 40, big, DEV)
    p2 = torch.empty_like(perms)
    _abi.call("rls_rand_perms", P(p2), 64, 40, C.c_uint64(big), 0, S())
    assert torch.equal(perms, p2)
    dd = torch.from_numpy(dist).to(DEV)
    length, l_raw = torch.empty(64, device=DEV), torch.empty(64, device=DEV)
    R.tsp_tour_length(dd, perms, length)
    _abi.call("rls_tsp_tour_length", P(dd), 40, P(perms), 64, P(l_raw), S())
    assert torch.equal(length, l_raw)
    with pytest.raises(NotImplementedError):
        R.maxcut_obj(h, xs.cpu(), obj.cpu())                       # no CPU kernel registered
    with pytest.raises(RuntimeError):
        R.maxcut_obj(h, xs.float().double(), obj)                  # dtype checked in the op
    with pytest.raises(RuntimeError):
        R.maxcut_obj(0, xs, obj)                                   # null graph handle
    # shapes are checked against the graph handle and against each other (no device out-of-bounds access)
    with pytest.raises(RuntimeError, match="must be \\[B, 256\\]"):
        R.maxcut_obj(h, xs[:, :200].contiguous(), obj)
    with pytest.raises(RuntimeError, match="obj must hold 130"):
        R.maxcut_obj(h, xs, obj[:100])
    with pytest.raises(RuntimeError, match="out must be"):
        R.maxcut_delta_all(h, xs, d[:, :100].contiguous())
    with pytest.raises(RuntimeError, match="action must hold"):
        R.maxcut_step(h, xs, y1, act[:5], o1, r1, None, None, 0.0)
    with pytest.raises(RuntimeError, match="dist must be"):
        R.tsp_tour_length(dd[:30, :30].contiguous(), perms, length)
    with pytest.raises(RuntimeError, match="mask_scratch must hold at least"):
        R.mcpg_merge_best(torch.zeros(128, device=DEV), torch.zeros((2, 50), dtype=torch.int64, device=DEV),
                          torch.zeros(128, device=DEV), torch.zeros((2, 50), dtype=torch.int64, device=DEV), 128,
                          torch.zeros(1, dtype=torch.int64, device=DEV), None, None)
    import types
    from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
    env = Gym(types.SimpleNamespace(num_nodes=n, num_envs=B, num_steps=3), mygraph=generate_gnm(n, 1200, 5), device=DEV)
    assert env._step_op is torch.ops.rlsolver_hip.maxcut_step


def test_dreinforce_shaped_outer_loop_improves():
    """The call pattern of L2A/demo_instance.py:141-165 (sub-set sampling replaced by random
    restarts): local_search_inplace x searchers -> pick_xs_by_vs -> update_xs_by_vs -> best tracking."""
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    from rlsolver_amd.methods.util_read_data import pick_xs_by_vs, update_xs_by_vs
    n, m = 800, 4694
    garr = gnm_arr(n, m, 14)
    env = EnvMaxcut(mygraph=[tuple(int(v) for v in r) for r in garr], device=DEV, if_bidirectional=True, num_nodes=n)
    torch.manual_seed(0)
    num_sims, num_repeats = 16, 8
    best_xs = env.generate_xs_randomly(num_sims)
    best_vs = env.calculate_obj_values(best_xs)
    v_start = best_vs.clone()
    for it in range(3):
        full_xs = best_xs.repeat(num_repeats, 1)
        full_xs ^= torch.rand(full_xs.shape, device=DEV) < 0.02
        full_vs = env.calculate_obj_values(full_xs)
        for _ in range(2):
            env.local_search_inplace(full_xs, full_vs, num_iters=4, num_spin=8)
        good_xs, good_vs = pick_xs_by_vs(full_xs, full_vs, num_repeats, True)
        update_xs_by_vs(best_xs, best_vs, good_xs, good_vs, True)
    assert (best_vs >= v_start).all() and float(best_vs.float().mean()) > 0.6 * m
    assert np.array_equal(best_vs.cpu().numpy(), onp.maxcut_obj(best_xs.cpu().numpy(), garr, True))


def test_k_spin_simulator_aliases(golden):
    from rlsolver_amd.envs.env_k_spin import MaxcutSimulatorReinforce, SimulatorGraphMaxCut
    z = golden("maxcut_obj")
    graph = [tuple(int(v) for v in r) for r in z["BA_100_ID0/graph"]]
    for cls, kw in ((SimulatorGraphMaxCut, dict(graph=graph)), (MaxcutSimulatorReinforce, dict(graph=graph))):
        for bidir in (False, True):
            sim = cls(device=DEV, if_bidirectional=bidir, **kw)
            t = f"BA_100_ID0/bidir{int(bidir)}/seed0"
            xs = torch.from_numpy(z[f"{t}/xs"]).to(DEV).bool()
            assert np.array_equal(sim.calculate_obj_values(xs).cpu().numpy(), z[f"{t}/obj"])
            assert np.array_equal(sim.calculate_obj_values_for_loop(xs, if_sum=False).cpu().numpy(), z[f"{t}/cutdeg"])
            sol = sim.generate_solutions_randomly(8)
            assert sol.shape == (8, sim.num_nodes) and not sol[:, 0].any()


def test_rollout_captured_in_a_hipgraph_equals_eager():
    """64 gym steps over two ping-pong state buffers as ONE graph launch (rlsolver_amd.hipgraph)."""
    import numpy as np
    from rlsolver_amd import ops
    from rlsolver_amd.hipgraph import CapturedLaunches
    from tests.gpu_util import device_graph, gnm_arr
    n, m, B, T = 800, 4694, 256, 64
    g = device_graph(gnm_arr(n, m, seed=14), n, 0)
    x0 = ops.rand_spins(B, n, 3, DEV)
    acts = torch.stack([ops.rand_actions(B, n, 5, t, DEV) for t in range(T)])
    bufs = [x0.clone(), torch.empty_like(x0)]
    obj0 = ops.maxcut_obj(g, x0).to(torch.int32)
    obj, rew, tot = obj0.clone(), torch.empty(B, dtype=torch.float32, device=DEV), torch.zeros(B, device=DEV)

    def rollout():
        for t in range(T):
            ops.maxcut_step(g, bufs[t & 1], bufs[(t + 1) & 1], acts[t], obj, rew)
            tot.add_(rew)

    rollout()                                            # eager reference
    want_x, want_obj, want_tot = bufs[T & 1].clone(), obj.clone(), tot.clone()
    cap = CapturedLaunches(rollout, DEV)                 # warm-up + capture ran the rollout twice more; reset, replay
    for _ in range(2):
        bufs[0].copy_(x0); obj.copy_(obj0); tot.zero_()
        cap.replay()
        torch.cuda.synchronize()
        assert torch.equal(bufs[T & 1], want_x) and torch.equal(obj, want_obj) and torch.equal(tot, want_tot)
    assert torch.equal(ops.maxcut_obj(g, bufs[T & 1]), obj.long())


def test_env_state_dict_roundtrip_gym_and_local_search():
    """env-state checkpoint -- a gym env restored from state_dict() continues bit for bit; the
    LocalSearch incumbents likewise."""
    import types
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    from rlsolver_amd.envs.vec_env import MaxcutVecEnv
    from rlsolver_amd.graph import generate_gnm
    from rlsolver_amd.methods.LocalSearch import LocalSearch
    n, B = 300, 70
    mg = generate_gnm(n, 1500, 2)
    env = MaxcutVecEnv(mg, n, B, max_step=7)
    torch.manual_seed(0)
    env.reset()
    acts = [torch.randint(0, n, (B,), device=DEV) for _ in range(10)]
    for a in acts[:4]:
        env.step(a)
    snap = env.state_dict()
    outs = [tuple(t.clone() if torch.is_tensor(t) else t for t in env.step(a)[:4]) for a in acts[4:]]
    env2 = MaxcutVecEnv(mg, n, B, max_step=7)
    env2.load_state_dict(snap)
    outs2 = [tuple(t.clone() if torch.is_tensor(t) else t for t in env2.step(a)[:4]) for a in acts[4:]]
    assert all(torch.equal(x, y) for o, p in zip(outs, outs2) for x, y in zip(o, p))
    sim = EnvMaxcut(mygraph=mg, device=DEV, num_nodes=n)
    ls = LocalSearch(sim, n)
    ls.reset(sim.generate_xs_randomly(B))
    torch.manual_seed(5)
    ls.random_search(num_iters=2, num_spin=4)
    sd = ls.state_dict()
    torch.manual_seed(6)
    a = ls.random_search(num_iters=2, num_spin=4)
    ls2 = LocalSearch(sim, n)
    ls2.load_state_dict(sd)
    torch.manual_seed(6)
    b = ls2.random_search(num_iters=2, num_spin=4)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
