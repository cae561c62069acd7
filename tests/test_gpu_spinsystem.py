"""SpinSystem (S2V/ECO/PECO surface) vs golden traces of the reference's batched PECO env run on a
shared +-1-weighted graph."""
import numpy as np
import pytest
import torch

from tests.gpu_util import DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
@pytest.mark.parametrize("cname", ["eco", "dense", "stag"])
def test_spinsystem_golden(golden, gname, cname):
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
    z = golden("spinsystem")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    tag = f"{gname}/{cname}"
    max_steps = int(z[f"{tag}/max_steps"])
    cfg = {"eco": dict(reward_signal=RewardSignal.BLS, norm_rewards=True, basin_reward=1.0 / n),
           "dense": dict(reward_signal=RewardSignal.DENSE, norm_rewards=False, basin_reward=None),
           "stag": dict(reward_signal=RewardSignal.CUSTOM_BLS, norm_rewards=False, basin_reward=0.25,
                        stag_punishment=0.125)}[cname]
    env = SpinSystem([tuple(int(v) for v in r) for r in g], n, 6, max_steps=max_steps,
                     observables=ECO_PECO_OBSERVABLES, spin_basis=SpinBasis.BINARY, device=DEV, **cfg)
    assert env.action_space.n == n and env.observation_space.shape == [n, 7]
    assert env.get_allowed_action_states() == (0, 1)
    obs = env.reset(spins=torch.from_numpy(z[f"{tag}/spins0"]))
    assert obs.shape == (6, 7 + n, n) and obs.dtype == torch.float32
    assert np.array_equal(env.max_local_reward_available_.cpu().numpy(), z[f"{tag}/max_local"])
    assert np.array_equal(obs[:, :7].cpu().numpy(), z[f"{tag}/obs0"])
    assert np.array_equal(obs[0, 7:].cpu().numpy(), z[f"{tag}/adj_rows"])
    assert np.array_equal(env.score.cpu().numpy(), z[f"{tag}/score0"])
    for t in range(max_steps):
        o, r, d = env.step(torch.from_numpy(z[f"{tag}/actions"][t]).to(DEV))
        assert np.array_equal(o[:, :7].cpu().numpy(), z[f"{tag}/obs"][t]), t
        assert np.array_equal(r.cpu().numpy(), z[f"{tag}/rew"][t]), t
        assert np.array_equal(d.cpu().numpy(), z[f"{tag}/done"][t])
        assert np.array_equal(env.score.cpu().numpy(), z[f"{tag}/score"][t])
        assert np.array_equal(env.get_best_cut().cpu().numpy(), z[f"{tag}/best_score"][t])
    assert np.array_equal(env.best_spins.cpu().numpy(), z[f"{tag}/best_spins"])
    with pytest.raises(NotImplementedError):
        env.step(torch.zeros(6, dtype=torch.int64, device=DEV))     # already done, like the reference


@pytest.mark.parametrize("gname", ["ba20", "er24", "ba40u"])
@pytest.mark.parametrize("cname", ["eco", "stag", "dense"])
def test_spinsystem_per_env_matrices_golden(golden, gname, cname):
    """The training form of the env (graph_generator=: a couplings matrix per env) against traces of the reference's PECO env
    on graphs drawn by the reference's own BA / ER generators -- BA's seed clique carries self-loops, which the reference
    counts in its own way (score change delta_a - 2 W_aa): observations (matrix rows included), rewards, scores, best."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
    z = golden("spinsystem_perenv")
    tag = f"{gname}/{cname}"
    m = z[f"{tag}/matrix"]
    B, n, _ = m.shape

    class Recorded:
        n_spins, biased, calls = n, False, 0

        def get(self):
            Recorded.calls += 1
            return torch.from_numpy(m).to(DEV)

    cfg = {"eco": dict(reward_signal=RewardSignal.BLS, norm_rewards=True, basin_reward=1.0 / n),
           "stag": dict(reward_signal=RewardSignal.CUSTOM_BLS, norm_rewards=False, basin_reward=0.25, stag_punishment=0.125),
           "dense": dict(reward_signal=RewardSignal.DENSE, norm_rewards=False, basin_reward=None)}[cname]
    env = SpinSystem(None, None, B, max_steps=2 * n, observables=ECO_PECO_OBSERVABLES, spin_basis=SpinBasis.BINARY, device=DEV,
                     graph_generator=Recorded(), **cfg)
    assert env.n_spins == n and Recorded.calls == 1
    obs = env.reset(spins=torch.from_numpy(z[f"{tag}/spins0"]))
    assert Recorded.calls == 2                                                  # a fresh draw per reset, as the reference
    assert np.array_equal(env.max_local_reward_available_.cpu().numpy(), z[f"{tag}/max_local"])
    assert np.array_equal(obs.cpu().numpy(), z[f"{tag}/obs0"])
    assert np.array_equal(env.score.cpu().numpy(), z[f"{tag}/score0"])
    assert np.array_equal(env.calculate_cut().cpu().numpy(), z[f"{tag}/score0"])
    for t in range(2 * n):
        o, r, d = env.step(torch.from_numpy(z[f"{tag}/actions"][t]).to(DEV))
        assert np.array_equal(o[:, :7].cpu().numpy(), z[f"{tag}/obs"][t]), t
        assert np.array_equal(r.cpu().numpy(), z[f"{tag}/rew"][t]), t
        assert np.array_equal(env.score.cpu().numpy(), z[f"{tag}/score"][t])
        assert np.array_equal(env.get_best_cut().cpu().numpy(), z[f"{tag}/best_score"][t])
    assert np.array_equal(o.cpu().numpy(), z[f"{tag}/last_obs"]) and bool(d.all())
    assert np.array_equal(env.best_spins.cpu().numpy(), z[f"{tag}/best_spins"])


def test_spinsystem_per_env_matrices_rejections():
    """Empty graphs are drawn again (spinsystem_PECO.py:164-169); matrices the integer gain cache cannot hold raise."""
    from rlsolver_amd.envs.spinsystem import SpinSystem
    n, B = 16, 5
    rng = np.random.RandomState(0)
    good = np.triu((rng.rand(B, n, n) < 0.3).astype(np.float32), 1)
    good = good + good.transpose(0, 2, 1)

    class Gen:
        n_spins, biased = n, False

        def __init__(self, seq):
            self.seq, self.calls = seq, 0

        def get(self):
            self.calls += 1
            return torch.from_numpy(self.seq[min(self.calls, len(self.seq)) - 1]).to(DEV)

    empty = good.copy()
    empty[3] = 0
    g = Gen([empty, empty, good])
    env = SpinSystem(None, None, B, graph_generator=g, device=DEV)
    assert g.calls == 3 and torch.equal(env.matrix, torch.from_numpy(good).to(DEV))
    for bad in (good * 0.5, np.triu(good)):
        with pytest.raises(ValueError):
            SpinSystem(None, None, B, graph_generator=Gen([bad.astype(np.float32)]), device=DEV)
    with pytest.raises(ValueError):
        SpinSystem(None, None, B, graph_generator=Gen([empty]), device=DEV)
    with pytest.raises(ValueError):
        SpinSystem([(0, 1, 1)], 2, B, graph_generator=Gen([good]), device=DEV)


def test_spinsystem_large_consistency():
    """Incremental gain cache == recomputation, score == cut, on a G22-sized unweighted graph."""
    from rlsolver_amd import ops
    from rlsolver_amd.envs.spinsystem import S2V_OBSERVABLES, SpinSystem
    from rlsolver_amd.graph import generate_gnm
    n, m, B = 2000, 19990, 96
    mg = generate_gnm(n, m, 22)
    torch.manual_seed(1)
    env = SpinSystem(mg, n, B, max_steps=64, observables=S2V_OBSERVABLES, device=DEV, include_adjacency=False)
    obs = env.reset()
    assert obs.shape == (B, 1, n)
    rng = np.random.RandomState(0)
    for t in range(64):
        a = torch.from_numpy(rng.randint(0, n, size=B)).to(DEV)
        obs, r, d = env.step(a)
    x = (env.state[:, 0, :] < 0).contiguous()              # BINARY convention: s = -1 <-> 1
    assert torch.equal(ops.maxcut_obj(env.graph, x).float(), env.score)
    assert torch.equal(ops.maxcut_delta_all(env.graph, x), env._delta)
    assert bool(d.all()) and torch.equal(env.best_score, torch.maximum(env.best_score, env.score))


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
@pytest.mark.parametrize("cname", ["eco", "dense", "stag"])
def test_spinsystem_single_env_f64_golden(golden, gname, cname):
    """the numpy env of ECO_S2V/src/envs/spinsystem.py:333-482 (float64,
    obs [7 + N, N]) -- the same HIP kernel instantiated for double, bit for bit against the reference's trace."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystemUnbiased
    z = golden("spinsystem_cpu")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    tag = f"{gname}/{cname}"
    T = int(z[f"{tag}/max_steps"])
    cfg = {"eco": dict(reward_signal=RewardSignal.BLS, norm_rewards=True, basin_reward=1.0 / n),
           "dense": dict(reward_signal=RewardSignal.DENSE, norm_rewards=False),
           "stag": dict(reward_signal=RewardSignal.CUSTOM_BLS, norm_rewards=False, basin_reward=0.25,
                        stag_punishment=0.125)}[cname]
    env = SpinSystemUnbiased([tuple(int(v) for v in r) for r in g], n, max_steps=T, observables=ECO_PECO_OBSERVABLES,
                             spin_basis=SpinBasis.BINARY, device=DEV, **cfg)
    assert env.max_local_reward_available == float(z[f"{tag}/max_local"])
    obs = env.reset(z[f"{tag}/spins0"])
    assert obs.dtype == np.float64 and obs.shape == (7 + n, n)
    assert np.array_equal(obs, z[f"{tag}/obs0"])
    assert env.score == float(z[f"{tag}/score0"])
    for t in range(T):
        o, r, d, info = env.step(int(z[f"{tag}/actions"][t]))
        assert info is None and isinstance(r, float) and isinstance(d, bool)
        assert np.array_equal(o[:7], z[f"{tag}/obs"][t]), t
        assert np.array_equal(env.get_immeditate_rewards_avaialable(), z[f"{tag}/gains"][t]), t
        assert r == z[f"{tag}/rew"][t], t
        assert d == bool(z[f"{tag}/done"][t])
        assert env.score == z[f"{tag}/score"][t] and env.best_score == z[f"{tag}/best_score"][t]
    assert np.array_equal(env.best_spins, z[f"{tag}/best_spins"])
    assert np.array_equal(o[7:], z[f"{tag}/adj_rows"])


def test_spinsystem_history_against_oracle_and_state_dict():
    """Visited-state memory at a size where revisits, hash pre-filter and the multi-word compare all occur
    (N = 130: 3 words per state; odd N: the scalar row path), against the dense oracle; then checkpoint /
    restore in the middle of an episode."""
    from oracle.oracle_spin import SpinSystemOracle
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
    from rlsolver_amd.graph import generate_gnm
    for n, m in ((130, 400), (67, 150)):
        rng = np.random.RandomState(n)
        mg = [(a, b, int(rng.choice([-1, 1]))) for a, b, _ in generate_gnm(n, m, 3)]
        W = np.zeros((n, n), np.float32)
        for a, b, w in mg:
            W[a, b] = W[b, a] = w
        B, T = 9, 60
        kw = dict(reward_signal=RewardSignal.BLS, norm_rewards=True, basin_reward=0.5, stag_punishment=0.25)
        env = SpinSystem(mg, n, B, max_steps=T, observables=ECO_PECO_OBSERVABLES, spin_basis=SpinBasis.BINARY,
                         device=DEV, **kw)
        ora = SpinSystemOracle(W, B, T, reward="BLS", norm_rewards=True, basin_reward=0.5, stag_punishment=0.25)
        s0 = (2 * rng.randint(0, 2, size=(B, n)) - 1).astype(np.float32)
        assert np.array_equal(env.reset(torch.from_numpy(s0))[:, :7].cpu().numpy(), ora.reset(s0))
        acts = rng.randint(0, n, size=(T, B))
        for t in range(2, T, 4):        # undo moves and 3-cycles of returns: plenty of revisits
            acts[t] = acts[t - 1]
        snap = None
        for t in range(T):
            if t == 20:
                snap = env.state_dict()
            o, r, d = env.step(torch.from_numpy(acts[t]).to(DEV))
            oo, rr, dd = ora.step(acts[t])
            assert np.array_equal(o[:, :7].cpu().numpy(), oo), (n, t)
            assert np.array_equal(r.cpu().numpy(), rr), (n, t)
        end = env.state_dict()
        env.load_state_dict(snap)
        for t in range(20, T):
            env.step(torch.from_numpy(acts[t]).to(DEV))
        again = env.state_dict()
        assert all(torch.equal(end[k], again[k]) if torch.is_tensor(end[k]) else end[k] == again[k] for k in end)


def test_spin_step_rejects_out_of_range_action():
    from rlsolver_amd.envs.spinsystem import S2V_OBSERVABLES, SpinSystem
    from rlsolver_amd.graph import generate_gnm
    env = SpinSystem(generate_gnm(40, 90, 1), 40, 4, max_steps=5, observables=S2V_OBSERVABLES, device=DEV,
                     include_adjacency=False)
    before = env.state.clone()
    _, r, _ = env.step(torch.tensor([3, 40, -1, 7], device=DEV))
    assert torch.isnan(r[1]) and torch.isnan(r[2]) and not torch.isnan(r[0]) and not torch.isnan(r[3])
    assert torch.equal(env.state[1:3], before[1:3]) and not torch.equal(env.state[0], before[0])


@pytest.mark.parametrize("n,dtype,basis_binary,adj", [(200, torch.float32, True, True), (37, torch.float32, False, True),
                                                       (64, torch.float64, True, True), (101, torch.float64, True, False),
                                                       (128, torch.float32, True, False)])
def test_observation_kernel_equals_the_reference_expression(n, dtype, basis_binary, adj):
    """get_observation = cat(state with row 0 mapped to the agent's spin basis, matrix.expand(B, N, N))
    (spinsystem_PECO.py:455) as one kernel: every element equal, for rows that are and are not multiples of 16 bytes,
    both dtypes, with and without the adjacency rows, into a fresh tensor and into a caller's buffer."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
    from rlsolver_amd.graph import generate_gnm
    rng = np.random.RandomState(n)
    mg = [(u, v, int(rng.choice([-1, 1]))) for u, v, _ in generate_gnm(n, 3 * n, 5)]
    B = 9
    env = SpinSystem(mg, n, B, max_steps=50, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS, norm_rewards=True,
                     spin_basis=SpinBasis.BINARY if basis_binary else SpinBasis.SIGNED, device=DEV, include_adjacency=adj,
                     dtype=dtype)
    env.reset()
    for t in range(3):
        obs, _, _ = env.step(torch.from_numpy(rng.randint(0, n, size=B)).to(DEV))
    want = env.state.clone()
    if basis_binary:
        want[:, 0, :] = (1 - want[:, 0, :]) / 2
    if adj:
        want = torch.cat((want, env.matrix.unsqueeze(0).expand(B, -1, -1)), dim=-2)
    assert obs.dtype == dtype and torch.equal(obs, want)
    buf = torch.full_like(want, 7.0)
    assert env.get_observation(out=buf) is buf and torch.equal(buf, want)
    with pytest.raises(ValueError):
        env.get_observation(out=buf[:, :-1].contiguous())


@pytest.mark.parametrize("edge_type", [1, 2, 3])
def test_rand_couplings_equal_their_restatement(edge_type):
    """rls_rand_couplings bit for bit against the numpy restatement of its draws (ER and BA, every edge type, a shard offset),
    plus the structure the reference's generators produce: symmetric 0 / +-1, ER zero diagonal, BA seed clique with its
    self-loops and exactly m earlier neighbours per later node, DISCRETE signs shared by the envs, RANDOM not."""
    from oracle import oracle_np as onp
    from rlsolver_amd.envs.util_envs_PECO import EdgeType, RandomBAGraphGenerator, RandomERGraphGenerator
    B, N, m = 70, 37, 4
    er = RandomERGraphGenerator(N, 0.3, EdgeType(edge_type), B, DEV, env_offset=5).get(seed=1234).cpu().numpy()
    assert np.array_equal(er, onp.rand_couplings_er(B, N, 0.3, edge_type, 1234, 5).astype(np.float32))
    ba = RandomBAGraphGenerator(N, m, EdgeType(edge_type), B, DEV, env_offset=5, dtype=torch.float64).get(seed=99).cpu().numpy()
    assert np.array_equal(ba, onp.rand_couplings_ba(B, N, m, edge_type, 99, 5))
    for mat in (er, ba):
        assert np.array_equal(mat, mat.transpose(0, 2, 1)) and set(np.unique(mat)) <= {-1.0, 0.0, 1.0}
    assert not np.diagonal(er, axis1=1, axis2=2).any()
    a = np.abs(ba)
    assert (a[:, :m + 1, :m + 1] == 1).all() and not np.diagonal(a, axis1=1, axis2=2)[:, m + 1:].any()
    assert all((np.tril(a[b], -1)[m + 1:].sum(1) == m).all() for b in range(B))
    same = all(np.array_equal(np.sign(ba[0]) * (a[0] * a[b]), np.sign(ba[b]) * (a[0] * a[b])) for b in range(B))
    assert same == (edge_type != 3)
    shard = RandomBAGraphGenerator(N, m, EdgeType(edge_type), 6, DEV, env_offset=5 + 64, dtype=torch.float64).get(seed=99).cpu().numpy()
    assert np.array_equal(shard, ba[64:70])                                   # a shard draws what the whole batch draws


def test_rand_couplings_distributions_and_training_env():
    """ER density = p; BA is preferential (old nodes collect more edges than late ones, degree ~ the reference's on the CPU
    within sampling error); an env built on the kernel generators steps, redraws its graphs at reset, and its resident gain
    cache equals s * (W s) on the drawn matrices after random steps."""
    from rlsolver_amd.envs.spinsystem import SpinSystem
    from rlsolver_amd.envs.util_envs_PECO import EdgeType, RandomBAGraphGenerator, RandomERGraphGenerator
    torch.manual_seed(5)
    B, N = 512, 64
    er = RandomERGraphGenerator(N, 0.15, EdgeType.UNIFORM, B, DEV).get()
    dens = float(er.sum() / (B * N * (N - 1)))
    assert abs(dens - 0.15) < 0.004
    ba = RandomBAGraphGenerator(N, 4, EdgeType.UNIFORM, B, DEV).get()
    deg = ba.sum(-1).mean(0).cpu().numpy()
    assert deg[:5].mean() > 2.5 * deg[-20:].mean() and abs(deg[-1] - 4.0) < 1e-6
    # mean row sums of the reference's own generator (torch.multinomial without replacement; N = 64, m = 4, 4096 graphs drawn on
    # the CPU when this test was written): node 0 21.9, node 5 16.4, node 16 8.45, node 32 5.72, node 48 4.62, node 63 4.0
    for node, want, tol in ((0, 21.9, 1.5), (5, 16.4, 1.3), (16, 8.45, 0.6), (32, 5.72, 0.4), (48, 4.62, 0.25)):
        assert abs(deg[node] - want) < tol, (node, deg[node])
    gg = RandomBAGraphGenerator(N, 4, EdgeType.DISCRETE, B, DEV)
    env = SpinSystem(None, None, B, max_steps=2 * N, graph_generator=gg, device=DEV)
    m0 = env.matrix.clone()
    for t in range(40):
        env.step(torch.randint(0, N, (B,), device=DEV))
    s = env.state[:, 0, :]
    assert torch.equal(env._delta.to(torch.float32), s * torch.einsum("bij,bj->bi", env.matrix, s))
    env.reset()
    assert not torch.equal(env.matrix, m0) and env.current_step == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_per_env_matrices_equal_the_shared_graph_env_on_one_graph(dtype):
    """The dense (matrix per env) and the CSR (one shared graph) forms of the step are the same env when every env holds the
    same zero-diagonal matrix: identical observations, rewards, scores, visited-state flags over a whole episode with
    revisits, in f32 and f64; and state_dict round-trips the dense env (matrix included)."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystem
    from rlsolver_amd.envs.util_envs_PECO import SetGraphGenerator
    rng = np.random.RandomState(4)
    n, B, T = 48, 9, 60
    up = np.triu((rng.rand(n, n) < 0.2) * rng.choice([-1, 1], size=(n, n)), 1)
    W = (up + up.T).astype(np.float64)
    mg = [(i, j, int(W[i, j])) for i in range(n) for j in range(i + 1, n) if W[i, j] != 0]
    kw = dict(max_steps=T, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.CUSTOM_BLS, spin_basis=SpinBasis.BINARY,
              stag_punishment=0.125, basin_reward=0.25, device=DEV, dtype=dtype)
    shared = SpinSystem(mg, n, B, **kw)
    dense = SpinSystem(None, None, B, graph_generator=SetGraphGenerator(np.broadcast_to(W, (B, n, n)).copy(), device=DEV), **kw)
    spins0 = torch.from_numpy(rng.randint(0, 2, size=(B, n)).astype(np.float64))
    o1, o2 = shared.reset(spins0), dense.reset(spins0)
    assert torch.equal(o1, o2) and torch.equal(shared.score, dense.score)
    assert torch.equal(shared.max_local_reward_available_, dense.max_local_reward_available_)
    g = torch.Generator().manual_seed(2)
    prev = None
    for t in range(T):
        a = torch.randint(0, n, (B,), generator=g)
        if t % 3 == 2:
            a = prev.clone()                                                   # undo: a revisited state
        prev = a
        if t == T // 2:
            snap = dense.state_dict()
        (o1, r1, d1), (o2, r2, d2) = shared.step(a.to(DEV)), dense.step(a.to(DEV))
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), t
        assert torch.equal(shared._visited_new, dense._visited_new) and torch.equal(shared.best_spins, dense.best_spins)
    assert torch.equal(shared.calculate_cut(), dense.calculate_cut())
    end_obs = o2.clone()
    dense.load_state_dict(snap)
    g2 = torch.Generator().manual_seed(2)
    acts = []
    for t in range(T):
        a = torch.randint(0, n, (B,), generator=g2)
        if t % 3 == 2:
            a = acts[-1].clone()
        acts.append(a)
    for t in range(T // 2, T):
        o3, _, _ = dense.step(acts[t].to(DEV))
    assert torch.equal(o3, end_obs)


def test_factory_make_builds_the_training_and_validation_envs():
    """ising_env.make("SpinSystem", generator, max_steps, **env_args, device=, num_envs=) as train_PECO.py:33-86 calls it; the
    validation generator hands back networkx's seeded graphs; configurations the device env does not build raise."""
    import networkx as nx
    from rlsolver_amd.envs import spinsystem as ss
    from rlsolver_amd.envs.util_envs_PECO import EdgeType, RandomBAGraphGenerator, ValidationGraphGenerator
    n, B = 20, 16
    env_args = dict(observables=ss.ECO_PECO_OBSERVABLES, reward_signal=ss.RewardSignal.BLS, extra_action=ss.ExtraAction.NONE,
                    optimisation_target=ss.OptimisationTarget.CUT, spin_basis=ss.SpinBasis.BINARY, norm_rewards=True, memory_length=None,
                    horizon_length=None, stag_punishment=None, basin_reward=1.0 / n, reversible_spins=True)
    train = ss.make("SpinSystem", RandomBAGraphGenerator(n, 4, EdgeType.DISCRETE, B, DEV), 2 * n, **env_args, device=DEV, num_envs=B)
    assert train.extra_action == ss.ExtraAction.NONE and train.num_envs == B and train.max_steps == 2 * n
    obs, rew, done = train.step(torch.zeros(B, dtype=torch.int64, device=DEV))
    assert obs.shape == (B, 7 + n, n) and rew.shape == (B,) and not bool(done.any())
    vg = ValidationGraphGenerator(DEV, n_spins=n, num_envs=4, seed=30, graph_type="BA")
    test = ss.make("SpinSystem", vg, 2 * n, **env_args, device=DEV, num_envs=4)
    want = np.stack([nx.to_numpy_array(nx.barabasi_albert_graph(n, 4, seed=30 + k)) for k in range(4)]).astype(np.float32)
    assert np.array_equal(test.matrix.cpu().numpy(), want) and np.array_equal(vg.get().cpu().numpy(), want)
    s = test.state[:, 0, :]
    assert torch.equal(test.calculate_cut(), (0.25 * (test.matrix.sum((1, 2)) - (s * torch.einsum("bij,bj->bi", test.matrix, s)).sum(1))))
    for bad in (dict(extra_action=ss.ExtraAction.PASS), dict(optimisation_target=ss.OptimisationTarget.ENERGY), dict(memory_length=5),
                dict(reversible_spins=False)):
        with pytest.raises(NotImplementedError):
            ss.make("SpinSystem", vg, 2 * n, **{**env_args, **bad}, device=DEV, num_envs=4)
    with pytest.raises(NotImplementedError):
        ss.make("Other", vg)


def test_single_instance_env_on_a_graph_generator():
    """SpinSystemUnbiased (float64, one env) with graph_generator=: a fresh [N, N] graph per reset, the same trajectory as the
    env built on that graph's edge list."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, RewardSignal, SpinBasis, SpinSystemUnbiased
    n = 30
    rng = np.random.RandomState(6)

    class Gen:
        n_spins, biased, calls = n, False, 0

        def get(self):
            Gen.calls += 1
            r = np.random.RandomState(100 + Gen.calls)
            up = np.triu((r.rand(n, n) < 0.25) * r.choice([-1, 1], size=(n, n)), 1)
            return (up + up.T).astype(np.float64)

    kw = dict(max_steps=2 * n, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS, spin_basis=SpinBasis.BINARY,
              norm_rewards=True, basin_reward=1.0 / n, device=DEV)
    env = SpinSystemUnbiased(None, None, graph_generator=Gen(), **kw)
    assert Gen.calls == 1 and env.n_spins == n
    spins = rng.randint(0, 2, size=n).astype(np.float64)
    obs = env.reset(spins)
    W = env.matrix
    assert Gen.calls == 2 and obs.shape == (7 + n, n) and np.array_equal(obs[7:], W) and obs.dtype == np.float64
    ref = SpinSystemUnbiased([(i, j, int(W[i, j])) for i in range(n) for j in range(i + 1, n) if W[i, j]], n, **kw)
    obs2 = ref.reset(spins)
    assert np.array_equal(obs, obs2) and env.max_local_reward_available == ref.max_local_reward_available
    for t in range(2 * n):
        a = int(rng.randint(n))
        (o1, r1, d1, _), (o2, r2, d2, _) = env.step(a), ref.step(a)
        assert np.array_equal(o1, o2) and r1 == r2 and d1 == d2
    assert env.best_score == ref.best_score and np.array_equal(env.best_spins, ref.best_spins)
    # the same env through the factory, the way train_ECO.py:83-92 builds it (keywords of spinsystem.py:30-46, PASS default)
    from rlsolver_amd.envs import spinsystem as ss
    fenv = ss.make("SpinSystem", Gen(), 2 * n, observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS,
                   extra_action=ss.ExtraAction.PASS, optimisation_target=ss.OptimisationTarget.CUT, spin_basis=SpinBasis.BINARY,
                   norm_rewards=True, memory_length=None, horizon_length=None, stag_punishment=None, basin_reward=1.0 / n,
                   reversible_spins=True, device=DEV, if_greedy=False)
    assert isinstance(fenv, SpinSystemUnbiased) and fenv.n_actions == n + 1 and fenv.reset().shape == (7 + n + 1, n + 1)
    # calculate_cut on the env's own and on foreign spins (spinsystem.py:601-607), set_seed (agents/util.py:26-30)
    W = fenv.matrix
    own = 1 - 2 * fenv.get_observation()[0, :n]                     # BINARY row 0 = (1 - s) / 2
    assert fenv.calculate_cut() == 0.25 * np.sum(W * (1 - np.outer(own, own))) == fenv.calculate_score()
    other = rng.randint(0, 2, size=n).astype(np.float64)
    so = 2 * other - 1
    assert fenv.calculate_cut(other) == 0.25 * np.sum(W * (1 - np.outer(so, so)))
    with pytest.raises(Exception):
        fenv.calculate_cut(so)                                       # signed spins into a BINARY env
    fenv.set_seed(11)
    assert fenv.seed() == 11


@pytest.mark.parametrize("cname", ["pass", "mem3", "pass_mem4_stag"])
def test_spinsystem_options_golden(golden, cname):
    """spinsystem_options.npz (the reference's numpy env with ExtraAction.PASS -- its default -- and / or a finite
    memory_length): SpinSystemUnbiased on the HIP env reproduces the padded observation and state, rewards, scores, best and
    best OBSERVABLE score bit for bit over 48 steps with passes, undo moves and revisits."""
    from rlsolver_amd.envs.spinsystem import ECO_PECO_OBSERVABLES, ExtraAction, RewardSignal, SpinBasis, SpinSystemUnbiased
    z = golden("spinsystem_options")
    g = z["graph"]
    n = int(g[:, :2].max()) + 1
    cfg = {"pass": dict(reward_signal=RewardSignal.BLS, norm_rewards=True, basin_reward=1.0 / n, extra_action=ExtraAction.PASS),
           "mem3": dict(reward_signal=RewardSignal.BLS, norm_rewards=False, memory_length=3),
           "pass_mem4_stag": dict(reward_signal=RewardSignal.CUSTOM_BLS, norm_rewards=False, basin_reward=0.25, stag_punishment=0.125,
                                  extra_action=ExtraAction.PASS, memory_length=4)}[cname]
    T = int(z["max_steps"])
    env = SpinSystemUnbiased([tuple(int(v) for v in r) for r in g], n, max_steps=T, observables=ECO_PECO_OBSERVABLES,
                             spin_basis=SpinBasis.BINARY, device=DEV, **cfg)
    na = int(z[f"{cname}/n_actions"])
    assert env.n_actions == na and env.action_space.n == na
    obs = env.reset(z[f"{cname}/spins0"])
    assert obs.shape == (7 + na, na) and np.array_equal(obs, z[f"{cname}/obs0"])
    assert np.array_equal(env.state, z[f"{cname}/state0"])
    for t in range(T):
        o, r, d, info = env.step(int(z[f"{cname}/actions"][t]))
        assert np.array_equal(o[:7], z[f"{cname}/obs"][t]), t
        assert np.array_equal(env.state, z[f"{cname}/state"][t]), t
        assert r == z[f"{cname}/rew"][t] and d == bool(z[f"{cname}/done"][t]), t
        assert env.score == z[f"{cname}/score"][t] and env.best_score == z[f"{cname}/best_score"][t], t
        assert env.best_obs_score == z[f"{cname}/best_obs_score"][t], t
    assert np.array_equal(o[7:], z[f"{cname}/adj_rows"]) and np.array_equal(env.best_spins, z[f"{cname}/best_spins"])


def test_spinsystem_options_against_oracle_random_and_refusals():
    """PASS + finite memory at a size with three words per packed state (N = 130, the PASS bit in the third), both
    visited-state rewards, random actions with passes and undo moves, against the float64 restatement; the options the
    reference cannot run are refused."""
    from oracle.oracle_spin import SpinSystemOracleF64
    from rlsolver_amd.envs.spinsystem import (ECO_PECO_OBSERVABLES, ExtraAction, RewardSignal, SpinBasis, SpinSystem, SpinSystemFactory,
                                              SpinSystemUnbiased)
    from rlsolver_amd.graph import generate_gnm
    for n, m, M in ((130, 420, 5), (64, 200, 2), (127, 380, 7)):
        rng = np.random.RandomState(n + M)
        mg = [(a, b, int(rng.choice([-1, 1]))) for a, b, _ in generate_gnm(n, m, 4)]
        W = np.zeros((n, n))
        for a, b, w in mg:
            W[a, b] = W[b, a] = w
        T = 70
        env = SpinSystemUnbiased(mg, n, max_steps=T, observables=ECO_PECO_OBSERVABLES, spin_basis=SpinBasis.BINARY, device=DEV,
                                 reward_signal=RewardSignal.BLS, norm_rewards=True, basin_reward=0.5, stag_punishment=0.25,
                                 extra_action=ExtraAction.PASS, memory_length=M)
        ora = SpinSystemOracleF64(W, T, reward="BLS", norm_rewards=True, basin_reward=0.5, stag_punishment=0.25, extra_pass=True,
                                  memory_length=M)
        s0 = (2 * rng.randint(0, 2, size=n) - 1).astype(np.float64)
        assert np.array_equal(env.reset(s0), ora.reset(s0))
        prev = 0
        for t in range(T):
            a = int(rng.randint(0, n))
            a = prev if t % 4 == 3 else (n if t % 7 == 2 else a)
            prev = a
            o, r, d, _ = env.step(a)
            wo, wr, wd = ora.step(a)
            assert np.array_equal(o, wo) and r == wr and d == wd, (n, M, t)
            assert env.best_obs_score == ora.best_obs_score and env.best_score == ora.best_score
    mg = [(0, 1, 1), (1, 2, -1)]
    with pytest.raises(NotImplementedError):
        SpinSystemUnbiased(mg, 3, extra_action=ExtraAction.RANDOMISE, device=DEV)
    with pytest.raises(ValueError):
        SpinSystemUnbiased(mg, 3, memory_length=1, device=DEV)
    from rlsolver_amd.envs.spinsystem import OptimisationTarget
    from rlsolver_amd.envs.util_envs_PECO import SetGraphGenerator
    batch = SetGraphGenerator(np.zeros((2, 3, 3), dtype=np.float32) + np.array([[0, 1, 0], [1, 0, -1], [0, -1, 0]], dtype=np.float32),
                              device=DEV)
    assert batch.edge_type.name == "DISCRETE" and batch.num_envs == 2
    for bad in (dict(extra_action=ExtraAction.PASS), dict(memory_length=4)):
        with pytest.raises(NotImplementedError):      # the batched factory: the reference's own constructor raises for these
            SpinSystemFactory.get(graph_generator=batch, **{**dict(extra_action=ExtraAction.NONE, optimisation_target=OptimisationTarget.CUT), **bad})
    with pytest.raises(NotImplementedError):
        SetGraphGenerator(np.zeros((2, 3, 3)), biases=[np.zeros(3)] * 2)
    benv = SpinSystemFactory.get(graph_generator=batch, extra_action=ExtraAction.NONE, optimisation_target=OptimisationTarget.CUT, device=DEV)
    benv.reset()
    sp = torch.tensor([[1., -1., 1.], [1., 1., 1.]], device=DEV)
    Wb = benv.matrix
    want = 0.25 * (Wb.sum((1, 2)) - (sp * torch.einsum("bij,bj->bi", Wb, sp)).sum(1))
    assert torch.equal(benv.calculate_cut(sp), want) and torch.equal(benv.calculate_score(sp), want)
    env = SpinSystem(mg, 3, 2, device=DEV)
    _, r, _ = env.step(torch.tensor([3, 0], device=DEV))             # without PASS action N stays out of range: NaN, env untouched
    assert bool(torch.isnan(r[0])) and not bool(torch.isnan(r[1]))


def test_calculate_cut_of_foreign_spins_on_a_shared_graph():
    """SpinSystem.calculate_cut(spins) (spinsystem_PECO.py:601-607) on the shared-graph form = the reference's expression."""
    from rlsolver_amd.envs.spinsystem import SpinSystem
    from rlsolver_amd.graph import generate_gnm
    n, B = 70, 9
    rng = np.random.RandomState(3)
    mg = [(a, b, int(rng.choice([-1, 1, 2]))) for a, b, _ in generate_gnm(n, 300, 2)]
    env = SpinSystem(mg, n, B, device=DEV)
    env.reset()
    W = env.matrix
    sp = torch.from_numpy(2.0 * rng.randint(0, 2, size=(B, n)) - 1).to(DEV, torch.float32)
    want = 0.25 * (torch.matmul(W, sp.unsqueeze(-1)).squeeze(-1) * -sp).sum(-1) + 0.25 * W.sum()
    assert torch.equal(env.calculate_cut(sp), want)
    own = env.state[:, 0, :]
    assert torch.equal(env.calculate_cut(), env.calculate_cut(own))
    env.set_seed(5)
    assert env.seed() == 5


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
def test_inference_twin_golden(golden, gname):
    """spinsystem_inference.npz, captured from ECO_S2V/src/envs/inference_network_env.py built as inference_PECO.py:84-99 builds
    it: step() -> (obs, done), best score / spins seeded from the best env of the batch, get_best_cut() 0-dim before a step."""
    from rlsolver_amd.envs import inference_network_env as inf
    from rlsolver_amd.envs.util_envs_PECO import SetGraphGenerator
    z = golden("spinsystem_inference")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    W = np.zeros((n, n), np.float32)
    W[g[:, 0], g[:, 1]] = g[:, 2]
    W[g[:, 1], g[:, 0]] = g[:, 2]
    T, B = int(z[f"{gname}/max_steps"]), z[f"{gname}/spins0"].shape[0]
    gg = SetGraphGenerator(torch.from_numpy(W).to(DEV), device=DEV)
    assert gg.n_spins == n and gg.num_envs is None and gg.get().shape == (n, n)
    env = inf.SpinSystemFactory.get(gg, T, observables=inf.ECO_PECO_OBSERVABLES, reward_signal=inf.RewardSignal.BLS,
                                    extra_action=inf.ExtraAction.NONE, optimisation_target=inf.OptimisationTarget.CUT,
                                    spin_basis=inf.SpinBasis.BINARY, norm_rewards=True, memory_length=None, horizon_length=None,
                                    stag_punishment=None, basin_reward=1.0 / n, reversible_spins=True, device=DEV, num_envs=B,
                                    if_greedy=False, use_tensor_core=False)
    assert env.num_envs == B and env.n_spins == n and env.max_steps == T and env.device == DEV
    obs = env.reset(torch.from_numpy(z[f"{gname}/spins0"]))
    assert obs.shape == (B, 7 + n, n) and np.array_equal(obs[:, :7].cpu().numpy(), z[f"{gname}/obs0"])
    assert np.array_equal(env.score.cpu().numpy(), z[f"{gname}/score0"])
    bc0 = env.get_best_cut()
    assert bc0.dim() == 0 and bc0.item() == float(z[f"{gname}/best_cut0"])
    assert np.array_equal(env.best_spins.cpu().numpy(), z[f"{gname}/best_spins0"])
    for t in range(T):
        res = env.step(torch.from_numpy(z[f"{gname}/actions"][t]).to(DEV))
        assert len(res) == 2
        o, d = res
        assert np.array_equal(o[:, :7].cpu().numpy(), z[f"{gname}/obs"][t]), t
        assert d.dtype == torch.bool and np.array_equal(d.cpu().numpy(), z[f"{gname}/done"][t])
        assert np.array_equal(env.score.cpu().numpy(), z[f"{gname}/score"][t])
        assert np.array_equal(env.get_best_cut().cpu().numpy(), z[f"{gname}/best_score"][t])
    assert np.array_equal(o[:, 7:].cpu().numpy(), np.broadcast_to(z[f"{gname}/adj_rows"], (B, n, n)))
    assert np.array_equal(env.best_spins.cpu().numpy(), z[f"{gname}/best_spins"])
    with pytest.raises(NotImplementedError):
        env.step(torch.zeros(B, dtype=torch.int64, device=DEV))                 # already done
    for bad in (dict(use_tensor_core=True), dict(extra_action=inf.ExtraAction.PASS), dict(memory_length=3)):
        with pytest.raises(NotImplementedError):
            inf.SpinSystemFactory.get(gg, T, **{**dict(extra_action=inf.ExtraAction.NONE, optimisation_target=inf.OptimisationTarget.CUT,
                                                      device=DEV, num_envs=B), **bad})


def test_inference_twin_against_oracle_at_size():
    """The same env at a size with several words per packed row and a larger batch, greedy and random actions, against the
    oracle's inference flavour (pinned by the fixture above)."""
    from oracle.oracle_spin import SpinSystemOracle
    from rlsolver_amd.envs import inference_network_env as inf
    from rlsolver_amd.envs.util_envs_PECO import SetGraphGenerator
    from rlsolver_amd.graph import generate_gnm
    n, B, T = 150, 40, 60
    rng = np.random.RandomState(8)
    W = np.zeros((n, n), np.float32)
    for a, b, _ in generate_gnm(n, 600, 3):
        W[a, b] = W[b, a] = rng.choice([-1, 1])
    env = inf.SpinSystemFactory.get(SetGraphGenerator(torch.from_numpy(W), device=DEV), T, extra_action=inf.ExtraAction.NONE,
                                    optimisation_target=inf.OptimisationTarget.CUT, spin_basis=inf.SpinBasis.BINARY,
                                    reward_signal=inf.RewardSignal.BLS, norm_rewards=True, device=DEV, num_envs=B)
    ora = SpinSystemOracle(W, B, T, reward="BLS", norm_rewards=True, inference=True)
    s0 = env.state[:, 0, :].cpu().numpy().copy()
    assert np.array_equal(env.get_observation()[:, :7].cpu().numpy(), ora.reset(s0))
    assert env.get_best_cut().item() == ora.best_score[0] and np.array_equal(env.best_spins.cpu().numpy(), ora.best_spins)
    for t in range(T):
        a = rng.randint(0, n, size=B)
        if t % 3 == 1:
            a = ora.state[:, 1].argmax(-1)
        o, d = env.step(torch.from_numpy(a).to(DEV))
        wo, wd = ora.step(a)
        assert np.array_equal(o[:, :7].cpu().numpy(), wo) and np.array_equal(d.cpu().numpy(), wd), t
        assert np.array_equal(env.get_best_cut().cpu().numpy(), ora.best_score)
    assert np.array_equal(env.best_spins.cpu().numpy(), ora.best_spins)


def _s2v_case(cname, n):
    from rlsolver_amd.envs.spinsystem import (ECO_PECO_OBSERVABLES, S2V_OBSERVABLES, ExtraAction, OptimisationTarget, RewardSignal,
                                              SpinBasis)
    E, C = OptimisationTarget.ENERGY, OptimisationTarget.CUT
    return {
        "s2v": dict(observables=S2V_OBSERVABLES, reward_signal=RewardSignal.DENSE, extra_action=ExtraAction.NONE, optimisation_target=C,
                    spin_basis=SpinBasis.BINARY, norm_rewards=True, reversible_spins=False),
        "eco_irreversible": dict(observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS, extra_action=ExtraAction.NONE,
                                 optimisation_target=C, spin_basis=SpinBasis.SIGNED, norm_rewards=True, basin_reward=1.0 / n,
                                 reversible_spins=False),
        "defaults": dict(),                       # SpinSystemFactory.get(gg, max_steps): DENSE, PASS, ENERGY, SIGNED, reversible
        "energy_bls_mem": dict(observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS, extra_action=ExtraAction.PASS,
                               optimisation_target=E, spin_basis=SpinBasis.BINARY, norm_rewards=True, basin_reward=1.0 / n, memory_length=3),
        "energy_custom_stag": dict(observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.CUSTOM_BLS, extra_action=ExtraAction.NONE,
                                   optimisation_target=E, spin_basis=SpinBasis.SIGNED, norm_rewards=False, basin_reward=0.25,
                                   stag_punishment=0.125),
        "energy_irreversible": dict(observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.DENSE, extra_action=ExtraAction.NONE,
                                    optimisation_target=E, spin_basis=SpinBasis.BINARY, norm_rewards=True, reversible_spins=False),
        "energy_isolated": dict(observables=ECO_PECO_OBSERVABLES, reward_signal=RewardSignal.BLS, extra_action=ExtraAction.NONE,
                                optimisation_target=E, spin_basis=SpinBasis.SIGNED, norm_rewards=False),
    }[cname]


@pytest.mark.parametrize("via", ["shared_graph", "generator"])
@pytest.mark.parametrize("cname", ["s2v", "eco_irreversible", "defaults", "energy_bls_mem", "energy_custom_stag", "energy_irreversible",
                                   "energy_isolated"])
def test_spinsystem_s2v_energy_golden(golden, cname, via):
    """spinsystem_s2v.npz: traces of the reference's numpy env with IRREVERSIBLE spins (S2V-DQN's env, train_S2V.py:37-47) and with
    OptimisationTarget.ENERGY (the default of SpinSystemFactory.get), through ``ising_env.make`` exactly as the training scripts
    call it (a generator -> per-env couplings on the device) and through the class on a shared graph: observation, state, reward,
    done (early, once no spin is +1), scores, best / best observable score, the maximum local reward (over the NONZERO entries:
    negative on the positive graph with an isolated node), immediate rewards, energy and cut -- bit for bit."""
    from rlsolver_amd.envs import spinsystem as sp
    z = golden("spinsystem_s2v")
    g = z["graph_isolated"] if cname == "energy_isolated" else z["graph"]
    n = 12 if cname == "energy_isolated" else int(g[:, :2].max()) + 1
    cfg = _s2v_case(cname, n)
    T = int(z[f"{cname}/max_steps"])
    if via == "generator":
        W = np.zeros((n, n))
        for a, b, w in g:
            W[a, b] = W[b, a] = w

        class Fixed:
            n_spins, biased = n, False

            def get(self, with_padding=False):
                return W.copy()
        env = sp.make("SpinSystem", Fixed(), T, device=DEV, **cfg)
    else:
        full = dict(extra_action=sp.ExtraAction.PASS, optimisation_target=sp.OptimisationTarget.ENERGY) if cname == "defaults" else cfg
        env = sp.SpinSystemUnbiased([tuple(int(v) for v in r) for r in g], n, max_steps=T, device=DEV, **full)
    assert isinstance(env, sp.SpinSystemUnbiased)
    irreversible = cfg.get("reversible_spins", True) is False
    assert env.reversible_spins == (not irreversible)
    R, na = len(env.observables), int(z[f"{cname}/n_actions"])
    assert env.n_actions == na
    obs = env.reset() if irreversible else env.reset(z[f"{cname}/spins0"])
    assert env.max_local_reward_available == float(z[f"{cname}/max_local"])
    allowed = env.get_allowed_action_states()
    assert np.array_equal(np.asarray(allowed).reshape(-1), z[f"{cname}/allowed"]) and isinstance(allowed, int) == irreversible
    assert obs.shape == (R + na, na) and np.array_equal(obs, z[f"{cname}/obs0"])
    assert np.array_equal(env.state, z[f"{cname}/state0"]) and env.score == float(z[f"{cname}/score0"])
    assert np.array_equal(env.get_immeditate_rewards_avaialable(), z[f"{cname}/imm0"])
    acts = z[f"{cname}/actions"]
    for t, a in enumerate(acts):
        o, r, d, info = env.step(int(a))
        assert np.array_equal(o[:R], z[f"{cname}/obs"][t]), t
        assert np.array_equal(env.state, z[f"{cname}/state"][t]), t
        assert r == z[f"{cname}/rew"][t] and d == bool(z[f"{cname}/done"][t]), t
        assert env.score == z[f"{cname}/score"][t] and env.best_score == z[f"{cname}/best_score"][t], t
        assert env.best_obs_score == z[f"{cname}/best_obs_score"][t], t
    assert d and (len(acts) < T) == irreversible
    assert np.array_equal(o[R:], z[f"{cname}/adj_rows"]) and np.array_equal(env.best_spins, z[f"{cname}/best_spins"])
    assert np.array_equal(env.get_immeditate_rewards_avaialable(), z[f"{cname}/imm_end"])
    if f"{cname}/energy_end" in z.files:
        assert env.calculate_energy() == float(z[f"{cname}/energy_end"]) and env.calculate_cut() == float(z[f"{cname}/cut_end"])
        assert env.calculate_score() == env.score
        with pytest.raises(NotImplementedError):
            env.get_best_cut()                                          # facts/energy_get_best_cut
        assert np.array_equal(env.matrix, o[R:R + n, :n])


def test_spinsystem_s2v_energy_against_oracle_random():
    """ENERGY / irreversible at sizes past one packed word with the visited-state rewards and a finite memory, random actions,
    against the float64 restatement; the batched factory keeps refusing what the reference's batched env cannot run."""
    from oracle.oracle_spin import SpinSystemOracleF64
    from rlsolver_amd.envs import spinsystem as sp
    from rlsolver_amd.graph import generate_gnm
    for n, m, M, rev in ((130, 420, 5, True), (70, 200, None, False), (127, 380, 3, False)):
        rng = np.random.RandomState(n + (M or 0))
        mg = [(a, b, int(rng.choice([-1, 1, 2]))) for a, b, _ in generate_gnm(n, m, 4)]
        W = np.zeros((n, n))
        for a, b, w in mg:
            W[a, b] = W[b, a] = w
        T = 90
        env = sp.SpinSystemUnbiased(mg, n, max_steps=T, observables=sp.ECO_PECO_OBSERVABLES, spin_basis=sp.SpinBasis.BINARY, device=DEV,
                                    reward_signal=sp.RewardSignal.CUSTOM_BLS, norm_rewards=True, basin_reward=0.5, stag_punishment=0.25,
                                    extra_action=sp.ExtraAction.PASS, memory_length=M, reversible_spins=rev,
                                    optimisation_target=sp.OptimisationTarget.ENERGY)
        ora = SpinSystemOracleF64(W, T, reward="CUSTOM_BLS", norm_rewards=True, basin_reward=0.5, stag_punishment=0.25, extra_pass=True,
                                  memory_length=M, target="ENERGY", reversible=rev)
        s0 = (2 * rng.randint(0, 2, size=n) - 1).astype(np.float64)
        assert np.array_equal(env.reset(s0 if rev else None), ora.reset(s0 if rev else None))
        assert env.max_local_reward_available == ora.max_local
        prev = 0
        for t in range(T):
            a = int(rng.randint(0, n))
            a = prev if t % 4 == 3 else (n if t % 7 == 2 else a)
            prev = a
            o, r, d, _ = env.step(a)
            wo, wr, wd = ora.step(a)
            assert np.array_equal(o, wo) and r == wr and d == wd, (n, M, t)
            assert env.score == ora.score and env.best_obs_score == ora.best_obs_score and env.best_score == ora.best_score
    from rlsolver_amd.envs.util_envs_PECO import SetGraphGenerator
    batch = SetGraphGenerator(np.zeros((2, 3, 3), dtype=np.float32) + np.array([[0, 1, 0], [1, 0, -1], [0, -1, 0]], dtype=np.float32),
                              device=DEV)
    for bad in (dict(optimisation_target=sp.OptimisationTarget.ENERGY), dict(reversible_spins=False)):
        with pytest.raises(NotImplementedError):
            sp.SpinSystemFactory.get(graph_generator=batch, **{**dict(extra_action=sp.ExtraAction.NONE,
                                                                    optimisation_target=sp.OptimisationTarget.CUT), **bad})
