"""Rank-count invariance of every seeded entry point ("per-env RNG keyed by global env id, not rank";
DESIGN.md section 8).  One process, one GPU: the batch is run once whole and once as two half-batches whose objects carry
``env_offset = 0`` and ``B / 2`` -- the halves must equal the corresponding rows of the whole batch BIT FOR BIT.

The few statistics the reference takes over the WHOLE batch (the local search's per-node weight range, the PECO reset's
redraw flag, MCPG's stop rule / mean / best-and-worst incumbent / policy-gradient sums) go through the objects' exchange hook:
the whole-batch run records them on a tape, the halves replay the tape -- what the collectives of a real sharded run deliver
(tests/test_gpu_two_process.py runs that real thing: two processes sharing cuda:0 over a gloo group)."""
import types

import numpy as np
import pytest
import torch

from gpu_util import DEV, gnm_arr
from rlsolver_amd.graph import generate_gnm

pytestmark = pytest.mark.gpu


class StatTape:
    """stat_hook of rlsolver_amd.seeding.Sharded: record (the object's batch IS the whole batch: local = global) or replay."""

    def __init__(self):
        self.items = []

    def recorder(self):
        def hook(kind, arg):
            if kind == "best":
                vs, row_of, off, maximize = arg
                li = (vs == vs.max()).nonzero()[0, 0]                     # FIRST maximum
                v = vs[li] if maximize else -vs[li]
                out = (v.clone(), (li + off).clone(), None if row_of is None else row_of(li).clone())
            else:
                out = arg.clone()
            self.items.append((kind, out))
            return out if kind == "best" else arg
        return hook

    def replayer(self):
        pos = [0]

        def hook(kind, arg):
            k, out = self.items[pos[0]]
            pos[0] += 1
            assert k == kind, f"exchange order differs: recorded {k}, asked {kind}"
            return tuple(None if o is None else o.clone() for o in out) if kind == "best" else out.clone()
        hook.pos = pos
        return hook


def _env(n, m, gseed, offset=0, bidir=False, **kw):
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    return EnvMaxcut(mygraph=generate_gnm(n, m, gseed), device=DEV, if_bidirectional=bidir, num_nodes=n, env_offset=offset, **kw)


@pytest.mark.parametrize("form", ["fused", "rounds", "decomposed"])
@pytest.mark.parametrize("n,m,B", [(200, 900, 256), (512, 3000, 1024), (203, 700, 192)])
def test_generate_and_local_search_halves_equal_whole(form, n, m, B):
    """generate_xs_randomly -> local_search_inplace: rows [0, B/2) and [B/2, B) computed by two objects with env_offset 0 and
    B/2 are the rows of the whole batch, in all three forms of the local search (the fused kernel, the threshold / round
    kernels, K2 + K6 + K5 with the kernels' draws as a tensor)."""
    def configure(env):
        env.fused_local_search = form != "decomposed"
        env.force_ls_fused = form == "fused"
        env.force_ls_rounds = form == "rounds"
        return env

    def run(env, count, hook):
        env.stat_hook = hook
        torch.manual_seed(11)
        xs = env.generate_xs_randomly(count)
        x0 = xs.clone()
        xs, vs = env.local_search_inplace(xs, torch.empty(()), num_iters=4, num_spin=6)
        return x0, xs, vs

    tape = StatTape()
    w0, wx, wv = run(configure(_env(n, m, 3)), B, tape.recorder())
    assert len(tape.items) == 1 and tape.items[0][0] == "minmax"
    h = B // 2
    for off in (0, h):
        x0, xs, vs = run(configure(_env(n, m, 3, offset=off)), h, tape.replayer())
        assert torch.equal(x0, w0[off:off + h]), "generate_xs_randomly is not keyed by the global env id"
        assert torch.equal(xs, wx[off:off + h]) and torch.equal(vs, wv[off:off + h])
    # and the search did something: the halves are not trivially equal
    assert not torch.equal(w0, wx) and (wv >= 0).all()


def test_local_search_forms_agree_under_one_seed():
    """The decomposed path draws the kernels' own normals now (rls_maxcut_ls_normals), so for one seed all three forms give
    the same rows (it used torch.randn: not keyed by the env, and a different search)."""
    n, m, B = 256, 1500, 512
    out = []
    for form in ("fused", "rounds", "decomposed"):
        env = _env(n, m, 5, offset=1000)
        env.fused_local_search, env.force_ls_fused, env.force_ls_rounds = form != "decomposed", form == "fused", form == "rounds"
        torch.manual_seed(2)
        xs = env.generate_xs_randomly(B)
        out.append(env.local_search_inplace(xs, torch.empty(()), num_iters=5, num_spin=7))
    for xs, vs in out[1:]:
        assert torch.equal(xs, out[0][0]) and torch.equal(vs, out[0][1])


def test_private_seed_stream_is_independent_of_torch():
    """``seed=`` gives a class its own stream: same rows whatever torch's generator holds; without it, torch.manual_seed rules."""
    a = _env(100, 300, 1, seed=99)
    torch.manual_seed(0)
    x1 = a.generate_xs_randomly(64)
    b = _env(100, 300, 1, seed=99)
    torch.manual_seed(12345)
    x2 = b.generate_xs_randomly(64)
    assert torch.equal(x1, x2)
    assert not torch.equal(a.generate_xs_randomly(64), x1)            # the stream advances
    c, d = _env(100, 300, 1), _env(100, 300, 1)
    torch.manual_seed(5)
    y1 = c.generate_xs_randomly(64)
    torch.manual_seed(5)
    assert torch.equal(d.generate_xs_randomly(64), y1)


def test_local_search_class_halves_equal_whole():
    """LocalSearch.reset_search / reset / random_search on a sharded simulator."""
    from rlsolver_amd.methods.LocalSearch import LocalSearch
    n, m, S = 300, 1400, 128

    def run(off, count, hook):
        sim = _env(n, m, 9, offset=off)
        sim.stat_hook = hook
        ls = LocalSearch(sim, n)
        torch.manual_seed(4)
        xs = ls.reset_search(count, num_repeats=S)          # S repeats: the batch's global size
        v0 = ls.reset(xs.clone()).clone()
        for _ in range(2):
            gx, gv, _ = ls.random_search(num_iters=4, num_spin=5)
        return xs, v0, gx.clone(), gv.clone()

    tape = StatTape()
    whole = run(0, S, tape.recorder())
    for off in (0, S // 2):
        part = run(off, S // 2, tape.replayer())
        for a, b in zip(part, whole):
            assert torch.equal(a, b[off:off + S // 2])
    assert (whole[3] >= whole[1]).all() and (whole[3] > whole[1]).any()


def test_gym_env_reset_and_steps_halves_equal_whole():
    from rlsolver_amd import ops
    from rlsolver_amd.envs.env_PPO import EnvMaxcut as Gym
    n, B, steps = 120, 256, 5
    g = generate_gnm(n, 500, 8)

    def run(off, count):
        env = Gym(types.SimpleNamespace(num_nodes=n, num_envs=count, num_steps=steps), mygraph=g, device=DEV, env_offset=off)
        torch.manual_seed(21)
        x0 = env.reset().clone()
        rs = []
        for t in range(steps):
            act = ops.rand_actions(count, n, seed=77, step=t, device=DEV, env_offset=off)
            _, r, done, cur = env.step(act)
            rs.append(torch.stack([r, done, cur]).clone())
        return x0, env.xs.clone(), torch.stack(rs)

    w = run(0, B)
    for off in (0, B // 2):
        p = run(off, B // 2)
        assert torch.equal(p[0], w[0][off:off + B // 2]) and torch.equal(p[1], w[1][off:off + B // 2])
        assert torch.equal(p[2], w[2][:, :, off:off + B // 2])


def test_isco_steps_halves_equal_whole():
    """ISCO_maxcut / ISCO_TSP: initial samples and three sampler steps with production draws."""
    from rlsolver_amd.envs.env_ISCO import ISCO_TSP
    from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
    from rlsolver_amd.graph import generate_tsp_coords, tsp_tables
    n, B = 96, 128
    g = gnm_arr(n, 400, 6)
    pd = {"edge_from": torch.from_numpy(g[:, 0].copy()).to(DEV), "edge_to": torch.from_numpy(g[:, 1].copy()).to(DEV),
          "num_nodes": n, "num_edges": g.shape[0]}

    def run_mc(off, count):
        s = ISCO_maxcut(pd, batch_size=count, device=DEV, env_offset=off)
        torch.manual_seed(3)
        x = s.random_gen_init_sample()
        outs = [x.clone()]
        for t in range(3):
            x, e, acc = s.step(x, 5, 0.7)
            outs += [x.clone(), e.clone(), acc.clone()]
        return outs

    w = run_mc(0, B)
    assert 0.2 < float(w[0][:, 0].mean()) < 0.8                  # node 0 is a coin here (no gauge fixing)
    for off in (0, B // 2):
        for a, b in zip(run_mc(off, B // 2), w):
            assert torch.equal(a, b[off:off + B // 2])

    dist, near, rand = tsp_tables(generate_tsp_coords(40, seed=1), K=8)
    params = {"distance": torch.from_numpy(dist).to(DEV), "nearest_indices": torch.from_numpy(near).to(DEV),
              "random_indices": torch.from_numpy(rand).to(DEV), "num_nodes": 40}

    def run_tsp(off, count):
        s = ISCO_TSP(params, batch_size=count, K=8, device=DEV, env_offset=off)
        torch.manual_seed(5)
        x = s.random_gen_init_sample()
        outs = [x.clone()]
        for t in range(3):
            x, acc, log_acc, cur = s.step(x, 4, 0.5, want_terms=True)
            outs += [x.clone(), log_acc.clone(), cur.clone()]
        return outs

    w = run_tsp(0, B)
    for off in (0, B // 2):
        for a, b in zip(run_tsp(off, B // 2), w):
            assert torch.equal(a, b[off:off + B // 2])


@pytest.mark.parametrize("kind", ["BA", "ER"])
def test_dense_spinsystem_reset_halves_equal_whole(kind):
    """The PECO training env: per-env couplings drawn by the generator kernel + a reset, both keyed by the global env id."""
    from rlsolver_amd.envs import spinsystem as ss
    from rlsolver_amd.envs.util_envs_PECO import EdgeType, RandomBAGraphGenerator, RandomERGraphGenerator
    n, B = 40, 64

    def run(off, count, hook):
        gen = (RandomBAGraphGenerator(n_spins=n, m_insertion_edges=4, edge_type=EdgeType.DISCRETE, num_envs=count, device=DEV,
                                      env_offset=off) if kind == "BA" else
               RandomERGraphGenerator(n_spins=n, p_connection=0.15, edge_type=EdgeType.DISCRETE, num_envs=count, device=DEV,
                                      env_offset=off))
        env = ss.SpinSystem(None, None, count, max_steps=2 * n, graph_generator=gen, device=DEV, env_offset=off)
        env.stat_hook = hook
        torch.manual_seed(8)
        obs = env.reset().clone()
        act = (torch.arange(count, device=DEV) + off) % n          # the action of GLOBAL env e is e % n
        obs2, rew, done = env.step(act)
        return obs, env._matrix.clone(), obs2.clone(), rew.clone()

    tape = StatTape()
    w = run(0, B, tape.recorder())
    for off in (0, B // 2):
        p = run(off, B // 2, tape.replayer())
        assert torch.equal(p[1], w[1][off:off + B // 2]), "couplings are not keyed by the global env id"
        assert torch.equal(p[0], w[0][off:off + B // 2])
        assert torch.equal(p[2], w[2][off:off + B // 2]) and torch.equal(p[3], w[3][off:off + B // 2])
    assert w[1].abs().sum() > 0


def test_shared_graph_spinsystem_reset_halves_equal_whole():
    from rlsolver_amd.envs import spinsystem as ss
    n, B = 60, 128
    g = generate_gnm(n, 200, 2)

    def run(off, count):
        env = ss.SpinSystem(g, n, count, max_steps=2 * n, device=DEV, env_offset=off)
        torch.manual_seed(8)
        return env.reset().clone()

    w = run(0, B)
    for off in (0, B // 2):
        assert torch.equal(run(off, B // 2), w[off:off + B // 2])


def _mcpg_setup(n, m, seed):
    from rlsolver_amd.methods import MCPG as amcpg
    g = gnm_arr(n, m, seed)
    return amcpg, amcpg.make_data(n, g[:, 0].copy(), g[:, 1].copy(), DEV)


def test_mcpg_round_shards_equal_whole():
    """MCPGRound over three rounds: (a) the whole batch through the fused merge kernel; (b) the whole batch in the sharded
    code path (per-chain merge kernel + exchange + column writes) -- must equal (a) bit for bit; (c) two shards of the kept
    chains replaying (b)'s exchange -- their chains, expected cuts, incumbents and start states are (a)'s."""
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    n, m, M, R, num_ls, rounds = 400, 1800, 256, 4, 2, 3
    amcpg, data = _mcpg_setup(n, m, 12)
    torch.manual_seed(0)
    xs_init = (torch.rand(n, M, device=DEV) < 0.5).float()
    vs_init = torch.zeros(M, device=DEV)
    probs = torch.rand(n, device=DEV) * 0.6 + 0.2

    def run(m0, Ml, hook, force_sharded=False):
        rnd = amcpg.MCPGRound(data, xs_init[:, m0:m0 + Ml].contiguous(), vs_init[m0:m0 + Ml], Ml, R, num_ls,
                              kept_offset=m0, total_kept=M)
        rnd.stat_hook = hook
        if force_sharded:
            rnd.sharded = True
        torch.manual_seed(6)
        log = []
        for r in range(rounds):
            value, best = rnd.step(probs)
            loss = rnd.get_return(probs.clone().requires_grad_(True))
            log.append(dict(samples=rnd.samples.words.clone(), expected=rnd.expected.clone(), value=value.clone(),
                            best=best.clone(), res=rnd.now_max_res.clone(), info=rnd.now_max_info.words.clone(),
                            start=rnd.start.words.clone(), loss=loss.detach().clone()))
        return rnd, log

    _, a = run(0, M, None)
    tape = StatTape()
    rb, b = run(0, M, tape.recorder(), force_sharded=True)
    kinds = [k for k, _ in tape.items]
    assert kinds.count("best") == 2 * rounds and "sum" in kinds
    for ra, rbb in zip(a, b):
        for k in ("samples", "expected", "res", "info", "start", "best"):
            assert torch.equal(ra[k], rbb[k]), k
        assert torch.allclose(ra["value"], rbb["value"], rtol=0, atol=1e-3)       # f32 mean vs f64 sum / C
        assert torch.allclose(ra["loss"], rbb["loss"], rtol=1e-4, atol=1e-4)
    v, x = rb.best_solution()
    assert v == float(a[-1]["best"]) and x.dtype == torch.bool and x.shape == (n,)
    h = M // 2
    tl = h // 64                                    # tiles per repeat of a shard; the whole batch has 2 tl per repeat
    for m0 in (0, h):
        _, c = run(m0, h, tape.replayer())
        for ra, rc in zip(a, c):
            # local tile (r, t) <-> global tile (r, m0 / 64 + t); local chain r * h + j <-> global r * M + m0 + j
            gt = torch.tensor([r * (M // 64) + m0 // 64 + t for r in range(R) for t in range(tl)], device=DEV)
            gc = torch.tensor([r * M + m0 + j for r in range(R) for j in range(h)], device=DEV)
            assert torch.equal(rc["samples"], ra["samples"][gt]), "a shard's chains are not the whole batch's"
            assert torch.equal(rc["expected"], ra["expected"][gc])
            assert torch.equal(rc["res"], ra["res"][m0:m0 + h])
            assert torch.equal(rc["info"], ra["info"][m0 // 64:m0 // 64 + tl])
            assert torch.equal(rc["start"], ra["start"][m0 // 64:m0 // 64 + tl])
            assert torch.equal(rc["best"], ra["best"])
            assert torch.allclose(rc["loss"], ra["loss"], rtol=1e-4, atol=1e-4)


def test_a_shard_without_a_group_refuses_to_step():
    n, M = 128, 128
    amcpg, data = _mcpg_setup(n, 400, 3)
    xs = (torch.rand(n, 64, device=DEV) < 0.5).float()
    rnd = amcpg.MCPGRound(data, xs, torch.zeros(64, device=DEV), 64, 2, 1, kept_offset=64, total_kept=M)
    with pytest.raises(RuntimeError, match="needs group="):
        rnd.step(torch.full((n,), 0.5, device=DEV))
    with pytest.raises(ValueError):
        amcpg.MCPGRound(data, xs, torch.zeros(64, device=DEV), 64, 2, 1, kept_offset=32, total_kept=M)     # not a whole tile
    with pytest.raises(ValueError):
        amcpg.MCPGRound(data, xs, torch.zeros(64, device=DEV), 64, 2, 1, kept_offset=128, total_kept=M)    # outside the batch


def test_chain_ids_validation_and_functional_api():
    """metro_sampling_packed / sampler_func_packed with chain_ids: a plain offset shard equals the tail of the whole batch;
    the level kernel refuses ids that are not whole tiles."""
    from rlsolver_amd import _abi
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    n, m, C = 300, 1200, 512
    amcpg, data = _mcpg_setup(n, m, 4)
    torch.manual_seed(1)
    start = PackedChains.pack((torch.rand(n, C, device=DEV) < 0.5).float())
    probs = torch.rand(n, device=DEV) * 0.5 + 0.25
    whole = amcpg.metro_sampling_packed(probs, start, n // 10, seed=123)
    # the stop rule needs whole-batch counts: take them from a recorded run
    tape = StatTape()
    rec = types.SimpleNamespace(_global_sum=lambda t: tape.recorder()("sum", t))
    whole2 = amcpg.metro_sampling_packed(probs, start, n // 10, seed=123, stats=rec, total_chains=C)
    assert torch.equal(whole.words, whole2.words)
    for c0 in (0, C // 2):
        rp = tape.replayer()
        part = PackedChains(start.words[c0 // 64:(c0 + C // 2) // 64].contiguous(), C // 2)
        out = amcpg.metro_sampling_packed(probs, part, n // 10, seed=123, chain_ids=(c0, 0, 0),
                                          stats=types.SimpleNamespace(_global_sum=lambda t: rp("sum", t)), total_chains=C)
        assert torch.equal(out.words, whole.words[c0 // 64:(c0 + C // 2) // 64])
    vg, xg, val, loc = amcpg.sampler_func_packed(data, whole, 2, C, 1, in_place=False, seed=9)
    _, _, _, loc2 = amcpg.sampler_func_packed(data, PackedChains(whole.words[C // 128:].contiguous(), C // 2), 2, C // 2, 1,
                                              in_place=False, seed=9, chain_ids=(C // 2, 0, 0))
    assert torch.equal(loc2.words, loc.words[C // 128:])
    with pytest.raises(RuntimeError, match="multiples of 64"):
        amcpg.sampler_func_packed(data, whole, 2, C, 1, in_place=False, seed=9, chain_ids=(32, 0, 0))
    with pytest.raises(RuntimeError):
        amcpg.metro_sampling_packed(probs, start, n // 10, seed=1, chain_ids=(-1, 0, 0))
    assert _abi.version() >= 9


def test_launcher_keeps_its_graph_alive():
    """the step launcher's argument tuple holds the graph handle as a plain int; the closure must own the
    DeviceGraph or a caller that drops it launches on freed host / device memory."""
    import gc
    from rlsolver_amd import ops
    from rlsolver_amd.graph import build_csr
    n, B = 500, 512

    def make():
        g = ops.DeviceGraph(build_csr(generate_gnm(n, 2500, 1), num_nodes=n, if_bidirectional=False), DEV)
        xs = ops.rand_spins(B, n, seed=1, device=DEV)
        out = torch.empty_like(xs)
        obj = ops.maxcut_obj(g, xs).to(torch.int32)
        act = ops.rand_actions(B, n, seed=2, step=0, device=DEV)
        rew = torch.empty(B, dtype=torch.float32, device=DEV)
        want = ops.maxcut_obj(g, xs)
        ref_g = ops.DeviceGraph(build_csr(generate_gnm(n, 2500, 1), num_nodes=n, if_bidirectional=False), DEV)
        return ops.maxcut_step_launcher(g, xs, out, act, obj, rew), out, obj, ref_g

    launch, out, obj, ref_g = make()
    gc.collect()
    junk = [torch.full((1 << 20,), -1, dtype=torch.int32, device=DEV) for _ in range(8)]      # reuse whatever was freed
    torch.cuda.synchronize()
    launch()
    torch.cuda.synchronize()
    assert torch.equal(ops.maxcut_obj(ref_g, out).to(torch.int32), obj)
    del junk


def test_an_empty_shard_goes_through_every_call():
    """env_shard gives some ranks nothing when B < world: such a rank still makes every call (and so joins every exchange)."""
    from rlsolver_amd.methods.LocalSearch import LocalSearch
    env = _env(200, 900, 3, offset=77)
    seen = []
    env.stat_hook = lambda kind, t: (seen.append((kind, t.clone())), t)[1]
    torch.manual_seed(1)
    xs = env.generate_xs_randomly(0)
    assert xs.shape == (0, 200)
    xs, vs = env.local_search_inplace(xs, torch.empty(()), num_iters=3, num_spin=5)
    assert xs.shape == (0, 200) and vs.shape == (0,)
    assert [k for k, _ in seen] == ["minmax"]
    mm = seen[0][1]
    assert bool((mm[0] > mm[1]).all())                      # the neutral element of (min, max): loses against every real row
    ls = LocalSearch(env, 200)
    ls.reset(xs)
    gx, gv, n = ls.random_search(num_iters=2, num_spin=4)
    assert gx.shape == (0, 200) and int(n) == 0
