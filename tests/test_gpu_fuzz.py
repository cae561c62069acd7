"""Randomised differential test: many small random shapes (node counts around the 64-lane and 16-byte boundaries,
empty and dense graphs, isolated nodes, multi-edges, ragged batches) through every MaxCut kernel against the numpy
oracle.  Integer results must be bit-exact."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from rlsolver_amd import ops
from tests.gpu_util import DEV, device_graph, to_dev_bool

pytestmark = pytest.mark.gpu

_SIZES = [1, 2, 3, 5, 15, 16, 17, 31, 32, 48, 63, 64, 65, 100, 127, 128, 129, 200, 256, 300]


def _random_graph(rng, n):
    kind = rng.randint(4)
    if n == 1 or kind == 0:
        m = 0 if n == 1 else rng.randint(0, 3)
    elif kind == 1:
        m = rng.randint(1, 3 * n)
    elif kind == 2:
        m = rng.randint(n, 8 * n)
    else:
        m = min(n * (n - 1) // 2, rng.randint(1, 40 * n))           # dense-ish: degrees up to ~80
    e = rng.randint(0, n, size=(m, 2)) if n > 1 else np.zeros((0, 2), dtype=np.int64)
    e = e[e[:, 0] != e[:, 1]]                                        # multi-edges stay, self loops go
    if len(e) == 0 and n > 1:
        e = np.array([[0, n - 1]])
    if n == 1:
        return np.zeros((0, 3), dtype=np.int64)
    return np.concatenate([np.sort(e, axis=1), np.ones((len(e), 1), dtype=np.int64)], axis=1).astype(np.int64)


@pytest.mark.parametrize("seed", range(40))
def test_random_shapes_against_oracle(seed):
    rng = np.random.RandomState(1000 + seed)
    n = int(_SIZES[rng.randint(len(_SIZES))])
    graph = _random_graph(rng, n)
    if len(graph) == 0:
        pytest.skip("edgeless graph: DeviceGraph needs at least one edge")
    bidir = int(rng.randint(2))
    B = int(rng.choice([1, 2, 63, 64, 65, 130, 200]))
    g = device_graph(graph, n, bidir)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    x = to_dev_bool(xs)
    # K1, K1', K2, K3
    want = onp.maxcut_obj(xs, graph, bool(bidir))
    v = ops.maxcut_obj(g, x)
    assert np.array_equal(v.cpu().numpy(), want)
    assert np.array_equal(ops.maxcut_obj(g, x.float()).cpu().numpy(), want)
    assert np.array_equal(ops.maxcut_node_cutdeg(g, x).cpu().numpy(), onp.maxcut_node_cutdeg(xs, graph, n, bool(bidir)))
    assert np.array_equal(ops.maxcut_delta_all(g, x).cpu().numpy(), onp.maxcut_delta_all(xs, graph, n, None))
    # K4: emit and in place, twelve steps incl. repeated and boundary nodes
    env = onp.PPOEnvOracle(graph, n, 10 ** 9, bool(bidir))
    env.reset_to(xs)
    xa, xb = x.clone(), torch.empty_like(x)
    xi = x.clone()
    obj = v.to(torch.int32)
    obj_i = obj.clone()
    rew = torch.empty(B, dtype=torch.float32, device=DEV)
    rew_i = torch.empty_like(rew)
    for t in range(6):
        a = rng.randint(0, n, size=B)
        if t == 2:
            a[:] = n - 1
        if t == 3:
            a[:] = 0
        _, r, _, c = env.step(a)
        ad = torch.from_numpy(a).to(DEV)
        ops.maxcut_step(g, xa, xb, ad, obj, rew)
        xa, xb = xb, xa
        ops.maxcut_step(g, xi, xi, ad, obj_i, rew_i)
        assert np.array_equal(rew.cpu().numpy(), r) and np.array_equal(rew_i.cpu().numpy(), r)
        assert np.array_equal(obj.cpu().numpy().astype(np.float32), c) and torch.equal(obj, obj_i)
    assert torch.equal(xa, xi) and np.array_equal(xa.cpu().numpy().astype(np.float32), env.xs)
    # K6 and K5 (the oracle's sweep re-evaluates the objective per node: keep it to small cases)
    mask = rng.rand(B, n) < 0.1
    xp, vp = x.clone(), v.clone()
    ops.maxcut_propose_accept(g, xp, to_dev_bool(mask), vp)
    v1 = onp.maxcut_obj((xs.astype(bool) ^ mask).astype(np.uint8), graph, bool(bidir))
    keep = v1 >= want
    assert np.array_equal(vp.cpu().numpy(), np.where(keep, v1, want))
    assert np.array_equal(xp.cpu().numpy(), np.where(keep[:, None], xs.astype(bool) ^ mask, xs.astype(bool)))
    if n <= 130 and len(graph) <= 1500:
        xs_s, vs_s = x.clone(), v.clone()
        ops.maxcut_greedy_sweep(g, xs_s, vs_s)
        x_ref, v_ref = onp.greedy_sweep(xs.astype(bool), want.copy(), graph, bool(bidir))
        assert np.array_equal(xs_s.cpu().numpy(), x_ref) and np.array_equal(vs_s.cpu().numpy(), v_ref)


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_large_batch_paths(seed):
    """B >= 2048 selects the tile forms of K2/K3 (bit-sliced lane = node) and the multi-tile paths of the others."""
    rng = np.random.RandomState(5000 + seed)
    n = int(_SIZES[3 + rng.randint(len(_SIZES) - 3)])
    graph = _random_graph(rng, n)
    if len(graph) == 0:
        pytest.skip("edgeless graph")
    bidir = int(rng.randint(2))
    B = int(rng.choice([2048, 2111, 4099]))
    g = device_graph(graph, n, bidir)
    xs = rng.randint(0, 2, size=(B, n)).astype(np.uint8)
    x = to_dev_bool(xs)
    want = onp.maxcut_obj(xs, graph, bool(bidir))
    v = ops.maxcut_obj(g, x)
    assert np.array_equal(v.cpu().numpy(), want)
    assert np.array_equal(ops.maxcut_node_cutdeg(g, x).cpu().numpy(), onp.maxcut_node_cutdeg(xs, graph, n, bool(bidir)))
    assert np.array_equal(ops.maxcut_delta_all(g, x).cpu().numpy(), onp.maxcut_delta_all(xs, graph, n, None))
    ws = ops.maxcut_ls_weights(g, x, 2 if bidir else 1)[0]
    deg = np.bincount(g.csr.eu, minlength=n)
    assert np.array_equal(ws.cpu().numpy(), deg[None, :] - (2 if bidir else 1) * onp.maxcut_node_cutdeg(xs, graph, n, bool(bidir)))
    xs_s, vs_s = x.clone(), v.clone()
    ops.maxcut_greedy_sweep(g, xs_s, vs_s)
    assert torch.equal(ops.maxcut_obj(g, xs_s), vs_s) and bool((vs_s >= v).all())
    if n <= 65 and len(graph) <= 400:
        sub = np.arange(0, B, 97)
        x_ref, v_ref = onp.greedy_sweep(xs[sub].astype(bool), want[sub].copy(), graph, bool(bidir))
        assert np.array_equal(xs_s.cpu().numpy()[sub], x_ref) and np.array_equal(vs_s.cpu().numpy()[sub], v_ref)


@pytest.mark.parametrize("seed", range(16))
def test_mcpg_level_kernel_random_shapes(seed):
    """Level-parallel K7 (lane = node groups, hub groups, tie coins) and the lane = chain stream kernel against the
    oracle's sequential pass, on random graphs / chain counts / pass counts."""
    from rlsolver_amd import ops_mcpg_tsp as mops
    from rlsolver_amd.methods import MCPG as amcpg
    rng = np.random.RandomState(9000 + seed)
    n = int(rng.choice([2, 3, 17, 64, 65, 100, 129, 300]))
    graph = _random_graph(rng, n)
    if rng.randint(3) == 0 and n >= 100:                             # add a hub of degree > 64
        hub = rng.randint(n)
        extra = np.array([(min(hub, j), max(hub, j), 1) for j in rng.choice(n, size=min(n - 1, 90), replace=False) if j != hub])
        graph = np.concatenate([graph, extra])
    if len(graph) == 0:
        pytest.skip("edgeless graph")
    ei = graph[:, :2].T.copy()
    C = int(rng.choice([1, 63, 64, 65, 130]))
    num_ls = int(rng.randint(1, 4))
    deg = np.bincount(ei.reshape(-1), minlength=n)
    order = np.argsort(-deg, kind="stable")
    data = amcpg.make_data(n, ei[0], ei[1], DEV, sorted_degree_nodes=order)
    xs0 = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    coin = rng.randint(0, 2, size=(num_ls, n, C)).astype(bool)
    uni = np.where(coin, 0.25, 0.75).astype(np.float32)
    _, _, _, x_all, exp_w = onp.sampler_func(ei, n, order, xs0, num_ls, C, 1, uni)
    CB = (C + 63) // 64
    bits = np.zeros((num_ls * n, CB * 64), dtype=np.uint64)
    bits[:, :C] = coin.reshape(num_ls * n, C)
    words = (bits.reshape(num_ls * n, CB, 64) << np.arange(64, dtype=np.uint64)).sum(axis=2, dtype=np.uint64)
    x0 = torch.from_numpy(xs0).to(DEV)
    xs_g, exp_g = mops.mcpg_local_search_levels(data.graph, x0, data._lv_ptr, data._lv_data, num_ls, 0,
                                                coins=torch.from_numpy(words.view(np.int64)).to(DEV))
    assert np.array_equal(xs_g.cpu().numpy(), x_all) and np.array_equal(exp_g.cpu().numpy(), exp_w)
    xs_s, exp_s = mops.mcpg_local_search(data.graph, x0, data._order_i32, num_ls, torch.from_numpy(uni).to(DEV), 0,
                                         visit_stream=data._visit_stream)
    assert np.array_equal(xs_s.cpu().numpy(), x_all) and np.array_equal(exp_s.cpu().numpy(), exp_w)


@pytest.mark.parametrize("seed", range(10))
def test_fused_local_search_random_shapes(seed):
    """EnvMaxcut.local_search_inplace, fused kernel with supplied noise, against the oracle on random shapes."""
    from rlsolver_amd.envs.env_L2A import EnvMaxcut
    rng = np.random.RandomState(7000 + seed)
    n = int(rng.choice([20, 33, 64, 100, 128, 130, 257]))
    graph = _random_graph(rng, n)
    graph = graph[np.unique(graph[:, :2], axis=0, return_index=True)[1]]      # simple graph: the env takes mygraph tuples
    if len(graph) < 2:
        pytest.skip("too few edges")
    bidir = bool(rng.randint(2))
    B = int(rng.choice([1, 7, 64, 65, 130]))
    num_iters, num_spin = int(rng.randint(0, 5)), int(rng.randint(1, min(9, n - 1)))
    env = EnvMaxcut(mygraph=[tuple(int(t) for t in r) for r in graph], device=DEV, if_bidirectional=bidir, num_nodes=n)
    xs = rng.randint(0, 2, size=(B, n)).astype(bool)
    noise = rng.randn(num_iters + 1, B, n).astype(np.float32)
    x = torch.from_numpy(xs).to(DEV)
    v = env.calculate_obj_values(x)
    gx, gv = env.local_search_inplace(x, v.clone(), num_iters=num_iters, num_spin=num_spin, noise_std=0.3,
                                      noise=torch.from_numpy(noise).to(DEV))
    wx, wv = onp.local_search_inplace(xs.copy(), graph, n, bidir, noise, num_iters=num_iters, num_spin=num_spin,
                                      noise_std=0.3)
    assert np.array_equal(gv.cpu().numpy(), wv) and np.array_equal(gx.cpu().numpy(), wx)
