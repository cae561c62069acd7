"""Parity of the MCPG and TSP kernels / Python surfaces against golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as onp
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.methods import MCPG as amcpg
from tests.gpu_util import DEV, gnm_arr

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_mcpg_golden(golden, gname):
    z = golden("mcpg")
    graph = z[f"{gname}/graph"]
    n = int(graph[:, :2].max()) + 1
    ei = z[f"{gname}/edge_index"]
    data = amcpg.make_data(n, ei[0], ei[1], DEV, sorted_degree_nodes=z[f"{gname}/sorted_degree_nodes"])
    assert data.weighted_degree == z[f"{gname}/weighted_degree"].tolist()
    T = int(z[f"{gname}/metro/T"])
    out = amcpg.metro_sampling(dev(z[f"{gname}/metro/probs"]), dev(z[f"{gname}/metro/start"], torch.float32), T,
                               device=DEV, index=dev(z[f"{gname}/metro/index"]), u=dev(z[f"{gname}/metro/u"]))
    assert out.dtype == torch.float32
    assert np.array_equal(out.cpu().numpy().astype(np.uint8), z[f"{gname}/metro/out"])
    vs_good, xs_good, value = amcpg.sampler_func(
        data, dev(z[f"{gname}/sampler/xs_in"], torch.float32), int(z[f"{gname}/sampler/num_ls"]),
        int(z[f"{gname}/sampler/total_mcmc_num"]), int(z[f"{gname}/sampler/repeat_times"]), DEV,
        uniforms=dev(z[f"{gname}/sampler/uniforms"]))
    assert np.array_equal(vs_good.cpu().numpy(), z[f"{gname}/sampler/vs_good"])
    assert np.array_equal(xs_good.cpu().numpy(), z[f"{gname}/sampler/xs_good"])
    np.testing.assert_allclose(value.cpu().numpy(), z[f"{gname}/sampler/value"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("n,m,M,R,num_ls", [(300, 1500, 20, 5, 2), (2000, 19990, 64, 3, 1), (64, 200, 1, 1, 3)])
def test_mcpg_random_vs_oracle(n, m, M, R, num_ls):
    graph = gnm_arr(n, m, seed=5)
    ei = graph[:, :2].T.copy()
    rng = np.random.RandomState(3)
    C = M * R
    deg = np.bincount(ei.reshape(-1), minlength=n)
    order = np.argsort(-deg, kind="stable")
    data = amcpg.make_data(n, ei[0], ei[1], DEV, sorted_degree_nodes=order)
    probs = (rng.rand(n) * 0.6 + 0.2).astype(np.float32)
    start = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    T = max(1, n // 10)
    index = rng.randint(0, n, size=(5 * T, C)).astype(np.int64)
    u = rng.rand(5 * T, C).astype(np.float32)
    want, t_used = onp.metro_sampling(probs, start, T, index, u)
    got = amcpg.metro_sampling(dev(probs), dev(start), T, device=DEV, index=dev(index), u=dev(u))
    assert np.array_equal(got.cpu().numpy(), want)
    # forcing the stop rule to bite: tiny T budget relative to acceptance
    uni = rng.rand(num_ls, n, C).astype(np.float32)
    vs_w, xs_w, val_w, x_all, exp_w = onp.sampler_func(ei, n, order, want, num_ls, M, R, uni)
    vs_g, xs_g, val_g = amcpg.sampler_func(data, got, num_ls, M, R, DEV, uniforms=dev(uni))
    assert np.array_equal(vs_g.cpu().numpy(), vs_w)
    assert np.array_equal(xs_g.cpu().numpy(), xs_w)
    np.testing.assert_allclose(val_g.cpu().numpy(), val_w, atol=1e-3)
    # the streaming kernel (visit stream through an LDS ring) and the generic one agree exactly
    xa, ea = mops.mcpg_local_search(data.graph, got, data._order_i32, num_ls, dev(uni), 0, visit_stream=data._visit_stream)
    xb, eb = mops.mcpg_local_search(data.graph, got, data._order_i32, num_ls, dev(uni), 0, visit_stream=None)
    assert torch.equal(xa, xb) and torch.equal(ea, eb)
    assert np.array_equal(xa.cpu().numpy(), x_all) and np.array_equal(ea.cpu().numpy(), exp_w)


@pytest.mark.parametrize("kind,n,m,C,num_ls", [("gnm", 300, 1500, 100, 2), ("gnm", 2000, 19990, 130, 1), ("ba", 600, 5, 64, 3),
                                                ("star", 400, 0, 70, 2), ("gnm", 64, 200, 1, 3),
                                                ("gnm", 200, 9000, 65, 2),      # degrees ~90: every row on 2-8 lanes
                                                ("gnm", 150, 11175, 64, 2),     # complete graph, degree 149: hubs only
                                                ("gnm", 700, 30000, 3, 2)])     # degrees 60-110 around the 64-round limit
def test_mcpg_level_parallel_kernel_vs_oracle(kind, n, m, C, num_ls):
    """The production K7 kernel (lane = node on the level schedule, tie coins instead of uniforms) against the
    oracle's sequential pass fed with uniforms of 0.25 / 0.75: away from 1/2 the reference's float rule
    (s + u/4) < (deg + 1/4)/2 is exactly "2s < deg, or 2s == deg and u < 1/2"."""
    from rlsolver_amd import graph as G
    if kind == "gnm":
        graph = gnm_arr(n, m, seed=7)
    elif kind == "ba":
        graph = np.asarray(G.generate_ba(n, m, seed=3), dtype=np.int64)        # hubs above 64: lane = neighbour groups
    else:
        graph = np.array([(0, j, 1) for j in range(1, 301)] + [(j, j + 1, 1) for j in range(1, n - 1)], dtype=np.int64)
    ei = graph[:, :2].T.copy()
    rng = np.random.RandomState(11)
    deg = np.bincount(ei.reshape(-1), minlength=n)
    order = np.argsort(-deg, kind="stable")
    data = amcpg.make_data(n, ei[0], ei[1], DEV, sorted_degree_nodes=order)
    assert data._lv_ptr is not None
    xs0 = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    coin = rng.randint(0, 2, size=(num_ls, n, C)).astype(bool)                 # indexed [pass, visiting position, chain]
    uni = np.where(coin, 0.25, 0.75).astype(np.float32)
    _, _, _, x_all, exp_w = onp.sampler_func(ei, n, order, xs0, num_ls, C, 1, uni)
    CB = (C + 63) // 64
    bits = np.zeros((num_ls * n, CB * 64), dtype=np.uint64)
    bits[:, :C] = coin.reshape(num_ls * n, C)
    words = (bits.reshape(num_ls * n, CB, 64) << np.arange(64, dtype=np.uint64)).sum(axis=2, dtype=np.uint64)
    xs_g, exp_g = mops.mcpg_local_search_levels(data.graph, dev(xs0), data._lv_ptr, data._lv_data, num_ls, 0,
                                                coins=torch.from_numpy(words.view(np.int64)).to(DEV))
    assert np.array_equal(xs_g.cpu().numpy(), x_all)
    assert np.array_equal(exp_g.cpu().numpy(), exp_w)


@pytest.mark.parametrize("kind,n,m,C,num_ls", [("ba", 600, 5, 200, 2), ("star", 400, 0, 70, 2), ("gnm", 300, 4000, 130, 2)])
def test_mcpg_level_parallel_kernel_is_exact_within_ulps_of_one_half(kind, n, m, C, num_ls):
    """Recorded draws packed within a few float32 ulps of 1/2, where (s + u/4) < (deg + 1/4)/2 stops being "u < 1/2"
    (the sum rounds up to the threshold; the band widens with the degree): sampler_func turns them into tie coins by
    the reference's own float32 expression (tie_coins_from_uniforms), and the production kernel must then reproduce the
    sequential float pass of MCPG.py:136-142 bit for bit."""
    from rlsolver_amd import graph as G
    if kind == "gnm":
        graph = gnm_arr(n, m, seed=5)
    elif kind == "ba":
        graph = np.asarray(G.generate_ba(n, m, seed=3), dtype=np.int64)
    else:
        graph = np.array([(0, j, 1) for j in range(1, 301)] + [(j, j + 1, 1) for j in range(1, n - 1)], dtype=np.int64)
    ei = graph[:, :2].T.copy()
    rng = np.random.RandomState(23)
    deg = np.bincount(ei.reshape(-1), minlength=n)
    order = np.argsort(-deg, kind="stable")
    data = amcpg.make_data(n, ei[0], ei[1], DEV, sorted_degree_nodes=order)
    assert data._lv_ptr is not None
    xs0 = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    steps = rng.randint(-80, 81, size=(num_ls, n, C))                            # ulps of 0.5- (2^-25) around one half
    uni = (np.float32(0.5) + steps.astype(np.float32) * np.float32(2.0 ** -25)).astype(np.float32)
    coin_naive = uni < np.float32(0.5)
    M = C // 2 if C % 2 == 0 else C
    R = C // M
    vs_w, xg_w, val_w, x_all, exp_w = onp.sampler_func(ei, n, order, xs0, num_ls, M, R, uni)
    vs_g, xg_g, val_g = amcpg.sampler_func(data, dev(xs0), num_ls, M, R, DEV, uniforms=dev(uni))
    assert np.array_equal(xg_g.cpu().numpy(), xg_w)
    assert np.array_equal(vs_g.cpu().numpy(), vs_w)
    # the band is real: the float rule and "u < 1/2" disagree somewhere on these draws
    degp = deg[order].astype(np.float32)[None, :, None]
    coin_exact = (degp / np.float32(2) + uni * np.float32(0.25)) < (degp + np.float32(0.25)) / np.float32(2)
    assert (coin_exact != coin_naive).any()


def test_mcpg_production_rng_is_distributionally_sane():
    n, m, C = 500, 3000, 4096
    graph = gnm_arr(n, m, seed=9)
    ei = graph[:, :2].T.copy()
    data = amcpg.make_data(n, ei[0], ei[1], DEV)
    torch.manual_seed(0)
    probs = torch.full((n,), 0.5, device=DEV)
    start = torch.zeros((n, C), device=DEV)
    out = amcpg.metro_sampling(probs, start, n // 10, device=DEV)
    # p = 0.5 -> accept rate 1 -> every proposal flips; C*T accepts reached after exactly T rounds
    flips = out.sum(0)
    assert float(flips.max()) <= n // 10 and float(flips.mean()) > 0.8 * (n // 10) * 0.9
    vs_good, xs_good, value = amcpg.sampler_func(data, out, 4, 512, 8, DEV)
    # local search must beat random assignment (m/2) clearly, and values are consistent with xs_good
    assert float(vs_good.mean()) > 0.55 * m
    cut = onp.maxcut_obj(xs_good.cpu().numpy().T > 0, graph, False)
    assert np.array_equal(cut.astype(np.float32), vs_good.cpu().numpy())
    assert abs(float(value.mean())) < 1e-2


@pytest.mark.parametrize("name", ["a5", "berlin52"])
def test_tsp_golden(golden, name):
    z = golden("tsp")
    d = dev(z[f"{name}/distance"])
    perms = dev(z[f"{name}/perms"])
    K = int(z[f"{name}/K"])
    length = mops.tsp_tour_length(d, perms).cpu().numpy()
    np.testing.assert_allclose(length, z[f"{name}/length_f32"], rtol=1e-5)            # north_star tolerance
    np.testing.assert_allclose(length, z[f"{name}/length_f64_distance_calc"], rtol=1e-5)
    sel = onp.tsp_selected_partner(z[f"{name}/perms"], z[f"{name}/nearest_indices"], z[f"{name}/random_indices"],
                                   z[f"{name}/opt2/rand"], z[f"{name}/opt2/randint_nearest"],
                                   z[f"{name}/opt2/randint_random"], K)
    T = float(z[f"{name}/opt2/temperature"])
    lr, idx, ban = mops.tsp_swap_delta_all(d, perms, dev(sel), T)
    assert np.array_equal(idx.cpu().numpy(), z[f"{name}/opt2/indices"])
    assert np.array_equal(ban.cpu().numpy().astype(np.uint8), z[f"{name}/opt2/ban"])
    scale = np.abs(z[f"{name}/length_f32"]).max() / T
    np.testing.assert_allclose(lr.cpu().numpy(), z[f"{name}/opt2/logratio"], rtol=1e-5, atol=1e-5 * scale)
    x = perms.clone()
    mops.tsp_apply_swap(x, dev(z[f"{name}/switch/pos"]), idx)
    assert np.array_equal(x.cpu().numpy(), z[f"{name}/switch/out"])
    np.testing.assert_allclose(mops.tsp_tour_length(d, x).cpu().numpy(), z[f"{name}/switch/length_f32"], rtol=1e-5)
    env = z[f"{name}/twoopt/env"]
    dl = mops.tsp_2opt_delta(d, perms[dev(env)].contiguous(), dev(z[f"{name}/twoopt/i"]), dev(z[f"{name}/twoopt/j"]))
    np.testing.assert_allclose(dl.cpu().numpy(), z[f"{name}/twoopt/delta_f64"], rtol=1e-5,
                               atol=1e-5 * float(z[f"{name}/length_f32"].max()))


def _isco_draw_np(seed, env, a, b, stream):
    """The counter-based generator of the ISCO kernels (csrc/rls_draw.h: five murmur3 finalisers over seed, global env id, two
    counters and a stream id) restated in numpy -- the SPEC the in-kernel partner draw of K13 is held to."""
    M = np.uint64(0xFFFFFFFF)

    def mix(h):
        h = h ^ (h >> np.uint64(16)); h = (h * np.uint64(0x85EBCA6B)) & M
        h = h ^ (h >> np.uint64(13)); h = (h * np.uint64(0xC2B2AE35)) & M
        return h ^ (h >> np.uint64(16))
    seed, env = np.uint64(seed), np.asarray(env, dtype=np.uint64)
    a = np.asarray(a, dtype=np.uint64)
    h = mix((seed & M) ^ np.uint64(0x9E3779B9))
    h = mix(h ^ (seed >> np.uint64(32)))
    h = mix(h ^ (env & M))
    h = mix(h ^ (env >> np.uint64(32)) ^ ((a * np.uint64(0x9E3779B1)) & M))
    return mix(h ^ np.uint64((b * 0x85EBCA77) & 0xFFFFFFFF) ^ np.uint64((stream * 0xC2B2AE3D) & 0xFFFFFFFF))


@pytest.mark.parametrize("N,B,K", [(100, 1000, 20), (52, 130, 20), (7, 64, 3), (200, 70, 20), (300, 40, 20), (256, 33, 30)])
def test_tsp_opt_2_draws_its_partners_in_the_kernel(N, B, K):
    """K13 as the reference runs it (env_ISCO.py:245-262 draws inside opt_2): selected = None.  The drawn cities are exactly
    what the generator's numpy restatement predicts (streams 3 / 4 / 5 of iteration 0 -- the fused step's counters), always
    from the position's own neighbour tables; feeding them back through the recorded-draw hook reproduces every output bit
    for bit and the numpy oracle agrees; a shard of the batch under its env_offset draws what the whole batch draws."""
    from rlsolver_amd.graph import generate_tsp_coords, tsp_tables
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, seed=N), K=K)
    d, near32, rnd32 = dev(dist), dev(near.astype(np.int32)), dev(rnd.astype(np.int32))
    thr = float(np.float32(K / (K + 1)))
    perms = mops.rand_perms(B, N, seed=3, device=DEV)
    pn = perms.cpu().numpy()
    seed, off = 0x1234567ABCDEF, 7000
    lr, idx, ban, sel = mops.tsp_swap_delta_all(d, perms, None, 0.5, nearest=near32, random=rnd32, near_threshold=thr, seed=seed,
                                                env_offset=off, return_selected=True)
    env = (np.arange(B, dtype=np.uint64) + np.uint64(off))[:, None]
    pos = np.arange(N, dtype=np.uint64)[None, :]
    up = (_isco_draw_np(seed, env, pos, 0, 3) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    rn = ((_isco_draw_np(seed, env, pos, 0, 4) * np.uint64(K)) >> np.uint64(32)).astype(np.int64)
    rr = ((_isco_draw_np(seed, env, pos, 0, 5) * np.uint64(N - K - 1)) >> np.uint64(32)).astype(np.int64)
    want = np.where(up < np.float32(thr), near[pn, rn], rnd[pn, np.minimum(rr, rnd.shape[1] - 1)])
    assert np.array_equal(sel.cpu().numpy(), want)
    assert (sel.cpu().numpy() != pn).all()                                    # never the position's own city
    lr2, idx2, ban2 = mops.tsp_swap_delta_all(d, perms, sel, 0.5)
    assert torch.equal(lr, lr2) and torch.equal(idx, idx2) and torch.equal(ban, ban2)
    lr_w, idx_w, ban_w = onp.tsp_swap_delta_all(dist, pn, want, 0.5)
    assert np.array_equal(idx.cpu().numpy(), idx_w) and np.array_equal(ban.cpu().numpy(), ban_w)
    length = onp.tsp_tour_length_f64(dist, pn)
    np.testing.assert_allclose(lr.cpu().numpy(), lr_w, rtol=1e-5, atol=1e-5 * length.max() / 0.5)
    # the byte form of the tables (kept in LDS by the kernel) draws the same partners as the int32 tables read from memory
    tab8 = mops.tsp_tables8(near32, rnd32)
    assert (tab8 is None) == (N > 256)                                        # city ids must fit a byte
    if tab8 is not None:
        assert tab8.dtype == torch.uint8
        lr8, idx8, ban8, sel8 = mops.tsp_swap_delta_all(d, perms, None, 0.5, nearest=near32, random=rnd32, near_threshold=thr, seed=seed,
                                                        env_offset=off, return_selected=True, tables8=tab8)
        assert torch.equal(sel8, sel) and torch.equal(lr8, lr) and torch.equal(idx8, idx) and torch.equal(ban8, ban)
    h = B // 2
    lr_h, idx_h, ban_h = mops.tsp_swap_delta_all(d, perms[h:].contiguous(), None, 0.5, nearest=near32, random=rnd32, near_threshold=thr,
                                                 seed=seed, env_offset=off + h, tables8=tab8)
    assert torch.equal(lr_h, lr[h:]) and torch.equal(idx_h, idx[h:]) and torch.equal(ban_h, ban[h:])
    with pytest.raises((ValueError, RuntimeError)):
        mops.tsp_swap_delta_all(d, perms, None, 0.5)                          # no tables, nothing to draw from


def test_isco_tsp_class_opt_2_is_one_kernel_with_its_own_draws():
    """ISCO_TSP.opt_2(sample, T) -- the reference's signature -- draws inside the kernel from the object's seed stream; the
    near / far split is K / (K + 1) and `draw_partners` + selected= stays as the torch-generator form."""
    from rlsolver_amd.envs.env_ISCO import ISCO_TSP
    from rlsolver_amd.graph import generate_tsp_coords, tsp_tables
    dist, near, rnd = tsp_tables(generate_tsp_coords(60, seed=2), K=10)
    params = {"num_nodes": 60, "distance": dev(dist), "nearest_indices": dev(near), "random_indices": dev(rnd)}
    env = ISCO_TSP(params, batch_size=4096, device=DEV, K=10, seed=5)
    x = env.random_gen_init_sample()
    lr, idx, ban, sel = env.opt_2(x, 0.7, return_selected=True)
    near_hit = (env.nearest_indices[x] == sel[:, :, None]).any(dim=2).float().mean().item()
    # (the far branch picks among the first N - K - 1 of all other cities, which may be one of the K nearest too)
    want = 10 / 11 + (1 / 11) * (env.nearest_indices[:, :, None] == env.random_indices[:, None, :60 - 10 - 1]).any(dim=2).float().sum(dim=1).mean().item() / (60 - 10 - 1)
    assert abs(near_hit - want) < 5e-3
    lr_b, idx_b, ban_b = env.opt_2(x, 0.7)                                     # the next seed of the stream: other partners
    assert not torch.equal(idx, idx_b)
    s2 = env.draw_partners(x)
    lr_c, idx_c, ban_c = env.opt_2(x, 0.7, selected=s2)
    assert torch.equal(torch.gather(x, 1, idx_c), s2)                          # indices = where the partner city sits


@pytest.mark.parametrize("N,B", [(100, 1000), (52, 65), (200, 130), (7, 64), (256, 70), (300, 50)])
def test_tsp_random_properties(N, B):
    from rlsolver_amd.graph import generate_tsp_coords, tsp_tables
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, seed=N), K=min(20, N - 2))
    d = dev(dist)
    perms = mops.rand_perms(B, N, seed=77, device=DEV, env_offset=5)
    pn = perms.cpu().numpy()
    assert np.array_equal(pn, onp.rand_perms(B, N, 77, 5))
    assert np.array_equal(np.sort(pn, axis=1), np.tile(np.arange(N), (B, 1)))
    length = mops.tsp_tour_length(d, perms).cpu().numpy()
    np.testing.assert_allclose(length, onp.tsp_tour_length_f64(dist, pn), rtol=1e-5)
    # invariant under rotation and reversal
    np.testing.assert_allclose(mops.tsp_tour_length(d, torch.roll(perms, 3, 1).contiguous()).cpu().numpy(), length, rtol=1e-5)
    np.testing.assert_allclose(mops.tsp_tour_length(d, torch.flip(perms, [1]).contiguous()).cpu().numpy(), length, rtol=1e-5)
    rng = np.random.RandomState(1)
    # partner city = the city some 1..N-1 positions ahead (the sampler never draws a position's own
    # city: nearest/random tables exclude it, ISCO/util_TSP.py:9-16)
    off = rng.randint(1, N, size=(B, N))
    sel = np.take_along_axis(pn, (np.arange(N)[None, :] + off) % N, axis=1)
    lr_w, idx_w, ban_w = onp.tsp_swap_delta_all(dist, pn, sel, 0.5)
    lr, idx, ban = mops.tsp_swap_delta_all(d, perms, dev(sel), 0.5)
    assert np.array_equal(idx.cpu().numpy(), idx_w) and np.array_equal(ban.cpu().numpy(), ban_w)
    np.testing.assert_allclose(lr.cpu().numpy(), lr_w, rtol=1e-5, atol=1e-5 * length.max() / 0.5)
    # swap delta == length(after) - length(before) for one random non-banned position per env
    pos = np.array([rng.choice(np.flatnonzero(~ban_w[b])) if (~ban_w[b]).any() else -1 for b in range(B)])
    x = perms.clone()
    mops.tsp_apply_swap(x, dev(pos), idx)
    after = mops.tsp_tour_length(d, x).cpu().numpy()
    for b in range(B):
        if pos[b] >= 0:
            assert abs((after[b] - length[b]) - (-lr_w[b, pos[b]] * 0.5)) <= 2e-5 * length[b]
    # true 2-opt delta == length difference of the reversed segment
    i = rng.randint(0, N - 1, size=B)
    j = np.array([rng.randint(a + 1, N) for a in i])
    dl = mops.tsp_2opt_delta(d, perms, dev(i), dev(j)).cpu().numpy()
    rev = pn.copy()
    for b in range(B):
        rev[b, i[b]:j[b] + 1] = rev[b, i[b]:j[b] + 1][::-1]
    want = onp.tsp_tour_length_f64(dist, rev) - onp.tsp_tour_length_f64(dist, pn)
    np.testing.assert_allclose(dl, want, atol=2e-5 * length.max())


def test_isco_tsp_class_runs(golden):
    from rlsolver_amd.envs.env_ISCO import ISCO_TSP
    z = golden("tsp")
    params = {"num_nodes": 52, "distance": dev(z["berlin52/distance"]),
              "nearest_indices": dev(z["berlin52/nearest_indices"]), "random_indices": dev(z["berlin52/random_indices"])}
    torch.manual_seed(0)
    s = ISCO_TSP(params, batch_size=64, K=20, device=DEV)
    x = s.random_gen_init_sample(params)
    l0 = s.calculate_distance(x)
    temp = s.init_temperature
    for it in range(30):
        x, acc = s.step(x, 3, temp)
        assert 0.0 <= float(acc) <= 1.0
    xs = x.cpu().numpy()
    assert np.array_equal(np.sort(xs, axis=1), np.tile(np.arange(52), (64, 1)))
    l1 = s.calculate_distance(x)
    np.testing.assert_allclose(l1.cpu().numpy(), onp.tsp_tour_length_f64(z["berlin52/distance"], xs), rtol=1e-5)
    assert float(l1.mean()) < float(l0.mean())     # annealing at T=1 on berlin52 improves random tours


# ------------------------------------------------------------------ bit-packed chains and the on-device round
def _ba_data(n, m, seed):
    from rlsolver_amd.graph import generate_ba
    g = np.asarray(generate_ba(n, m, seed=seed), dtype=np.int64)
    return amcpg.make_data(n, g[:, 0].copy(), g[:, 1].copy(), DEV), g


@pytest.mark.parametrize("n,C", [(300, 192), (1001, 64 * 5 + 17), (64, 70)])
def test_packed_chains_roundtrip_and_layouts_agree(n, C):
    """pack / unpack are inverse; the packed metro and K7 kernels give bit for bit what the f32 node-major kernels give
    for the same seed (production draws are keyed by chain id, not by layout), for C not a multiple of 64 too."""
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    data, g = _ba_data(n, 4, 3)
    rng = np.random.RandomState(n)
    xs = dev((rng.rand(n, C) < 0.5).astype(np.float32))
    pk = PackedChains.pack(xs)
    assert pk.words.shape == ((C + 63) // 64, n) and torch.equal(pk.unpack(), xs)
    assert torch.equal(PackedChains.pack(xs.bool()).words, pk.words)
    probs = dev((rng.rand(n) * 0.6 + 0.2).astype(np.float32))
    T = max(1, n // 10)
    # metro, production draws: f32 kernel vs packed kernel, one chunk with per-round accept counts
    a32 = torch.zeros(T, dtype=torch.int64, device=DEV)
    apk = torch.zeros(T, dtype=torch.int64, device=DEV)
    o32 = xs.clone()
    mops.mcpg_metro_rounds(o32, probs, T, seed=77, accepts=a32)
    opk = pk.clone()
    mops.mcpg_metro_rounds(opk, probs, T, seed=77, accepts=apk)
    assert torch.equal(opk.unpack(), o32) and torch.equal(a32, apk) and int(a32.sum()) > 0
    # the reference-shaped function (pack -> packed walk -> unpack) with recorded draws == the oracle
    index = rng.randint(0, n, size=(5 * T, C)).astype(np.int64)
    u = rng.rand(5 * T, C).astype(np.float32)
    want, _ = onp.metro_sampling(probs.cpu().numpy(), xs.cpu().numpy(), T, index, u)
    got = amcpg.metro_sampling(probs, xs, T, device=DEV, index=dev(index), u=dev(u))
    assert np.array_equal(got.cpu().numpy(), want)
    # K7 levels: f32 in / f32 out vs packed in / packed out (in place), same seed
    x32, e32 = mops.mcpg_local_search_levels(data.graph, o32, data._lv_ptr, data._lv_data, 3, seed=5)
    xpk, epk = mops.mcpg_local_search_levels(data.graph, opk, data._lv_ptr, data._lv_data, 3, seed=5, out=opk)
    assert xpk is opk and torch.equal(xpk.unpack(), x32) and torch.equal(e32, epk)
    cut = ops_obj(data, x32)
    assert torch.equal((data.num_edges - 2 * cut).float(), e32)


def ops_obj(data, xs_nm):
    from rlsolver_amd import ops
    return ops.maxcut_obj(data.graph, (xs_nm.t() > 0).contiguous())


def test_packed_broadcast_start_pick_and_merge_vs_oracle():
    """C_in broadcast (xs_bool.repeat(1, R) never materialised), best-of-repeats on packed chains, and the best-merge
    of MCPG.py:376-391 against its numpy restatement."""
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    n, M, R = 500, 128, 6
    data, g = _ba_data(n, 5, 9)
    rng = np.random.RandomState(1)
    kept = dev((rng.rand(n, M) < 0.5).astype(np.float32))
    probs = dev((rng.rand(n) * 0.6 + 0.2).astype(np.float32))
    T = n // 10
    start = PackedChains.pack(kept)
    torch.manual_seed(4)
    got = amcpg.metro_sampling_packed(probs, start, T, num_chains=M * R)
    torch.manual_seed(4)
    want = amcpg.metro_sampling_packed(probs, PackedChains.pack(kept.repeat(1, R)), T)
    assert got.num_chains == M * R and torch.equal(got.words, want.words)
    assert not torch.equal(got.unpack()[:, :M], got.unpack()[:, M:2 * M])         # repeats walk with their own draws
    # sampler on packed chains == the f32 path for the same seed
    torch.manual_seed(6)
    vs_p, xg_p, val_p, xloc = amcpg.sampler_func_packed(data, got.clone(), 2, M, R)
    torch.manual_seed(6)
    vs_f, xg_f, val_f = amcpg.sampler_func(data, got.unpack(), 2, M, R, DEV)
    assert torch.equal(vs_p, vs_f) and torch.equal(xg_p.unpack(), xg_f) and torch.equal(val_p, val_f)
    assert torch.equal(ops_obj(data, xg_f).float(), vs_f)
    # merge: incumbents that are partly better, partly worse, with ties on both ends
    now_res = vs_f.cpu().numpy() + rng.randint(-3, 4, size=M).astype(np.float32)
    now_res[7] = now_res.max() + 2
    now_res[9] = now_res[7]                      # tie at the top: the first wins
    now_res[30] = now_res.min() - 1
    now_res[40] = now_res[30]                    # tie at the bottom
    now_info = (rng.rand(n, M) < 0.5).astype(np.float32)
    w_res, w_info, w_temp, w_max, w_idx = onp.mcpg_merge_best(vs_f.cpu().numpy(), xg_f.cpu().numpy(), now_res, now_info)
    d_res = dev(now_res)
    d_info = PackedChains.pack(dev(now_info))
    bv, bi = mops.mcpg_merge_best(vs_p, xg_p, d_res, d_info)
    assert np.array_equal(d_res.cpu().numpy(), w_res) and np.array_equal(d_info.unpack().cpu().numpy(), w_info)
    assert np.array_equal(xg_p.unpack().cpu().numpy(), w_temp)
    assert float(bv) == float(w_max) and int(bi) == w_idx == 7


def test_get_return_packed_vs_oracle_and_autograd():
    """get_return (MCPG.py:292-302) from the bit sums: objective and gradient against the float64 restatement, and
    against torch autograd on the reference's own [C, N] float expression."""
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    n, M, R = 333, 64, 3
    C = M * R
    rng = np.random.RandomState(2)
    s = (rng.rand(C, n) < 0.4).astype(np.float32)
    value = (rng.randn(C) * 20).astype(np.float32)
    lin = torch.tensor(rng.randn(n).astype(np.float32), device=DEV, requires_grad=True)
    probs = torch.sigmoid(lin)
    samples = PackedChains.pack(dev(s.T.copy()))
    obj = amcpg.get_return(probs, samples, dev(value), M, R)
    obj.backward()
    w_obj, w_grad_p = onp.mcpg_get_return(probs.detach().cpu().numpy(), s, value, M, R)
    p = probs.detach().cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(float(obj.detach()), w_obj, rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(lin.grad.cpu().numpy(), w_grad_p * p * (1 - p), rtol=2e-4, atol=2e-4)
    # the reference's expression through autograd (f32): same objective / gradient within f32 summation noise
    lin2 = torch.tensor(lin.detach().cpu().numpy(), device=DEV, requires_grad=True)
    p2 = torch.sigmoid(lin2)
    sd = dev(s)
    ref = ((sd * p2 + (1 - sd) * (1 - p2)).log().sum(dim=1) * dev(value)).mean()
    ref.backward()
    np.testing.assert_allclose(float(obj.detach()), float(ref.detach()), rtol=1e-4, atol=1e-2)
    np.testing.assert_allclose(lin.grad.cpu().numpy(), lin2.grad.cpu().numpy(), rtol=1e-3, atol=1e-3)
    # the reference-shaped call (float [C, N] samples) goes through the same kernel
    assert float(amcpg.get_return(torch.sigmoid(lin.detach()), sd, dev(value), M, R)) == pytest.approx(float(obj.detach()), rel=1e-6)


def test_mcpg_round_on_device_improves_and_prints_like_the_reference():
    """run_mcpg: the sampling loop of mcpg() (MCPG.py:353-413) on MCPGRound; incumbents never get worse, every kept
    value is the cut of its kept chain, the prints are the reference's."""
    from rlsolver_amd import ops
    n, M, R = 800, 64, 8
    data, g = _ba_data(n, 4, 11)
    torch.manual_seed(0)
    xs0 = ops.rand_spins(M, n, 3, DEV)
    vs0 = ops.maxcut_obj(data.graph, xs0).float()
    lines = []
    v, x, rates = amcpg.run_mcpg(data, xs0.t().contiguous().float(), vs0, M, R, num_ls=2, num_rounds=6, sample_epoch_num=2,
                                 log=lambda *a: lines.append(" ".join(str(t) for t in a)))
    assert len(rates) == 6 and all(r > 0 for r in rates)
    assert sum(l.startswith("value ") and "entropy" in l for l in lines) == 6
    assert sum(l.startswith("num_samples_per_second:") for l in lines) == 6
    assert v >= float(vs0.max()) and v > 0.6 * data.num_edges
    assert int(ops.maxcut_obj(data.graph, x[None, :].contiguous())) == int(v)
    # the round's get_return forms its two chain sums once and reuses them over the policy epochs: same objective and gradient as
    # the stand-alone function on the same samples, for any probs
    rnd = amcpg.MCPGRound(data, xs0.t().contiguous().float(), vs0, M, R, num_ls=2)
    rnd.step(torch.full((n,), 0.5, device=DEV))
    for k in range(3):
        pr = (torch.rand(n, device=DEV, generator=torch.Generator(device=DEV).manual_seed(k)) * 0.8 + 0.1).requires_grad_()
        a = rnd.get_return(pr)
        ga, = torch.autograd.grad(a, pr)
        pr2 = pr.detach().clone().requires_grad_()
        b = amcpg.get_return(pr2, rnd.samples, rnd.value)
        gb, = torch.autograd.grad(b, pr2)
        assert torch.equal(a.detach(), b.detach()) and torch.equal(ga, gb)
    sums = rnd._sums
    rnd.step(torch.full((n,), 0.5, device=DEV))
    assert rnd._sums is None and sums is not None            # a new round drops them


# ------------------------------------------------------------------ weighted sampler of the upstream MCPG package
@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_mcpg_weighted_sampler_golden(golden, gname):
    """mcpg_sampling_maxcut (methods/MCPG/sampling.py:89-127: gauge fix, weighted node-sequential search, weighted
    expected value) against the reference's trace with every torch draw recorded: bit for bit."""
    from rlsolver_amd.methods import MCPG_maxcut as wm
    z = golden("mcpg_weighted")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    data = wm.make_data(n, g[:, 0], g[:, 1], g[:, 2], DEV, sorted_degree_nodes=z[f"{gname}/sorted_degree_nodes"])
    assert data.weighted_degree == z[f"{gname}/weighted_degree"].tolist()
    assert data.edge_weight_sum == float(z[f"{gname}/edge_weight_sum"])
    vs, xs_good, start, value = wm.mcpg_sampling_maxcut(
        data, dev(z[f"{gname}/start"], torch.float32), dev(z[f"{gname}/probs"]), int(z[f"{gname}/num_ls"]),
        int(z[f"{gname}/change_times"]), int(z[f"{gname}/M"]), DEV, index=dev(z[f"{gname}/metro_index"]),
        u=dev(z[f"{gname}/metro_u"]), uniforms=dev(z[f"{gname}/uniforms"]))
    assert np.array_equal(start.cpu().numpy().astype(np.uint8), z[f"{gname}/metro_out"])
    assert np.array_equal(vs.cpu().numpy(), z[f"{gname}/vs"])
    assert np.array_equal(xs_good.cpu().numpy(), z[f"{gname}/xs_good"])
    np.testing.assert_allclose(value.cpu().numpy(), z[f"{gname}/value"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("n,m,M,R,num_ls,wset", [(300, 1500, 16, 4, 2, (-1, 1)), (2000, 19990, 64, 2, 1, (-3, -1, 1, 2)),
                                                 (130, 600, 1, 3, 3, (1,))])
def test_mcpg_weighted_sampler_vs_oracle(n, m, M, R, num_ls, wset):
    """Larger / denser weighted graphs (rows longer than the record's first 28 neighbours, multi-batch levels) against
    the numpy restatement; production draws: the gauge node ends at 0 in every chain and the kept value is the weight
    of the kept chain's cut."""
    from rlsolver_amd.methods import MCPG_maxcut as wm
    g = gnm_arr(n, m, seed=8)
    rng = np.random.RandomState(n)
    g[:, 2] = rng.choice(wset, size=len(g))
    C = M * R
    adeg = np.zeros(n)
    np.add.at(adeg, g[:, 0], np.abs(g[:, 2]))
    np.add.at(adeg, g[:, 1], np.abs(g[:, 2]))
    order = np.argsort(-adeg, kind="stable")
    data = wm.make_data(n, g[:, 0], g[:, 1], g[:, 2], DEV, sorted_degree_nodes=order)
    probs = (rng.rand(n) * 0.6 + 0.2).astype(np.float32)
    start = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    T = max(1, n // 10)
    index = rng.randint(0, n, size=(5 * T, C)).astype(np.int64)
    u = rng.rand(5 * T, C).astype(np.float32)
    uni = rng.rand(num_ls, n, C).astype(np.float32)
    w_vs, w_xs, w_start, w_val, w_exp = onp.mcpg_sampling_maxcut(g, n, order, start, probs, num_ls, T, M, index, u, uni)
    vs, xs_good, st, value = wm.mcpg_sampling_maxcut(data, dev(start), dev(probs), num_ls, T, M, DEV, index=dev(index), u=dev(u),
                                                     uniforms=dev(uni))
    assert np.array_equal(st.cpu().numpy(), w_start) and np.array_equal(vs.cpu().numpy(), w_vs)
    assert np.array_equal(xs_good.cpu().numpy(), w_xs)
    np.testing.assert_allclose(value.cpu().numpy(), w_val, rtol=0, atol=2e-3)
    torch.manual_seed(0)
    vs, xs_good, st, value = wm.mcpg_sampling_maxcut(data, dev(start), dev(probs), num_ls, T, M, DEV)
    xg = xs_good.cpu().numpy()
    cutw = ((xg[g[:, 0]] != xg[g[:, 1]]) * g[:, 2][:, None]).sum(axis=0)
    assert np.array_equal(vs.cpu().numpy(), cutw.astype(np.float32))
    assert set(np.unique(xg)) <= {0.0, 1.0}


def test_tsp_2opt_local_search_golden(golden):
    """local_search_2_opt on the device (one kernel per pass) returns the reference's tours and float64 distances on
    a5, berlin52 and two uniform instances, until-no-improvement and for exactly two passes; the batch form ends in
    2-opt local optima no longer than it started and agrees with the oracle."""
    from rlsolver_amd.methods import tsp_opt_2 as t2
    z = golden("tsp_2opt")
    for name in z["names"]:
        d = z[f"{name}/distance_f64"]
        for t in range(2):
            start = [z[f"{name}/t{t}/start_tour"].tolist(), float(z[f"{name}/t{t}/start_distance"])]
            assert t2.distance_calc(d, start) == start[1]
            for rs in (-1, 2):
                route, dist = t2.local_search_2_opt(d, start, recursive_seeding=rs, verbose=False, device=DEV)
                assert route == z[f"{name}/t{t}/rs{rs}/tour"].tolist(), (name, t, rs)
                assert dist == float(z[f"{name}/t{t}/rs{rs}/distance"])
    rng = np.random.RandomState(3)
    N, B = 30, 40
    coords = rng.rand(N, 2) * 50
    d = np.sqrt(((coords[:, None] - coords[None]) ** 2).sum(-1))
    perms = np.stack([rng.permutation(N) for _ in range(B)])
    dd = dev(d)
    start_len = dd[dev(perms), torch.roll(dev(perms), -1, 1)].sum(1)
    out, lengths = t2.local_search_2_opt_batch(d, dev(perms))                       # delta ranking
    assert bool((lengths <= start_len + 1e-9).all()) and bool((out.sort(dim=1).values == torch.arange(N, device=DEV)).all())
    bi, bj, bd = mops.tsp_2opt_best(dd, out)
    assert bool((bd == 0).all()) and bool((bi == -1).all())                      # 2-opt local optima
    seed_len = torch.tensor([onp.tsp_distance_calc(d, [int(c) + 1 for c in p] + [int(p[0]) + 1]) for p in perms], dtype=torch.float64, device=DEV)
    oute, lene = t2.local_search_2_opt_batch(d, dev(perms), exact=True, lengths=seed_len)   # the reference's ranking
    for b in range(6):
        tour = [int(c) + 1 for c in perms[b]] + [int(perms[b][0]) + 1]
        r, dist = onp.tsp_local_search_2_opt(d, tour, onp.tsp_distance_calc(d, tour), -1)
        assert [int(c) + 1 for c in oute[b].tolist()] == r[:-1] and float(lene[b]) == dist
    asym = d + np.triu(np.ones_like(d), 1)                                         # asymmetric: exact ranking only
    tour = [int(c) + 1 for c in perms[0]] + [int(perms[0][0]) + 1]
    r, dist = onp.tsp_local_search_2_opt(asym, tour, onp.tsp_distance_calc(asym, tour), 3)
    r2, dist2 = t2.local_search_2_opt(asym, [tour, onp.tsp_distance_calc(asym, tour)], recursive_seeding=3, verbose=False, device=DEV)
    assert r2 == r and dist2 == dist
    with pytest.raises(ValueError):
        t2.local_search_2_opt_batch(d + np.triu(np.ones_like(d), 1), dev(perms))   # asymmetric
    for bad_route in ([1, 2, 3, 1], list(range(1, N + 1)) + [2], [1] + list(range(1, N)) + [1]):
        with pytest.raises(ValueError):
            t2.local_search_2_opt(d, [bad_route, 1.0], verbose=False, device=DEV)
    # any number of workgroups per tour picks the same pair (ties included: a matrix of small integers)
    di = dev(np.rint(d / 10.0))
    cur = di[dev(perms), torch.roll(dev(perms), -1, 1)].sum(1)
    for cl in (None, cur):
        want = mops.tsp_2opt_best(di, dev(perms), cl, slices=1)
        for sl in (2, 7, None):
            got = mops.tsp_2opt_best(di, dev(perms), cl, slices=sl)
            assert all(torch.equal(g, w) for g, w in zip(got, want)), (cl is None, sl)


@pytest.mark.parametrize("M", [1, 5, 70])
def test_merge_best_when_best_and_worst_incumbent_coincide(M):
    """MCPG.py:383-391 with argmax == argmin (one kept chain, or every incumbent equal after the merge): the incumbent
    column is left alone but temp_max_info's column STILL takes it -- it seeds the next round.  (Found by
    tools/fuzz/fuzz_mcpg.py: the kernel used to return early.)"""
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    rng = np.random.RandomState(M)
    n = 77
    temp_max = np.full(M, 10.0, np.float32)
    now_res = np.full(M, 12.0, np.float32)                      # all incumbents better and equal: argmax = argmin = 0
    temp_info = (rng.rand(n, M) < 0.5).astype(np.float32)
    now_info = (rng.rand(n, M) < 0.5).astype(np.float32)
    w_res, w_info, w_temp, w_max, w_idx = onp.mcpg_merge_best(temp_max, temp_info, now_res, now_info)
    assert w_idx == 0 and np.array_equal(w_temp[:, 0], now_info[:, 0])
    d_res, d_info, d_temp = dev(now_res), PackedChains.pack(dev(now_info)), PackedChains.pack(dev(temp_info))
    bv, bi = mops.mcpg_merge_best(dev(temp_max), d_temp, d_res, d_info)
    assert np.array_equal(d_res.cpu().numpy(), w_res) and np.array_equal(d_info.unpack().cpu().numpy(), w_info)
    assert np.array_equal(d_temp.unpack().cpu().numpy(), w_temp) and float(bv) == float(w_max) and int(bi) == w_idx


@pytest.mark.parametrize("M", [64, 128, 192, 320, 70])
def test_merge_best_mask_scratch_is_exactly_ceil_M_over_64_words(M):
    """rls_mcpg_merge_best's mask_scratch is uint64[ceil(M / 64)]: a canary word right behind a sub-allocated scratch
    must survive (the mask kernel's wave leaders used to store one word past it whenever M % 64 != 1)."""
    from rlsolver_amd import _abi
    from rlsolver_amd.ops_mcpg_tsp import PackedChains, _ptr, _stream
    rng = np.random.RandomState(M)
    n = 50
    temp_max = rng.randint(0, 50, M).astype(np.float32)
    now_res = rng.randint(0, 50, M).astype(np.float32)
    temp_info = (rng.rand(n, M) < 0.5).astype(np.float32)
    now_info = (rng.rand(n, M) < 0.5).astype(np.float32)
    w_res, w_info, w_temp, w_max, w_idx = onp.mcpg_merge_best(temp_max, temp_info, now_res, now_info)
    d_res, d_info, d_temp = dev(now_res), PackedChains.pack(dev(now_info)), PackedChains.pack(dev(temp_info))
    tiles = (M + 63) // 64
    CANARY = 0x5A5A5A5A5A5A5A5A
    arena = torch.full((tiles + 8,), CANARY, dtype=torch.int64, device=DEV)
    bv = torch.empty(1, dtype=torch.float32, device=DEV)
    bi = torch.empty(1, dtype=torch.int64, device=DEV)
    _abi.call("rls_mcpg_merge_best", _ptr(dev(temp_max)), _ptr(d_temp.words), _ptr(d_res), _ptr(d_info.words), n, M,
              _ptr(arena), _ptr(bv), _ptr(bi), 1, _stream(torch.device(DEV)))
    assert arena[tiles:].eq(CANARY).all(), "a word behind mask_scratch was written"
    assert np.array_equal(d_res.cpu().numpy(), w_res) and np.array_equal(d_info.unpack().cpu().numpy(), w_info)
    assert np.array_equal(d_temp.unpack().cpu().numpy(), w_temp) and float(bv) == float(w_max) and int(bi) == w_idx


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_mcpg_glue_golden_three_rounds(golden, gname):
    """mcpg_glue.npz: three consecutive rounds of the reference's outer loop (metro_sampling -> sampler_func -> the
    best-merge of MCPG.py:376-391, exec'd from the reference file -> get_return with its autograd gradient).  The HIP
    path -- metro walk, level sampler, rls_mcpg_merge_best on bit-packed chains, rls_mcpg_value_bit_sums -- reproduces
    every recorded state bit for bit (get_return: within f32 summation noise of the reference's f32 loop)."""
    from rlsolver_amd.ops_mcpg_tsp import PackedChains
    z = golden("mcpg_glue")
    graph = z[f"{gname}/graph"]
    n = int(graph[:, :2].max()) + 1
    ei = graph[:, :2].T.copy()
    M, R, num_ls, T = (int(z[f"{gname}/{k}"]) for k in ("M", "R", "num_ls", "T"))
    data = amcpg.make_data(n, ei[0], ei[1], DEV, sorted_degree_nodes=z[f"{gname}/sorted_degree_nodes"])
    probs = dev(z[f"{gname}/probs"])
    for rnd in range(3):
        t = f"{gname}/round{rnd}"
        xs = amcpg.metro_sampling(probs, dev(z[f"{t}/start"], torch.float32), T, device=DEV, index=dev(z[f"{t}/metro_index"]),
                                  u=dev(z[f"{t}/metro_u"]))
        assert np.array_equal(xs.cpu().numpy().astype(np.uint8), z[f"{t}/xs_sample"])
        temp_max, temp_info, value = amcpg.sampler_func(data, xs, num_ls, M, R, DEV, uniforms=dev(z[f"{t}/uniforms"]))
        assert np.array_equal(temp_max.cpu().numpy(), z[f"{t}/temp_max"])
        assert np.array_equal(temp_info.cpu().numpy(), z[f"{t}/temp_max_info"])
        np.testing.assert_allclose(value.cpu().numpy(), z[f"{t}/value"], rtol=0, atol=1e-4)
        # best-merge on the device, bit-packed
        d_res = dev(z[f"{t}/now_max_res_before"])
        d_info = PackedChains.pack(dev(z[f"{t}/now_max_info_before"]))
        d_temp = PackedChains.pack(temp_info.contiguous())
        bv, bi = mops.mcpg_merge_best(temp_max.contiguous(), d_temp, d_res, d_info)
        assert np.array_equal(d_res.cpu().numpy(), z[f"{t}/now_max_res_after"])
        assert np.array_equal(d_info.unpack().cpu().numpy(), z[f"{t}/now_max_info_after"])
        assert np.array_equal(d_temp.unpack().cpu().numpy(), z[f"{t}/temp_max_info_after"])
        assert float(bv) == float(z[f"{t}/now_max"]) and int(bi) == int(z[f"{t}/now_max_index"])
        # get_return: value and gradient as autograd gives them in the reference (f32 there)
        pl = probs.clone().requires_grad_(True)
        obj = amcpg.get_return(pl, xs.t().contiguous(), dev(z[f"{t}/value"]), M, R)
        obj.backward()
        np.testing.assert_allclose(float(obj.detach()), float(z[f"{t}/get_return"]), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(pl.grad.cpu().numpy(), z[f"{t}/get_return_grad"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
def test_mcpg_data_object_golden(golden, gname):
    """The lists maxcut_dataloader hangs on its Data object (MCPG.py:187-289), built lazily here, against the reference's."""
    z = golden("mcpg_data")
    n, ei = int(z[f"{gname}/num_nodes"]), z[f"{gname}/edge_index"]
    data = amcpg.make_data(n, ei[0], ei[1], DEV)
    assert "neighbors" not in data.__dict__                         # nothing built until somebody asks
    off = z[f"{gname}/neighbors_offsets"]
    assert len(data.neighbors) == n and data.neighbors[0].dtype == torch.int64 and data.neighbors[0].device.type == "cuda"
    assert np.array_equal(torch.cat(data.neighbors).cpu().numpy(), z[f"{gname}/neighbors_flat"])
    assert [int(t.numel()) for t in data.neighbors] == np.diff(off).tolist()
    assert np.array_equal(np.array([list(t.shape) for t in data.neighbor_edges]), z[f"{gname}/neighbor_edges_shapes"])
    assert np.array_equal(torch.cat([t.reshape(-1) for t in data.neighbor_edges]).cpu().numpy(), z[f"{gname}/neighbor_edges_flat"])
    assert data.single_degree == z[f"{gname}/single_degree"].tolist()
    assert data.weighted_degree == z[f"{gname}/weighted_degree"].tolist()
    assert data.add_items.dtype == torch.float32
    assert np.array_equal(data.add_items.cpu().numpy(), z[f"{gname}/add_items"])
    # both argsorts are unstable in the reference: any order with non-increasing keys is a valid outcome
    wd = np.abs(z[f"{gname}/weighted_degree"]).astype(np.float32)
    key_e = wd[ei[0]] + wd[ei[1]]
    for mine, ref, key in ((data.sorted_degree_edges.cpu().numpy(), z[f"{gname}/sorted_degree_edges"], key_e),
                           (data.sorted_degree_nodes.cpu().numpy(), z[f"{gname}/sorted_degree_nodes"], wd)):
        assert sorted(mine.tolist()) == list(range(key.size))
        assert np.all(np.diff(key[mine]) <= 0) and np.all(np.diff(key[ref]) <= 0)
        assert np.array_equal(key[mine], key[ref])
    assert amcpg.append_neighbors(data) is data
