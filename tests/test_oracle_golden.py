"""Pin the numpy oracle against golden vectors captured from the imported reference
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import oracle_np as onp

GRAPH_NAMES = ["BA_5_ID0", "BA_5_ID1", "PL_20_ID0", "BA_100_ID0", "ER_100_ID0", "PL_100_ID0", "gset_14_stub"]


@pytest.mark.parametrize("gname", GRAPH_NAMES)
@pytest.mark.parametrize("bidir", [0, 1])
def test_maxcut_obj_and_cutdeg(golden, gname, bidir):
    z = golden("maxcut_obj")
    graph = z[f"{gname}/graph"]
    n = int(z[f"{gname}/bidir{bidir}/num_nodes"])
    assert n == onp.num_nodes_distinct(graph)
    u, _ = onp.stored_edges(graph, bool(bidir))
    assert np.array_equal(np.bincount(u, minlength=n)[None, :], z[f"{gname}/bidir{bidir}/n0_num_n1"])
    for seed in (0, 1, 2):
        t = f"{gname}/bidir{bidir}/seed{seed}"
        xs = z[f"{t}/xs"]
        assert (xs[:, 0] == 0).all()
        obj = onp.maxcut_obj(xs, graph, bool(bidir))
        assert obj.dtype == np.int64 and str(z[f"{t}/obj_dtype"]) == "torch.int64"
        assert np.array_equal(obj, z[f"{t}/obj"])
        raw = onp.maxcut_obj_for_loop(xs, graph, n, bool(bidir), if_sum=False)
        assert np.array_equal(raw, z[f"{t}/cutdeg"])
        assert str(z[f"{t}/cutdeg_dtype"]) == ("torch.float32" if bidir else "torch.int64")
        loop = onp.maxcut_obj_for_loop(xs, graph, n, bool(bidir), if_sum=True)
        assert np.array_equal(loop, z[f"{t}/obj_loop"])
        if f"{t}/edge_mask" in z.files and not bidir:
            assert np.array_equal(onp.maxcut_edge_mask(xs, graph, False).astype(np.uint8), z[f"{t}/edge_mask"])


def test_maxcut_delta_matches_definition_and_weighted_gain(golden):
    z = golden("weighted_gain")
    g = z["graph"]
    xs = z["xs"]
    d = onp.maxcut_delta_all(xs, g, 100, weights=g[:, 2])
    # compute_gain (methods_problem_specific/maxcut/util.py:67-76) = +w if same side else -w = cut gain of the flip
    assert np.array_equal(d, z["gain"])
    u, v, w = g[:, 0], g[:, 1], g[:, 2]
    cut = ((xs[:, u] != xs[:, v]) * w).sum(1)
    assert np.array_equal(cut, z["cut"])


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0", "ER_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [0, 1])
def test_greedy_sweep_and_local_search(golden, gname, bidir):
    z = golden("maxcut_local_search")
    graph = z[f"{gname}/graph"]
    n = onp.num_nodes_distinct(graph)
    tag = f"{gname}/bidir{bidir}"
    xs = z[f"{tag}/sweep/xs_in"].astype(bool)
    vs = onp.maxcut_obj(xs, graph, bool(bidir))
    onp.greedy_sweep(xs, vs, graph, bool(bidir))
    assert np.array_equal(xs.astype(np.uint8), z[f"{tag}/sweep/xs_out"])
    assert np.array_equal(vs, z[f"{tag}/sweep/vs_out"])
    xs = z[f"{tag}/ls/xs_in"].astype(bool)
    gx, gv = onp.local_search_inplace(xs, graph, n, bool(bidir), z[f"{tag}/ls/noise"], num_iters=8,
                                      num_spin=int(z[f"{tag}/ls/num_spin"]), noise_std=0.3)
    assert np.array_equal(gx.astype(np.uint8), z[f"{tag}/ls/xs_out"])
    assert np.array_equal(gv, z[f"{tag}/ls/vs_out"])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_local_search_class(golden, gname):
    z = golden("local_search_class")
    graph = z[f"{gname}/graph"]
    n = onp.num_nodes_distinct(graph)
    tag = f"{gname}/bidir0"
    good_xs = z[f"{tag}/xs_in"].astype(bool)
    good_vs = onp.maxcut_obj(good_xs, graph, False)
    assert np.array_equal(good_vs, z[f"{tag}/vs_reset"])
    for r in range(2):
        good_xs, good_vs, nupd = onp.local_search_class_random_search(
            good_xs, good_vs, graph, n, z[f"{tag}/round{r}/noise"], num_iters=4, num_spin=4)
        assert np.array_equal(good_xs.astype(np.uint8), z[f"{tag}/round{r}/xs"])
        assert np.array_equal(good_vs, z[f"{tag}/round{r}/vs"])
        assert nupd == int(z[f"{tag}/round{r}/num_update"])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [0, 1])
def test_env_ppo(golden, gname, bidir):
    z = golden("env_ppo")
    graph = z[f"{gname}/graph"]
    tag = f"{gname}/bidir{bidir}"
    n = int(graph[:, :2].max()) + 1
    env = onp.PPOEnvOracle(graph, n, 20, bool(bidir))
    env.reset_to(z[f"{tag}/xs0"])
    assert np.array_equal(env.last, z[f"{tag}/cut0"])
    for t in range(50):
        xs, r, d, c = env.step(z[f"{tag}/actions"][t])
        assert np.array_equal(r, z[f"{tag}/rewards"][t])
        assert np.array_equal(d, z[f"{tag}/dones"][t])
        assert np.array_equal(c, z[f"{tag}/curs"][t])
    assert np.array_equal((env.xs > 0).astype(np.uint8), z[f"{tag}/xs_final"])
    assert list(z[f"{tag}/ret_dtypes"]) == ["torch.float32"] * 4


@pytest.mark.parametrize("gname", ["BA_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [0, 1])
def test_env_ppo_ref_shaped_torch_baseline(golden, gname, bidir):
    """The torch-CPU op chain bench.py times as cpu_baseline_ref_shaped is the reference's env_PPO.step, bit for bit."""
    import torch
    from oracle.oracle_torch import PPOEnvRefShaped
    z = golden("env_ppo")
    graph = z[f"{gname}/graph"]
    tag = f"{gname}/bidir{bidir}"
    n = int(graph[:, :2].max()) + 1
    env = PPOEnvRefShaped(graph, n, z[f"{tag}/xs0"].shape[0], 20, bool(bidir))
    env.reset_to(z[f"{tag}/xs0"])
    assert np.array_equal(env.last_reward.numpy(), z[f"{tag}/cut0"])
    for t in range(50):
        xs, r, d, c = env.step(torch.from_numpy(z[f"{tag}/actions"][t]))
        assert np.array_equal(r.numpy(), z[f"{tag}/rewards"][t]) and np.array_equal(d.numpy(), z[f"{tag}/dones"][t])
        assert np.array_equal(c.numpy(), z[f"{tag}/curs"][t])
    assert np.array_equal((env.xs > 0).numpy().astype(np.uint8), z[f"{tag}/xs_final"])


def test_select_ops(golden):
    z = golden("select_ops")
    for mx in (1, 0):
        a, b = z["update/xs0"].astype(bool), z["update/vs0"].copy()
        ret = onp.update_xs_by_vs(a, b, z["update/xs1"].astype(bool), z["update/vs1"], bool(mx))
        assert np.array_equal(a.astype(np.uint8), z[f"update/max{mx}/xs"])
        assert np.array_equal(b, z[f"update/max{mx}/vs"])
        assert ret == int(z[f"update/max{mx}/ret"]) == a.shape[0]
        gx, gv = onp.pick_xs_by_vs(z["update/xs0"], z["update/vs0"], int(z["pick/R"]), bool(mx))
        assert np.array_equal(gx, z[f"pick/max{mx}/xs"])
        assert np.array_equal(gv, z[f"pick/max{mx}/vs"])
    a, b = z["update/xs0"].copy(), z["evo/vs_in"].copy()
    onp.evolutionary_replacement(a, b, 5, z["evo/max1/perm"], True)
    assert np.array_equal(a, z["evo/max1/xs"])
    assert np.array_equal(b, z["evo/max1/vs"])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_mcpg(golden, gname):
    z = golden("mcpg")
    graph = z[f"{gname}/graph"]
    n = int(graph[:, :2].max()) + 1
    ei = z[f"{gname}/edge_index"]
    assert np.array_equal(ei, graph[:, :2].T)
    nb = onp.mcpg_neighbors(ei, n)
    assert np.array_equal(np.asarray([len(x) for x in nb], np.float64), z[f"{gname}/weighted_degree"])
    out, t_used = onp.metro_sampling(z[f"{gname}/metro/probs"], z[f"{gname}/metro/start"],
                                     int(z[f"{gname}/metro/T"]), z[f"{gname}/metro/index"], z[f"{gname}/metro/u"])
    assert t_used == z[f"{gname}/metro/index"].shape[0]
    assert np.array_equal(out.astype(np.uint8), z[f"{gname}/metro/out"])
    assert str(z[f"{gname}/metro/out_dtype"]) == "torch.float32"
    vs_good, xs_good, value, _, _ = onp.sampler_func(
        ei, n, z[f"{gname}/sorted_degree_nodes"], z[f"{gname}/sampler/xs_in"],
        int(z[f"{gname}/sampler/num_ls"]), int(z[f"{gname}/sampler/total_mcmc_num"]),
        int(z[f"{gname}/sampler/repeat_times"]), z[f"{gname}/sampler/uniforms"])
    assert np.array_equal(vs_good, z[f"{gname}/sampler/vs_good"])
    assert np.array_equal(xs_good, z[f"{gname}/sampler/xs_good"])
    np.testing.assert_allclose(value, z[f"{gname}/sampler/value"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("name", ["a5", "berlin52"])
def test_tsp(golden, name):
    z = golden("tsp")
    from rlsolver_amd.graph import tsp_tables
    K = int(z[f"{name}/K"])
    dist, nearest, rnd = tsp_tables(z[f"{name}/coords"], K)
    np.testing.assert_allclose(dist, z[f"{name}/distance"], rtol=1e-6, atol=1e-3)
    assert np.array_equal(rnd, z[f"{name}/random_indices"])
    # torch.topk and a stable argsort may order exactly-tied distances differently: compare the
    # neighbour distances, and the indices wherever the distance is not tied
    gn = z[f"{name}/nearest_indices"]
    gd = np.take_along_axis(z[f"{name}/distance"], gn, axis=1)
    md = np.take_along_axis(z[f"{name}/distance"], nearest, axis=1)
    assert np.array_equal(gd, md)
    tied = np.zeros_like(gd, dtype=bool)
    tied[:, 1:] |= gd[:, 1:] == gd[:, :-1]
    tied[:, :-1] |= gd[:, :-1] == gd[:, 1:]
    assert np.array_equal(nearest[~tied], gn[~tied])
    d = z[f"{name}/distance"]
    perms = z[f"{name}/perms"]
    np.testing.assert_allclose(onp.tsp_tour_length(d, perms), z[f"{name}/length_f32"], rtol=1e-6)
    np.testing.assert_allclose(onp.tsp_tour_length_f64(d, perms), z[f"{name}/length_f64_distance_calc"], rtol=1e-12)
    np.testing.assert_allclose(z[f"{name}/length_f32"], z[f"{name}/length_f64_distance_calc"], rtol=1e-5)
    sel = onp.tsp_selected_partner(perms, z[f"{name}/nearest_indices"], z[f"{name}/random_indices"],
                                   z[f"{name}/opt2/rand"], z[f"{name}/opt2/randint_nearest"],
                                   z[f"{name}/opt2/randint_random"], K)
    lr, idx, ban = onp.tsp_swap_delta_all(d, perms, sel, float(z[f"{name}/opt2/temperature"]))
    assert np.array_equal(idx, z[f"{name}/opt2/indices"])
    assert np.array_equal(ban.astype(np.uint8), z[f"{name}/opt2/ban"])
    np.testing.assert_allclose(lr, z[f"{name}/opt2/logratio"], rtol=1e-5, atol=1e-2 if name == "a5" else 1e-3)
    sw = onp.tsp_switch(perms, z[f"{name}/switch/pos"], idx)
    assert np.array_equal(sw, z[f"{name}/switch/out"])
    # swap delta == length difference of the applied swap (non-banned positions)
    T = float(z[f"{name}/opt2/temperature"])
    pos = z[f"{name}/switch/pos"]
    for b in range(perms.shape[0]):
        if pos[b] >= 0:
            want = z[f"{name}/switch/length_f32"][b] - z[f"{name}/length_f32"][b]
            got = -lr[b, pos[b]] * T
            assert abs(got - want) <= 1e-5 * z[f"{name}/length_f32"][b] + 1e-3
    dl = onp.tsp_2opt_delta(d, perms, z[f"{name}/twoopt/env"], z[f"{name}/twoopt/i"], z[f"{name}/twoopt/j"])
    np.testing.assert_allclose(dl, z[f"{name}/twoopt/delta_f64"], rtol=1e-9, atol=1e-6)


def test_encoder_base64(golden):
    z = golden("encoder_base64")
    for n in (5, 100, 800):
        x = z[f"n{n}/x"].astype(bool)
        s = str(z[f"n{n}/str"])
        assert onp.b64_bool_to_str(x, n) == s
        assert np.array_equal(onp.b64_str_to_bool(s, n), x)
    for k in ("G14", "G15", "G22", "G49", "G50", "G55", "G70"):
        n = int(z[f"kat/{k}/num_nodes"])
        assert np.array_equal(onp.b64_str_to_bool(str(z[f"kat/{k}/str"]), n).astype(np.uint8), z[f"kat/{k}/x"])


def test_philox_known_answer():
    # Random123 known-answer test vectors for philox4x32-10
    r = onp.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = onp.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = onp.philox4x32_10(0xa4093822, 0x299f31d0, 0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    xs = onp.rand_spins(7, 300, seed=12345, env_offset=3)
    assert xs.shape == (7, 300) and (xs[:, 0] == 0).all() and 0.4 < xs.mean() < 0.6
    # sharding invariance: rows depend on the global env id only
    assert np.array_equal(onp.rand_spins(3, 300, 12345, env_offset=5), xs[2:5])


# ------------------------------------------------------------------ the C oracle (oracle/oracle.c)
@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0", "ER_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [0, 1])
def test_c_oracle_maxcut(golden, gname, bidir):
    from oracle import oracle_c as oc
    z = golden("maxcut_obj")
    graph = z[f"{gname}/graph"]
    n = int(z[f"{gname}/bidir{bidir}/num_nodes"])
    eu, ev = onp.stored_edges(graph, bool(bidir))
    erp = np.concatenate([[0], np.cumsum(np.bincount(eu, minlength=n))])
    for seed in (0, 1, 2):
        t = f"{gname}/bidir{bidir}/seed{seed}"
        assert np.array_equal(oc.maxcut_obj(z[f"{t}/xs"], eu, ev, bidir), z[f"{t}/obj"])
        want = z[f"{t}/cutdeg"]
        assert np.array_equal(oc.node_cutdeg(z[f"{t}/xs"], erp, ev), (want * 2 if bidir else want).astype(np.int64))
    zs = golden("maxcut_local_search")
    tag = f"{gname}/bidir{bidir}"
    xs = zs[f"{tag}/sweep/xs_in"].copy()
    vs = oc.maxcut_obj(xs, eu, ev, bidir)
    oc.greedy_sweep(xs, vs, eu, ev, bidir)
    assert np.array_equal(xs, zs[f"{tag}/sweep/xs_out"])
    assert np.array_equal(vs, zs[f"{tag}/sweep/vs_out"])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "gset_14_stub"])
@pytest.mark.parametrize("bidir", [0, 1])
def test_c_oracle_ppo_step(golden, gname, bidir):
    from oracle import oracle_c as oc
    z = golden("env_ppo")
    graph = z[f"{gname}/graph"]
    tag = f"{gname}/bidir{bidir}"
    eu, ev = onp.stored_edges(graph, bool(bidir))
    xs = z[f"{tag}/xs0"].astype(np.float32)
    last = z[f"{tag}/cut0"].copy()
    xs8 = z[f"{tag}/xs0"].copy()
    last8 = z[f"{tag}/cut0"].astype(np.int64)
    for t in range(50):
        r, c = oc.ppo_step(xs, z[f"{tag}/actions"][t], eu, ev, bidir, last)
        assert np.array_equal(r, z[f"{tag}/rewards"][t]) and np.array_equal(c, z[f"{tag}/curs"][t])
        r8 = oc.step_u8(xs8, z[f"{tag}/actions"][t], eu, ev, bidir, last8)
        assert np.array_equal(r8.astype(np.float32), z[f"{tag}/rewards"][t])
    assert np.array_equal((xs > 0).astype(np.uint8), z[f"{tag}/xs_final"])
    assert np.array_equal(xs8, z[f"{tag}/xs_final"])


def test_c_oracle_tsp(golden):
    from oracle import oracle_c as oc
    z = golden("tsp")
    for name in ("a5", "berlin52"):
        got = oc.tsp_tour_length(z[f"{name}/distance"], z[f"{name}/perms"])
        np.testing.assert_allclose(got, z[f"{name}/length_f32"], rtol=1e-5)


def test_fixtures_carry_provenance():
    """Every committed fixture names the command that regenerates it (tools/gen_golden.py --only <key>) and its
    seeding rule; the generator seeds the global RNGs per key, so any subset regenerates byte-identically."""
    import glob
    import os
    import re
    from tests.conftest import GOLDEN
    files = sorted(glob.glob(os.path.join(GOLDEN, "*.npz")))
    assert len(files) >= 14
    gen_src = open(os.path.join(os.path.dirname(GOLDEN), "..", "tools", "gen_golden.py")).read()
    keys = set(re.findall(r'"(\w+)": gen_\w+', gen_src))
    for f in files:
        z = np.load(f, allow_pickle=False)
        assert "__generator__" in z.files and "__seeding__" in z.files, f
        m = re.search(r"tools/gen_golden.py --only (\w+)$", str(z["__generator__"]))
        assert m and m.group(1) in keys, (f, str(z["__generator__"]))
        assert "seed" in str(z["__seeding__"])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_mcpg_weighted_sampler_oracle(golden, gname):
    """The weighted MaxCut sampler of the upstream MCPG package (methods/MCPG/sampling.py:89-127), bit for bit."""
    z = golden("mcpg_weighted")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    vs, xs_good, start, value, expected = onp.mcpg_sampling_maxcut(
        g, n, z[f"{gname}/sorted_degree_nodes"], z[f"{gname}/start"], z[f"{gname}/probs"], int(z[f"{gname}/num_ls"]),
        int(z[f"{gname}/change_times"]), int(z[f"{gname}/M"]), z[f"{gname}/metro_index"], z[f"{gname}/metro_u"], z[f"{gname}/uniforms"])
    assert np.array_equal(start.astype(np.uint8), z[f"{gname}/metro_out"])
    assert np.array_equal(vs, z[f"{gname}/vs"]) and np.array_equal(xs_good, z[f"{gname}/xs_good"])
    np.testing.assert_allclose(value, z[f"{gname}/value"], rtol=0, atol=1e-4)
    assert float(z[f"{gname}/edge_weight_sum"]) == float(g[:, 2].sum())


def test_tsp_2opt_local_search_oracle_vs_reference(golden):
    """The restatement of local_search_2_opt (methods_problem_specific/TSP/opt_2.py:27-57) reproduces what the reference
    returned: the same tours and bit-identical float64 distances (berlin52's until-no-improvement run is left to the GPU
    test: O(N^3) per pass in Python)."""
    z = golden("tsp_2opt")
    for name in z["names"]:
        d = z[f"{name}/distance_f64"]
        for t in range(2):
            tour = z[f"{name}/t{t}/start_tour"].tolist()
            sd = float(z[f"{name}/t{t}/start_distance"])
            assert onp.tsp_distance_calc(d, tour) == sd
            for rs in ((2,) if len(tour) > 40 else (-1, 2)):
                r, dist = onp.tsp_local_search_2_opt(d, tour, sd, rs)
                assert r == z[f"{name}/t{t}/rs{rs}/tour"].tolist() and dist == float(z[f"{name}/t{t}/rs{rs}/distance"])


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_mcpg_glue_merge_and_get_return(golden, gname):
    """mcpg_glue.npz: three rounds of the reference's metro_sampling -> sampler_func -> best-merge (MCPG.py:376-391,
    exec'd from the reference file) -> get_return (:292-302, value and autograd gradient).  The restatements reproduce
    every recorded array; merge and sampler outputs bit for bit."""
    z = golden("mcpg_glue")
    assert z["merge/reference_lines"].tolist() == [375, 391]     # "# update now_max" .. the temp_max_info re-seed
    graph = z[f"{gname}/graph"]
    n = int(graph[:, :2].max()) + 1
    ei = graph[:, :2].T.copy()
    M, R, num_ls, T = (int(z[f"{gname}/{k}"]) for k in ("M", "R", "num_ls", "T"))
    order = z[f"{gname}/sorted_degree_nodes"]
    probs = z[f"{gname}/probs"]
    for rnd in range(3):
        t = f"{gname}/round{rnd}"
        if rnd:   # the next round starts from the re-seeded kept chains, repeated (MCPG.py:393-394)
            assert np.array_equal(z[f"{t}/start"], np.tile(z[f"{gname}/round{rnd - 1}/temp_max_info_after"], (1, R)).astype(np.uint8))
        xs, _ = onp.metro_sampling(probs, z[f"{t}/start"], T, z[f"{t}/metro_index"], z[f"{t}/metro_u"])
        assert np.array_equal(xs.astype(np.uint8), z[f"{t}/xs_sample"])
        vs_good, xs_good, value, _, _ = onp.sampler_func(ei, n, order, z[f"{t}/xs_sample"], num_ls, M, R, z[f"{t}/uniforms"])
        assert np.array_equal(vs_good, z[f"{t}/temp_max"]) and np.array_equal(xs_good, z[f"{t}/temp_max_info"])
        np.testing.assert_allclose(value, z[f"{t}/value"], rtol=0, atol=1e-4)
        res, info, temp, now_max, idx = onp.mcpg_merge_best(z[f"{t}/temp_max"], z[f"{t}/temp_max_info"],
                                                            z[f"{t}/now_max_res_before"], z[f"{t}/now_max_info_before"])
        assert np.array_equal(res, z[f"{t}/now_max_res_after"]) and np.array_equal(info, z[f"{t}/now_max_info_after"])
        assert np.array_equal(temp, z[f"{t}/temp_max_info_after"])
        assert float(now_max) == float(z[f"{t}/now_max"]) and idx == int(z[f"{t}/now_max_index"])
        obj, grad = onp.mcpg_get_return(probs, z[f"{t}/xs_sample"].T.astype(np.float32), z[f"{t}/value"], M, R)
        np.testing.assert_allclose(obj, float(z[f"{t}/get_return"]), rtol=1e-5, atol=1e-5)    # the reference sums in f32; value has mean ~0
        np.testing.assert_allclose(grad, z[f"{t}/get_return_grad"], rtol=1e-4, atol=1e-4)


def _evaluator_stream(z, tag):
    n = int(z[f"{tag}/num_bits"])
    xs_all = np.unpackbits(z[f"{tag}/xs_packed"], axis=1)[:, :n].astype(bool)
    best_x = np.unpackbits(z[f"{tag}/best_x_packed"], axis=1)[:, :n].astype(bool)
    off = 0
    for k, cnt in enumerate(z[f"{tag}/counts"].tolist()):
        yield k + 1, xs_all[off:off + cnt], z[f"{tag}/vs"][off:off + cnt], bool(z[f"{tag}/single"][k]), \
            bool(z[f"{tag}/if_update"][k]), float(z[f"{tag}/best_v"][k]), best_x[k]
        off += cnt


@pytest.mark.parametrize("maximize", [True, False])
@pytest.mark.parametrize("vdt", ["int64", "float32"])
def test_evaluator_oracle(golden, maximize, vdt):
    z = golden("evaluator")
    tag = f"max{int(maximize)}/{vdt}"
    ev = onp.EvaluatorOracle(z[f"{tag}/x0"].astype(bool), float(z[f"{tag}/v0"]), maximize)
    for it, xs, vs, single, upd, best_v, best_x in _evaluator_stream(z, tag):
        ev.record1(it, float(vs.max()))
        got = ev.record2(it, vs[0] if single else vs, xs[0] if single else xs)
        assert bool(got) is upd and ev.best_v == best_v and np.array_equal(ev.best_x, best_x), (tag, it)
    assert np.array_equal(np.asarray(ev.recorder2, dtype=np.float64), z[f"{tag}/recorder2_i_v"])
    assert np.array_equal(np.asarray(ev.recorder1, dtype=np.float64), z[f"{tag}/recorder1"])
    assert ev.first_v == float(z[f"{tag}/first_v"])


@pytest.mark.parametrize("k,rows,cols", [(49, 30, 100), (50, 25, 120)])
def test_oracle_on_toroidal_gset_known_answer(golden, k, rows, cols):
    """SURVEY 8c item 10 without a data file: Gset's G49 / G50 are 4-regular toroidal grids, and the reference's X_G49 / X_G50
    strings (util_evaluator.py:272-281) flip phase every 100 / 120 nodes -- on the 30 x 100 / 25 x 120 torus with row-major
    numbering they cut exactly the claimed 6000 / 5880 edges.  The oracle's objective on the decoded string is that number."""
    from oracle import oracle_np as onp
    z = golden("encoder_base64")
    x = z[f"kat/G{k}/x"].astype(np.uint8)
    idx = np.arange(rows * cols).reshape(rows, cols)
    e = np.concatenate([np.stack([idx, np.roll(idx, -1, axis=1)], axis=-1).reshape(-1, 2),
                        np.stack([idx, np.roll(idx, -1, axis=0)], axis=-1).reshape(-1, 2)])
    graph = np.concatenate([np.sort(e, axis=1), np.ones((e.shape[0], 1), dtype=np.int64)], axis=1)
    for bidir in (False, True):
        got = onp.maxcut_obj(np.stack([x, 1 - x]), graph, bidir)
        assert got.tolist() == [int(z[f"kat/G{k}/claimed_cut"])] * 2
