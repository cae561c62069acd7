"""Pin the spin-system oracle (dense restatement) against the reference's golden trace (CPU)."""
import numpy as np
import pytest

from oracle.oracle_spin import SpinSystemOracle, SpinSystemOracleF64

CFG = {"eco": dict(reward="BLS", norm_rewards=True), "dense": dict(reward="DENSE", norm_rewards=False),
       "stag": dict(reward="CUSTOM_BLS", norm_rewards=False, basin_reward=0.25, stag_punishment=0.125)}


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
@pytest.mark.parametrize("cname", ["eco", "dense", "stag"])
def test_spin_oracle_golden(golden, gname, cname):
    z = golden("spinsystem")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    W = np.zeros((n, n), np.float32)
    W[g[:, 0], g[:, 1]] = g[:, 2]
    W[g[:, 1], g[:, 0]] = g[:, 2]
    tag = f"{gname}/{cname}"
    T = int(z[f"{tag}/max_steps"])
    cfg = dict(CFG[cname])
    if cname == "eco":
        cfg["basin_reward"] = 1.0 / n
    env = SpinSystemOracle(W, 6, T, **cfg)
    assert np.array_equal(env.max_local, z[f"{tag}/max_local"])
    assert np.array_equal(env.reset(z[f"{tag}/spins0"]), z[f"{tag}/obs0"])
    assert np.array_equal(env.score, z[f"{tag}/score0"])
    for t in range(T):
        o, r, d = env.step(z[f"{tag}/actions"][t])
        assert np.array_equal(o, z[f"{tag}/obs"][t]), t
        assert np.array_equal(r, z[f"{tag}/rew"][t]), t
        assert np.array_equal(d, z[f"{tag}/done"][t])
        assert np.array_equal(env.score, z[f"{tag}/score"][t])
        assert np.array_equal(env.best_score, z[f"{tag}/best_score"][t])
    assert np.array_equal(env.best_spins, z[f"{tag}/best_spins"])
    assert np.array_equal(W, z[f"{tag}/adj_rows"])


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
@pytest.mark.parametrize("cname", ["eco", "dense", "stag"])
def test_spin_oracle_f64_single_env_golden(golden, gname, cname):
    """the numpy env (ECO_S2V/src/envs/spinsystem.py) in float64, bit for bit."""
    z = golden("spinsystem_cpu")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    W = np.zeros((n, n))
    W[g[:, 0], g[:, 1]] = g[:, 2]
    W[g[:, 1], g[:, 0]] = g[:, 2]
    tag = f"{gname}/{cname}"
    T = int(z[f"{tag}/max_steps"])
    cfg = dict(CFG[cname])
    if cname == "eco":
        cfg["basin_reward"] = 1.0 / n
    env = SpinSystemOracleF64(W, T, **cfg)
    assert env.max_local == float(z[f"{tag}/max_local"])
    obs = env.reset(z[f"{tag}/spins0"])
    assert obs.dtype == np.float64 and np.array_equal(obs, z[f"{tag}/obs0"])
    assert env.score == float(z[f"{tag}/score0"])
    for t in range(T):
        o, r, d = env.step(int(z[f"{tag}/actions"][t]))
        assert np.array_equal(o[:7], z[f"{tag}/obs"][t]), t
        assert np.array_equal(env.gains(), z[f"{tag}/gains"][t]), t
        assert r == z[f"{tag}/rew"][t], t
        assert d == bool(z[f"{tag}/done"][t])
        assert env.score == z[f"{tag}/score"][t] and env.best_score == z[f"{tag}/best_score"][t]
    assert np.array_equal(env.best_spins, z[f"{tag}/best_spins"])
    assert np.array_equal(o[7:], z[f"{tag}/adj_rows"])
    assert (z[f"{tag}/rew"] != 0).any()


@pytest.mark.parametrize("cname", ["pass", "mem3", "pass_mem4_stag"])
def test_f64_oracle_options_golden(golden, cname):
    """spinsystem_options.npz: ExtraAction.PASS and a finite memory_length on the reference's numpy env (spinsystem.py:349-351,
    :398-404), alone and together with the visited-state rewards: the restatement reproduces the padded state, the
    observation, rewards, scores and the best observable score bit for bit."""
    z = golden("spinsystem_options")
    g = z["graph"]
    n = int(g[:, :2].max()) + 1
    W = np.zeros((n, n))
    for a, b, w in g:
        W[a, b] = W[b, a] = w
    cfg = {"pass": dict(reward="BLS", norm_rewards=True, basin_reward=1.0 / n, extra_pass=True),
           "mem3": dict(reward="BLS", memory_length=3),
           "pass_mem4_stag": dict(reward="CUSTOM_BLS", basin_reward=0.25, stag_punishment=0.125, extra_pass=True, memory_length=4)}[cname]
    T = int(z["max_steps"])
    env = SpinSystemOracleF64(W, T, **cfg)
    assert env.na == int(z[f"{cname}/n_actions"])
    obs = env.reset(z[f"{cname}/spins0"])
    assert np.array_equal(obs, z[f"{cname}/obs0"]) and np.array_equal(env.state, z[f"{cname}/state0"])
    for t in range(T):
        o, r, d = env.step(int(z[f"{cname}/actions"][t]))
        assert np.array_equal(o[:7], z[f"{cname}/obs"][t]), t
        assert np.array_equal(env.state, z[f"{cname}/state"][t]), t
        assert r == z[f"{cname}/rew"][t] and d == bool(z[f"{cname}/done"][t]), t
        assert env.score == z[f"{cname}/score"][t] and env.best_score == z[f"{cname}/best_score"][t]
        assert env.best_obs_score == z[f"{cname}/best_obs_score"][t], t
    assert np.array_equal(o[7:], z[f"{cname}/adj_rows"]) and np.array_equal(env.best_spins, z[f"{cname}/best_spins"])


def test_reference_facts_about_unrunnable_options(golden):
    """What the reference does with the options the build refuses: the batched env cannot be constructed with an extra
    action or a finite memory, RANDOMISE raises on first use in the numpy env (exception type names recorded by gen_golden)."""
    z = golden("spinsystem_options")
    assert str(z["facts/batched_none_constructs"]) == "ok"
    assert str(z["facts/batched_pass_ctor"]) == "RuntimeError" and str(z["facts/batched_randomise_ctor"]) == "RuntimeError"
    assert str(z["facts/batched_memory3_ctor"]) == "TypeError" and str(z["facts/numpy_randomise_first_use"]) == "ValueError"


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
def test_spin_oracle_inference_twin_golden(golden, gname):
    """spinsystem_inference.npz (inference_network_env.py as inference_PECO.py builds it): the oracle's inference flavour --
    best score / spins seeded from the best env of the batch, step() -> (obs, done) -- bit for bit."""
    z = golden("spinsystem_inference")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    W = np.zeros((n, n), np.float32)
    W[g[:, 0], g[:, 1]] = g[:, 2]
    W[g[:, 1], g[:, 0]] = g[:, 2]
    T, B = int(z[f"{gname}/max_steps"]), z[f"{gname}/spins0"].shape[0]
    env = SpinSystemOracle(W, B, T, reward="BLS", norm_rewards=True, basin_reward=1.0 / n, inference=True)
    assert np.array_equal(env.reset(z[f"{gname}/spins0"]), z[f"{gname}/obs0"])
    assert np.array_equal(env.score, z[f"{gname}/score0"])
    assert z[f"{gname}/best_cut0"].ndim == 0 and np.all(env.best_score == z[f"{gname}/best_cut0"])
    assert np.array_equal(env.best_spins, z[f"{gname}/best_spins0"])
    for t in range(T):
        o, d = env.step(z[f"{gname}/actions"][t])
        assert np.array_equal(o, z[f"{gname}/obs"][t]), t
        assert np.array_equal(d, z[f"{gname}/done"][t])
        assert np.array_equal(env.score, z[f"{gname}/score"][t])
        assert np.array_equal(env.best_score, z[f"{gname}/best_score"][t])
    assert np.array_equal(env.best_spins, z[f"{gname}/best_spins"])
    assert np.array_equal(W, z[f"{gname}/adj_rows"])


S2V_CASES = {
    "s2v": dict(reward="DENSE", norm_rewards=True, reversible=False, s2v=True, binary=True),
    "eco_irreversible": dict(reward="BLS", norm_rewards=True, basin_reward=1.0 / 20, reversible=False, binary=False),
    "defaults": dict(reward="DENSE", extra_pass=True, target="ENERGY", binary=False),
    "energy_bls_mem": dict(reward="BLS", norm_rewards=True, basin_reward=1.0 / 20, extra_pass=True, memory_length=3, target="ENERGY", binary=True),
    "energy_custom_stag": dict(reward="CUSTOM_BLS", basin_reward=0.25, stag_punishment=0.125, target="ENERGY", binary=False),
    "energy_irreversible": dict(reward="DENSE", norm_rewards=True, reversible=False, target="ENERGY", binary=True),
    "energy_isolated": dict(reward="BLS", target="ENERGY", binary=False),
}


@pytest.mark.parametrize("cname", sorted(S2V_CASES))
def test_f64_oracle_s2v_energy_golden(golden, cname):
    """spinsystem_s2v.npz: irreversible spins (S2V-DQN's config, train_S2V.py:37-47) and OptimisationTarget.ENERGY (the default of
    SpinSystemFactory.get) on the reference's numpy env: the restatement reproduces state, observation, rewards, done, scores,
    the maximum local reward (over the NONZERO entries: negative on the positive graph with an isolated node) bit for bit."""
    z = golden("spinsystem_s2v")
    g = z["graph_isolated"] if cname == "energy_isolated" else z["graph"]
    n = 12 if cname == "energy_isolated" else int(g[:, :2].max()) + 1
    W = np.zeros((n, n))
    for a, b, w in g:
        W[a, b] = W[b, a] = w
    cfg = S2V_CASES[cname]
    T = int(z[f"{cname}/max_steps"])
    env = SpinSystemOracleF64(W, T, **cfg)
    R = 1 if cfg.get("s2v") else 7
    assert env.na == int(z[f"{cname}/n_actions"]) and env.max_local == float(z[f"{cname}/max_local"])
    obs = env.reset(None if not cfg.get("reversible", True) else z[f"{cname}/spins0"])
    assert np.array_equal(env.state[0, :n], z[f"{cname}/spins0"])
    assert np.array_equal(obs, z[f"{cname}/obs0"]) and np.array_equal(env.state[:R], z[f"{cname}/state0"])
    assert env.score == float(z[f"{cname}/score0"]) and np.array_equal(env.gains(), z[f"{cname}/imm0"])
    acts = z[f"{cname}/actions"]
    for t, a in enumerate(acts):
        o, r, d = env.step(int(a))
        assert np.array_equal(o[:R], z[f"{cname}/obs"][t]), t
        assert np.array_equal(env.state[:R], z[f"{cname}/state"][t]), t
        assert r == z[f"{cname}/rew"][t] and d == bool(z[f"{cname}/done"][t]), t
        assert env.score == z[f"{cname}/score"][t] and env.best_score == z[f"{cname}/best_score"][t]
        assert env.best_obs_score == z[f"{cname}/best_obs_score"][t], t
    assert d and (len(acts) < T) == (cname in ("s2v", "eco_irreversible", "energy_irreversible"))   # done early: no spin left at +1
    assert np.array_equal(o[R:], z[f"{cname}/adj_rows"]) and np.array_equal(env.best_spins, z[f"{cname}/best_spins"])
    assert np.array_equal(env.gains(), z[f"{cname}/imm_end"])


def test_reference_facts_about_batched_s2v_options(golden):
    """The BATCHED reference env with these options: ENERGY cannot be constructed (AttributeError in its constructor); irreversible
    spins construct and step, but the reset line `state[0, :n_spins] = 1` (spinsystem_PECO.py:221) fills every ROW of env 0 and
    leaves the other envs' spins at 0 -- not a behaviour to reproduce: the batched factory here refuses both."""
    z = golden("spinsystem_s2v")
    assert str(z["facts/batched_energy_ctor"]) == "AttributeError" and str(z["facts/energy_get_best_cut"]) == "NotImplementedError"
    assert str(z["facts/batched_irreversible_ctor"]) == "ok" and str(z["facts/batched_irreversible_step"]) == "ok"
