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
