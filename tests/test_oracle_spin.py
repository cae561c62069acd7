"""Pin the spin-system oracle (dense restatement) against the reference's golden trace (CPU)."""
import numpy as np
import pytest

from oracle.oracle_spin import SpinSystemOracle


@pytest.mark.parametrize("gname", ["PL_20_ID0", "BA_100_ID0"])
@pytest.mark.parametrize("cname", ["eco", "dense"])
def test_spin_oracle_golden(golden, gname, cname):
    z = golden("spinsystem")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    W = np.zeros((n, n), np.float32)
    W[g[:, 0], g[:, 1]] = g[:, 2]
    W[g[:, 1], g[:, 0]] = g[:, 2]
    tag = f"{gname}/{cname}"
    T = int(z[f"{tag}/max_steps"])
    env = SpinSystemOracle(W, 6, T, reward="BLS" if cname == "eco" else "DENSE", norm_rewards=cname == "eco",
                           basin_reward=(1.0 / n) if cname == "eco" else None)
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
