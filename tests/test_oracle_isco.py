"""Pin the ISCO sampler-step oracle (numpy restatement of envs/env_ISCO.py + methods/util.py:498-570) against the
traces captured from the imported reference (tests/golden/isco_steps.npz).  CPU only.

Integer / boolean results (selected mask, proposal, walked tour, accepted sample) must match exactly.  The f32 path
log-probabilities are compared with the conditioning-aware tolerance of tests/isco_tol.py: the reference's own
arithmetic takes log(1 - exp(.)) of a cumulative sum that approaches 1, so an ulp of difference in exp / log /
summation order is amplified by 1 / (probability mass not yet drawn)."""
import numpy as np
import pytest

from oracle import oracle_isco as oi
from tests.isco_tol import RTOL, assert_ll_close


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_isco_maxcut_step_oracle_golden(golden, gname):
    z = golden("isco_steps")
    g = z[f"maxcut/{gname}/graph"]
    eu, ev = g[:, 0], g[:, 1]
    for k in range(3):
        t = f"maxcut/{gname}/step{k}"
        r = oi.maxcut_step(z[f"{t}/x"], eu, ev, z[f"{t}/path_length"], float(z[f"{t}/temperature"]),
                           z[f"{t}/rand_gumbel"], z[f"{t}/rand_accept"])
        assert np.array_equal(r["mask"], z[f"{t}/mask"]), k
        assert int(r["mask"][0].sum()) == 1 and int(r["mask"][1].sum()) == g[:, :2].max() + 1
        assert np.array_equal(r["y_prop"], z[f"{t}/y_prop"])
        for key in ("ll_x", "ll_y", "energy"):           # no renormalisation involved: tight
            np.testing.assert_allclose(r[key], z[f"{t}/{key}"], rtol=RTOL, atol=1e-5, err_msg=f"{t}/{key}")
        checked = [assert_ll_close(r[key], z[f"{t}/{key}"], r["remaining_mass"], f"{t}/{key}", z[f"{t}/path_length"])
                   for key in ("ll_x2y", "ll_y2x", "log_acc")]
        assert min(checked) >= 10                        # all but the path_length = N env carry information
        sure = r["accept_margin"] > 2 * (2e-5 + 2e-6 / np.maximum(r["remaining_mass"], 1e-6))
        assert sure.sum() >= 9 and np.array_equal(r["y"][sure], z[f"{t}/y"][sure])


@pytest.mark.parametrize("name", ["a5", "berlin52"])
def test_isco_tsp_step_oracle_golden(golden, name):
    z = golden("isco_steps")
    p = f"tsp/{name}"
    for k in range(2):
        t = f"{p}/step{k}"
        r = oi.tsp_step(z[f"{t}/x"], z[f"{p}/distance"], z[f"{p}/nearest_indices"], z[f"{p}/random_indices"], int(z[f"{p}/K"]),
                        int(z[f"{t}/path_length"]), float(z[f"{t}/temperature"]), z[f"{t}/rand_partner"],
                        z[f"{t}/randint_nearest"], z[f"{t}/randint_random"], z[f"{t}/rand_gumbel"], z[f"{t}/rand_accept"])
        assert np.array_equal(r["cur_x"], z[f"{t}/cur_x"]), k
        np.testing.assert_allclose(r["log_acc"], z[f"{t}/log_acc"], rtol=RTOL, atol=2e-5)
        assert np.array_equal(r["y"], z[f"{t}/y"])
        np.testing.assert_allclose(r["mean_acc"], z[f"{t}/mean_acc"], rtol=RTOL, atol=1e-7)
        assert all(sorted(row) == list(range(r["y"].shape[1])) for row in r["y"].tolist())
