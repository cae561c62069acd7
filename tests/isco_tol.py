"""Tolerances of the ISCO sampler-step comparisons (oracle vs reference trace, HIP kernel vs reference trace).

The path log-probabilities ll_x2y / ll_y2x come from noreplacement_sampling_renormalize (rlsolver/methods/util.py:
507-512): sum_k [ll_k - log(1 - exp(log(cumsum_k - p_k) + base))].  The inner 1 - exp(.) is the probability mass NOT
yet drawn; computed in f32 its relative error is ~eps / (remaining mass), so the terms of late draws are
ill-conditioned IN THE REFERENCE ITSELF: any other order of the same f32 operations (numpy vs torch on the CPU, wave
reductions on the GPU) moves them by that much.  Measured numpy-vs-torch on the committed traces: error ~ 1e-7 /
remaining mass; 7 % of the value when all nodes are drawn (path_length = N: the last term is the log of rounding noise).

The cumulative sum itself carries the rounding of its L terms (a sequential f32 cumsum in torch / numpy, a wave scan
on the GPU: ~sqrt(L) eps each, over L terms), which the same 1 / (remaining mass) amplifies: measured 0.06 absolute on
ll_x2y = -5900 (L = 1000 draws of N = 2000, remaining mass 0.03) between numpy and the kernel.

So the comparison is exact for everything discrete (selected mask, proposal, walked tour, accepted sample unless the
accept test itself sits inside the tolerance) and, for the log-probabilities, absolute tolerance
    2e-5 + (2e-6 + 1e-7 L^1.5) / remaining_mass      (+ 1e-5 relative)
with L = path_length and remaining_mass from the oracle in float64; envs below 1e-6 of remaining mass are compared
on the discrete outputs only."""
import numpy as np

RTOL = 1e-5
MIN_MASS = 1e-6


def ll_atol(remaining_mass, path_length=1):
    L = np.asarray(path_length, dtype=np.float64)
    return 2e-5 + (2e-6 + 1e-7 * L ** 1.5) / np.maximum(np.asarray(remaining_mass, dtype=np.float64), MIN_MASS)


def assert_ll_close(actual, desired, remaining_mass, what="", path_length=1):
    actual, desired = np.asarray(actual, np.float64), np.asarray(desired, np.float64)
    ok = np.asarray(remaining_mass) >= MIN_MASS
    err = np.abs(actual - desired)
    tol = ll_atol(remaining_mass, path_length) + RTOL * np.abs(desired)
    bad = ok & ~(err <= tol)
    assert not bad.any(), f"{what}: envs {np.flatnonzero(bad).tolist()} err {err[bad]} tol {tol[bad]} mass {np.asarray(remaining_mass)[bad]}"
    return int(ok.sum())
