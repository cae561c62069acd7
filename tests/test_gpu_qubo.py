"""QUBO samplers vs golden vectors captured from rlsolver/methods/MCPG/sampling.py."""
import numpy as np
import pytest
import torch

from rlsolver_amd.methods import MCPG_qubo as q
from tests.gpu_util import DEV

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t.to(dtype) if dtype is not None else t


@pytest.mark.parametrize("name", ["nbiq_5", "rand_24"])
@pytest.mark.parametrize("mode", ["pm1", "bin"])
def test_qubo_golden(golden, name, mode):
    z = golden("qubo")
    data = {"Q": dev(z[f"{name}/Q"]), "nvar": z[f"{name}/Q"].shape[0]}
    fn = q.mcpg_sampling_qubo if mode == "pm1" else q.mcpg_sampling_qubo_bin
    max_res, best, raw, value = fn(data, dev(z[f"{name}/start"], torch.float32), dev(z[f"{name}/probs"]),
                                   int(z[f"{name}/num_ls"]), int(z[f"{name}/change_times"]), int(z[f"{name}/M"]),
                                   DEV, index=dev(z[f"{name}/{mode}/index"]), u=dev(z[f"{name}/{mode}/u"]))
    assert np.array_equal(raw.cpu().numpy().astype(np.uint8), z[f"{name}/{mode}/raw"])
    assert np.array_equal(max_res.cpu().numpy(), z[f"{name}/{mode}/max_res"])
    assert np.array_equal(best.cpu().numpy(), z[f"{name}/{mode}/best"])
    np.testing.assert_allclose(value.cpu().numpy(), z[f"{name}/{mode}/value"], rtol=1e-6, atol=1e-3)


def test_qubo_value_is_quadratic_form_and_sweep_is_monotone():
    rng = np.random.RandomState(3)
    n, C = 150, 130
    Qn = rng.randint(-50, 51, size=(n, n)).astype(np.float32)
    Qn = Qn + Qn.T
    x0 = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    Q = dev(Qn)
    for binary in (False, True):
        x1, v1 = q.qubo_local_search_value(Q, dev(x0), 0, binary)       # no sweeps: pure value
        s = x0 if binary else 2 * x0 - 1
        assert np.array_equal(x1.cpu().numpy(), x0)
        assert np.array_equal(v1.cpu().numpy(), np.einsum("ic,ij,jc->c", s, Qn, s).astype(np.float32))
        x2, v2 = q.qubo_local_search_value(Q, dev(x0), 3, binary)
        s2 = x2.cpu().numpy() if binary else 2 * x2.cpu().numpy() - 1
        assert np.array_equal(v2.cpu().numpy(), np.einsum("ic,ij,jc->c", s2, Qn, s2).astype(np.float32))
        assert (v2 >= v1).all()                                          # symmetric Q: coordinate ascent never loses
