"""ISCO_maxcut surface vs golden vectors from the reference's autograd formulation."""
import numpy as np
import pytest
import torch

from tests.gpu_util import DEV

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("gname", ["BA_100_ID0", "PL_20_ID0"])
def test_local_dist_golden(golden, gname):
    from rlsolver_amd.envs.env_ISCO_maxcut import ISCO_maxcut
    z = golden("isco_maxcut")
    g = z[f"{gname}/graph"]
    n = int(g[:, :2].max()) + 1
    params = {"num_nodes": n, "num_edges": len(g), "edge_from": torch.from_numpy(g[:, 0].copy()).to(DEV),
              "edge_to": torch.from_numpy(g[:, 1].copy()).to(DEV)}
    s = ISCO_maxcut(params, batch_size=12, device=DEV)
    assert str(z[f"{gname}/x_dtype"]) == "torch.float32"
    x = torch.from_numpy(z[f"{gname}/x"]).to(DEV).float()
    for T in (1.0, 0.37):
        energy, logp = s.get_local_dist(x, torch.tensor(T))
        assert energy.dtype == torch.float32 and logp.shape == (12, n)
        np.testing.assert_allclose(energy.cpu().numpy(), z[f"{gname}/T{T}/energy"], rtol=1e-6)
        np.testing.assert_allclose(logp.cpu().numpy(), z[f"{gname}/T{T}/log_prob"], rtol=1e-5, atol=1e-5)
    # a few sampler steps run and keep samples binary
    torch.manual_seed(0)
    x = s.random_gen_init_sample()
    e0 = s.model(x, 1.0)
    for it in range(20):
        pl = torch.full((12,), 3, dtype=torch.int64, device=DEV)
        x, en, acc = s.step(x, pl, torch.tensor(0.5, device=DEV))
    assert set(np.unique(x.cpu().numpy())) <= {0.0, 1.0}
    assert float(s.model(x, 1.0).mean()) >= float(e0.mean())    # annealing towards larger cuts
