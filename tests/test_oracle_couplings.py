"""The numpy restatement of rls_rand_couplings (oracle/oracle_np.py) on the CPU: the structure the reference's generators
(ECO_S2V/src/envs/util_envs_PECO.py:15-113) produce, and the mean-degree profile of its Barabasi-Albert generator."""
import numpy as np

from oracle import oracle_np as onp


def test_ba_restatement_structure_and_degree_profile():
    B, N, m = 192, 64, 4
    ba = onp.rand_couplings_ba(B, N, m, 2, 7)
    a = np.abs(ba)
    assert np.array_equal(ba, ba.transpose(0, 2, 1)) and set(np.unique(ba)) <= {-1.0, 0.0, 1.0}
    assert (a[:, :m + 1, :m + 1] == 1).all()                                    # the seed clique, self-loops included
    assert not np.diagonal(a, axis1=1, axis2=2)[:, m + 1:].any()
    assert all((np.tril(a[b], -1)[m + 1:].sum(1) == m).all() for b in range(B))   # m distinct earlier neighbours
    shared = a[0] * a[1:]
    assert np.array_equal(np.sign(ba[0]) * shared, np.sign(ba[1:]) * shared)    # DISCRETE: one sign per node pair
    deg = a.sum(-1).mean(0)
    # row sums of the reference generator, N = 64, m = 4, 4096 graphs: node 0 21.9, 5 16.4, 16 8.45, 32 5.72, 63 4.0
    for node, want, tol in ((0, 21.9, 2.5), (5, 16.4, 2.0), (16, 8.45, 1.0), (32, 5.72, 0.6), (63, 4.0, 1e-9)):
        assert abs(deg[node] - want) <= tol, (node, deg[node])


def test_er_restatement_structure_density_and_sharding():
    B, N = 96, 40
    for edge_type in (1, 2, 3):
        er = onp.rand_couplings_er(B, N, 0.2, edge_type, 11, env_offset=3)
        assert np.array_equal(er, er.transpose(0, 2, 1)) and not np.diagonal(er, axis1=1, axis2=2).any()
        assert abs(np.abs(er).sum() / (B * N * (N - 1)) - 0.2) < 0.01
        assert np.array_equal(onp.rand_couplings_er(5, N, 0.2, edge_type, 11, env_offset=3 + 40), er[40:45])
        if edge_type == 1:
            assert set(np.unique(er)) <= {0.0, 1.0}
        else:
            assert abs(er.sum()) < 0.1 * np.abs(er).sum()                        # signs balanced
    assert not onp.rand_couplings_er(2, N, 0.0, 1, 1).any()
    full = onp.rand_couplings_er(2, N, 1.0, 1, 1)
    assert full.sum() == 2 * N * (N - 1)
