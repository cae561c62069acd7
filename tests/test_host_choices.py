"""Host-side kernel choices that need no GPU."""
from rlsolver_amd.methods.MCPG_qubo import qubo_prefers_sparse


def test_qubo_kernel_choice_follows_the_measured_crossovers():
    # (n, fill, chains) -> CSR kernel wins; measured on MI355X (tools/sweeps/time_qubo_sparse.py, round 2)
    measured = [
        (1000, 0.02, 1 << 13, False), (1000, 0.02, 1 << 16, True), (1000, 0.1, 1 << 16, False),
        (500, 0.005, 1 << 13, False), (500, 0.005, 1 << 16, True), (2000, 0.005, 1 << 13, False),
        (2000, 0.02, 1 << 16, True), (2000, 0.1, 1 << 16, False), (2000, 0.25, 1 << 16, False),
    ]
    for n, fill, C, want in measured:
        assert qubo_prefers_sparse(n, int(n * n * fill), C) == want, (n, fill, C)
    assert not qubo_prefers_sparse(64, 64 * 64, 1 << 20)      # a full matrix never goes sparse
    assert not qubo_prefers_sparse(0, 0, 10)
