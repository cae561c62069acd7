"""Differential fuzz of the QUBO coordinate search (K11 on the matrix cores, and its CSR form) against the numpy restatement of
the reference's variable-by-variable loop (MCPG/sampling.py:332-337, :357-362): random sizes on both sides of every block
boundary, integer matrices (every sum exact in float32), with and without a diagonal, dense and sparse, 0-3 sweeps, chain
counts around the 32 / 64-chain tiles and the wave split.  `python tools/fuzz/fuzz_qubo.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd.methods import MCPG_qubo as q

DEV = torch.device("cuda:0")
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    n = int(rng.choice([rng.randint(1, 40), rng.randint(30, 70), rng.randint(60, 300), rng.randint(300, 700)]))
    C = int(rng.choice([1, 5, 31, 32, 33, 63, 64, 65, 130, 300]))
    density = float(rng.choice([1.0, 0.8, 0.2, 0.02]))
    Qn = (rng.randint(-40, 41, size=(n, n)) * (rng.rand(n, n) < density)).astype(np.float32)
    Qn = np.triu(Qn) + np.triu(Qn, 1).T
    if rng.rand() < 0.3:
        np.fill_diagonal(Qn, 0)
    num_ls = int(rng.randint(0, 4))
    binary = bool(rng.rand() < 0.5)
    x0 = rng.randint(0, 2, size=(n, C)).astype(np.float32)
    tag = f"it={it} n={n} C={C} density={density} num_ls={num_ls} binary={binary}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    s = x0.copy() if binary else 2 * x0 - 1
    for cnt in range(num_ls):
        for i in range(n):
            s[i] = 0
            res = Qn[i] @ s
            s[i] = ((res > -Qn[i, i] / 2).astype(np.float32)) if binary else (2 * (res > 0) - 1).astype(np.float32)
    want_x = s if binary else (s + 1) / 2
    want_v = np.einsum("ic,ij,jc->c", s.astype(np.float64), Qn.astype(np.float64), s.astype(np.float64)).astype(np.float32)
    Q = dev(Qn)
    xd, vd = q.qubo_local_search_value(Q, dev(x0), num_ls, binary)
    assert np.array_equal(xd.cpu().numpy(), want_x), "dense x " + tag
    assert np.array_equal(vd.cpu().numpy(), want_v), "dense value " + tag
    if (Qn != 0).any():
        csr = q.qubo_to_csr(Q)
        xs_, vs_ = q.qubo_sparse_local_search_value(csr, dev(x0), num_ls, binary)
        assert torch.equal(xd, xs_) and torch.equal(vd, vs_), "sparse (levels) " + tag
        xs_, vs_ = q.qubo_sparse_local_search_value(csr[:3], dev(x0), num_ls, binary)
        assert torch.equal(xd, xs_) and torch.equal(vd, vs_), "sparse (sequential) " + tag
    it += 1
print(f"fuzz_qubo: {it} random configurations, no mismatch")
