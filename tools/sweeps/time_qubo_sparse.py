"""Dev: sparse vs dense K11 over fill and chain count (calibrates the sampler's choice in MCPG_qubo.py)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd.methods import MCPG_qubo as q
dev = torch.device("cuda:0")


def t_us(fn, it=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


rng = np.random.RandomState(0)
for n in (1000, 2000):
    base = rng.randint(10, 101, size=(n, n)) * rng.choice([-1, 1], size=(n, n))
    for fill in (0.005, 0.02, 0.1):
        Qn = base * (rng.rand(n, n) < fill)
        Qn = (np.triu(Qn) + np.triu(Qn, 1).T).astype(np.float32)
        Q = torch.from_numpy(Qn).to(dev)
        csr = q.qubo_to_csr(Q)
        for C in (1 << 13, 1 << 15):
            x0 = (torch.rand(n, C, device=dev) < 0.5).float()
            td = t_us(lambda: q.qubo_local_search_value(Q, x0, 2, False))
            ts = t_us(lambda: q.qubo_sparse_local_search_value(csr, x0, 2, False))
            tq = t_us(lambda: q.qubo_sparse_local_search_value(csr[:3], x0, 2, False))
            print("n=%d fill=%.3f (deg %.1f, %d levels) C=%d: dense %.0f us  sparse by levels %.0f us  sparse sequential %.0f us" %
                  (n, fill, int(csr[0][-1]) / n, csr[3].numel() - 1, C, td, ts, tq), flush=True)
