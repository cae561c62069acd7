"""Dev timing of K11 (dense QUBO local search + value) by number of sweeps and chains."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd.methods import MCPG_qubo as q
dev = torch.device("cuda:0")
n = 1000
rng = np.random.RandomState(0)
Qn = rng.randint(10, 101, size=(n, n)) * rng.choice([-1, 1], size=(n, n)) * (rng.rand(n, n) < 0.8)
Qn = (np.triu(Qn) + np.triu(Qn, 1).T).astype(np.float32)
Q = torch.from_numpy(Qn).to(dev)
for C in (1 << 13, 1 << 15, 1 << 16):
    x0 = (torch.rand(n, C, device=dev) < 0.5).float()
    for ls in (0, 1, 2):
        for _ in range(2):
            q.qubo_local_search_value(Q, x0, ls, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(5):
            q.qubo_local_search_value(Q, x0, ls, False)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1e3
        fl = 2.0 * n * n * C * (ls + 1)
        print("C=%d num_ls=%d: %.1f us  %.1f TFLOP/s" % (C, ls, us, fl / us / 1e6))
# parity of the 64-chain-tile variant against the 32-chain one (integer Q: exact)
x0 = (torch.rand(n, 1 << 15, device=dev) < 0.5).float()
xa, va = q.qubo_local_search_value(Q, x0, 2, False)
xb, vb = q.qubo_local_search_value(Q, x0[:, :4096].contiguous(), 2, False)
print("NT=2 vs NT=1 equal:", bool(torch.equal(xa[:, :4096], xb)), bool(torch.equal(va[:4096], vb)))
xa, va = q.qubo_local_search_value(Q, x0, 2, True)
xb, vb = q.qubo_local_search_value(Q, x0[:, :4096].contiguous(), 2, True)
print("NT=2 vs NT=1 equal (0/1):", bool(torch.equal(xa[:, :4096], xb)), bool(torch.equal(va[:4096], vb)))
