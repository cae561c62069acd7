"""Time of one exact best-improvement 2-opt pass (rls_tsp_2opt_best) for a lone tour, by the number of workgroups sharing it."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from rlsolver_amd import ops_mcpg_tsp as mops
dev = torch.device("cuda:0")
for N in (52, 200, 500, 1000):
    rng = np.random.RandomState(N)
    c = rng.rand(N, 2)
    d = torch.from_numpy(np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1))).to(dev)
    perm = torch.from_numpy(rng.permutation(N)[None]).to(dev)
    cur = d[perm, torch.roll(perm, -1, 1)].sum(1)
    for exact in (True, False):
        for sl in (1, 8, 64, None):
            f = lambda: mops.tsp_2opt_best(d, perm, cur if exact else None, slices=sl)
            f(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): f()
            torch.cuda.synchronize()
            print(f"N={N} exact={exact} slices={sl}: {(time.perf_counter() - t0) / 5 * 1e6:.0f} us")
