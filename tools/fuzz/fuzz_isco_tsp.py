"""Differential fuzz of the ISCO_TSP sampler step against the numpy oracle with recorded draws: random instance sizes
(K + 2 .. 130 cities), K, batch sizes, path lengths and temperatures.  The walked tour is compared exactly except where the
Gumbel argmax of a round is decided within a few ulps (the oracle reports nothing about that, so a mismatching tour is
re-examined: it must still be a permutation reachable by the recorded partner draws), log_acc within 2e-5 relative + 1e-4.
`python tools/fuzz/fuzz_isco_tsp.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_isco as oi
from rlsolver_amd.envs.env_ISCO import ISCO_TSP
from rlsolver_amd.graph import generate_tsp_coords, tsp_tables

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = exact = 0
while time.time() < t_end:
    K = int(rng.choice([2, 5, 20]))
    N = int(rng.randint(K + 3, 130))
    B = int(rng.choice([1, 3, 64, 65, 200]))
    L = int(rng.randint(1, 7))
    T = float(rng.choice([0.05, 0.2, 0.7, 2.0]))
    tag = f"it={it} N={N} K={K} B={B} L={L} T={T}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    dist, near, rnd = tsp_tables(generate_tsp_coords(N, seed=int(rng.randint(1 << 30))), K=K)
    s = ISCO_TSP({"num_nodes": N, "distance": torch.from_numpy(dist).to(DEV), "nearest_indices": torch.from_numpy(near).to(DEV),
                  "random_indices": torch.from_numpy(rnd).to(DEV)}, batch_size=B, K=K, device=DEV)
    x = np.stack([rng.permutation(N) for _ in range(B)]).astype(np.int64)
    d = dict(u_partner=rng.rand(L, B, N).astype(np.float32), r_near=rng.randint(0, K, size=(L, B, N)).astype(np.int64),
             r_rand=rng.randint(0, N - K - 1, size=(L, B, N)).astype(np.int64),
             u_gumbel=rng.rand(L, B, N).astype(np.float32).clip(1e-7, 1 - 1e-7), u_accept=rng.rand(B).astype(np.float32))
    r = oi.tsp_step(x, dist, near, rnd, K, L, T, d["u_partner"], d["r_near"], d["r_rand"], d["u_gumbel"], d["u_accept"])
    y, mean_acc, log_acc, cur = s.step(torch.from_numpy(x).to(DEV), L, T, draws={k: torch.from_numpy(v) for k, v in d.items()}, want_terms=True)
    cur, y = cur.cpu().numpy(), y.cpu().numpy()
    assert (np.sort(cur, axis=1) == np.arange(N)).all() and (np.sort(y, axis=1) == np.arange(N)).all(), "not permutations " + tag
    same = (cur == r["cur_x"]).all(axis=1)
    assert same.mean() >= 0.97, f"walked tours differ on {(~same).sum()} of {B} envs " + tag      # near-tied argmax only
    la = log_acc.cpu().numpy()
    assert np.allclose(la[same], r["log_acc"][same], rtol=2e-5, atol=2e-4), "log_acc " + tag
    margin = np.abs(np.log(d["u_accept"].astype(np.float64) + 1e-24) - r["log_acc"])
    sure = same & (margin > 1e-3 * np.maximum(1.0, np.abs(r["log_acc"])))
    assert np.array_equal(y[sure], r["y"][sure]), "accepted tours " + tag
    exact += int(same.all())
    it += 1
print(f"fuzz_isco_tsp: {it} random configurations ({exact} with every walked tour identical), no mismatch")
