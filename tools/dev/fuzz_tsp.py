"""Differential fuzz of the TSP kernels against the numpy restatements: tour length (f32, 1e-5 relative), the 2-opt reversal
delta (f64), the best-improvement 2-opt pass in both rankings and the whole local_search_2_opt (routes and float64 distances
bit for bit), on Euclidean, integer (ties everywhere) and asymmetric matrices.  `python tools/dev/fuzz_tsp.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import ops_mcpg_tsp as mops
from rlsolver_amd.methods import tsp_opt_2 as t2

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    N = int(rng.choice([rng.randint(3, 9), rng.randint(9, 30), rng.randint(30, 70)]))
    kind = rng.choice(["euclid", "integer", "asym"])
    c = rng.rand(N, 2) * 100
    d = np.sqrt(((c[:, None] - c[None]) ** 2).sum(-1))
    if kind == "integer":
        d = np.rint(d / 20.0)
    elif kind == "asym":
        d = d + rng.rand(N, N) * 5
    np.fill_diagonal(d, 0)
    B = int(rng.randint(1, 40))
    perms = np.stack([rng.permutation(N) for _ in range(B)])
    dd = torch.from_numpy(d).to(DEV)
    pp = torch.from_numpy(perms).to(DEV)
    tag = f"it={it} N={N} kind={kind} B={B}"
    # K12
    d32 = d.astype(np.float32)
    got = mops.tsp_tour_length(torch.from_numpy(d32).to(DEV), pp).cpu().numpy()
    want = onp.tsp_tour_length_f64(d32, perms)
    assert np.allclose(got, want, rtol=1e-5, atol=0), "K12 " + tag
    # one exact pass per tour vs the restated loops
    cur = np.array([onp.tsp_distance_calc(d, [int(v) + 1 for v in p] + [int(p[0]) + 1]) for p in perms])
    bi, bj, bv = mops.tsp_2opt_best(dd, pp, torch.from_numpy(cur).to(DEV), slices=int(rng.choice([1, 3, 8])))
    for b in range(min(B, 4)):
        tour = [int(v) + 1 for v in perms[b]] + [int(perms[b][0]) + 1]
        r, dist = onp.tsp_local_search_2_opt(d, tour, cur[b], 1)
        if int(bi[b]) < 0:
            assert r == tour and dist == cur[b], "2-opt none " + tag
        else:
            i, j = int(bi[b]), int(bj[b])
            cand = list(tour)
            cand[i:j + 1] = cand[i:j + 1][::-1]
            cand[-1] = cand[0]
            assert cand == r and float(bv[b]) == dist, f"2-opt pass b={b} " + tag
    # the whole search on one tour
    if N <= 40:
        tour = [int(v) + 1 for v in perms[0]] + [int(perms[0][0]) + 1]
        rs = int(rng.choice([-1, 1, 3]))
        r, dist = onp.tsp_local_search_2_opt(d, tour, cur[0], rs)
        r2, dist2 = t2.local_search_2_opt(d, [tour, cur[0]], recursive_seeding=rs, verbose=False, device=DEV)
        assert r2 == r and dist2 == dist, f"local_search_2_opt rs={rs} " + tag
    # delta ranking on symmetric matrices: the reported delta is the best of all reversal deltas
    if kind != "asym":
        bi, bj, bd = mops.tsp_2opt_best(dd, pp)
        for b in range(min(B, 3)):
            best = 0.0
            for i in range(N - 1):
                for j in range(i + 1, N):
                    if i == 0 and j == N - 1:
                        continue
                    t = perms[b]
                    a_, b_, c_, e_ = t[i - 1], t[i], t[j], t[(j + 1) % N]
                    best = min(best, (d[a_, c_] + d[b_, e_]) - (d[a_, b_] + d[c_, e_]))
            assert float(bd[b]) == best, f"2-opt delta b={b} " + tag
    it += 1
print(f"fuzz_tsp: {it} random configurations, no mismatch")
