"""Invariant fuzz of the on-device MCPG round with PRODUCTION draws (no oracle: the kernels' own generators): random graphs
(G(n, m), BA, hubs), kept-chain counts M and repeats R around the tile sizes, a few rounds each.  Invariants: incumbents never
get worse, every incumbent value is the cut of its kept chain, the best value / index are the arg-max of the incumbents, the
worst incumbent has been replaced by the best, get_return is finite with a finite gradient.
`python tools/fuzz/fuzz_mcpg_round.py [seconds] [seed]`."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import oracle_np as onp
from rlsolver_amd import graph as G
from rlsolver_amd.methods import MCPG as amcpg
from rlsolver_amd.ops_mcpg_tsp import PackedChains

DEV = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t_end = time.time() + budget
it = 0
while time.time() < t_end:
    kind = rng.choice(["gnm", "ba", "hub"])
    n = int(rng.choice([rng.randint(12, 100), rng.randint(100, 600), rng.randint(600, 2500)]))
    if kind == "gnm":
        graph = np.asarray(G.generate_gnm(n, int(rng.randint(n, min(n * 6, n * (n - 1) // 2))), int(rng.randint(1 << 30))), dtype=np.int64)
    elif kind == "ba":
        graph = np.asarray(G.generate_ba(n, int(rng.randint(1, min(7, n - 1))), int(rng.randint(1 << 30))), dtype=np.int64)
    else:
        e = {(0, j) for j in rng.choice(np.arange(1, n), min(n - 1, int(rng.choice([70, 300, 900]))), replace=False).tolist()}
        for _ in range(2 * n):
            a, b = rng.randint(0, n, 2)
            if a != b:
                e.add((min(a, b), max(a, b)))
        graph = np.asarray([(a, b, 1) for a, b in sorted(e)], dtype=np.int64)
    M, R = int(rng.choice([64, 128, 320])), int(rng.choice([1, 3, 8]))
    num_ls = int(rng.randint(1, 4))
    tag = f"it={it} kind={kind} n={n} E={len(graph)} M={M} R={R} num_ls={num_ls}"
    if "-v" in sys.argv:
        print(tag, flush=True)
    torch.manual_seed(int(rng.randint(1 << 30)))
    ei = graph[:, :2].T.copy()
    data = amcpg.make_data(n, ei[0], ei[1], DEV)
    kept = (torch.rand((n, M), device=DEV) < 0.5).float()
    cut = lambda cols: torch.from_numpy(onp.maxcut_obj(cols.t().cpu().numpy().astype(np.uint8), graph, False)).float().to(DEV)
    rnd = amcpg.MCPGRound(data, PackedChains.pack(kept), cut(kept), M, R, num_ls)
    probs = torch.rand(n, device=DEV) * 0.6 + 0.2
    prev = rnd.now_max_res.clone()
    for r in range(3):
        rnd.step(probs)
        assert bool((rnd.now_max_res >= prev).all()), "incumbents got worse " + tag
        info = rnd.now_max_info.unpack()
        assert torch.equal(cut(info), rnd.now_max_res), "incumbent value is not the cut of its chain " + tag
        assert float(rnd.best_value) == float(rnd.now_max_res.max()) and float(rnd.now_max_res.min()) >= float(prev.min()), "best / worst " + tag
        assert float(rnd.now_max_res[int(rnd.best_index)]) == float(rnd.best_value), "best index " + tag
        prev = rnd.now_max_res.clone()
    pr = probs.clone().requires_grad_(True)
    obj = rnd.get_return(pr)
    obj.backward()
    assert bool(torch.isfinite(obj)) and bool(torch.isfinite(pr.grad).all()), "get_return " + tag
    it += 1
print(f"fuzz_mcpg_round: {it} random configurations, no violation")
